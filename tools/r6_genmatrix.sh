#!/bin/bash
# gen_matrix on the wave sponge (prover role G of k_prover_pre, verifier k_gen_matrix_wave): key generation / verifier / edge suites, kernel times
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r6; mkdir -p $O
python -m pytest tests/test_gpu_01_prover.py tests/test_gpu_02_verify.py tests/test_gpu_03_split_api.py tests/test_gpu_04_configs.py tests/test_gpu_07_api_paths.py tests/test_gpu_09_edges.py -m gpu -x -q 2>&1 | tail -3 || exit 1
BUSY_STEPS=60 BUSY_ARGS="--slots 6 --combine 6" tools/gpu_busy.sh gpurun_out/prof/busy6 40 > $O/gm_busy.txt 2>&1 || exit 1
grep -E "prover_pre|gen_matrix|decode_pk|GPU busy" $O/gm_busy.txt
for rep in 1 2; do
  examples/throughput --steps 3600 --warmup 180 > $O/gm_native_$rep.json 2> $O/gm.err || { tail -5 $O/gm.err; exit 1; }
  python - $O/gm_native_$rep.json <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); s = j["step_latency_ms"]
print("native 18/6 host: %.1f k  median %.2f p99 %.2f max %.2f  cores %.2f" % (j["proofs_per_s"] / 1e3, s["median"], s["p99"], s["max"], j["host_cpu_cores_busy"]))
PY
done
