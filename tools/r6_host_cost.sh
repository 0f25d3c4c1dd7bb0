#!/bin/bash
# host mode after the key-record job became one job per merged run: keygen / combine suites, native callers (cores), the line
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r6f; mkdir -p $O
python -m pytest tests/test_gpu_01_prover.py tests/test_gpu_10_combine.py tests/test_gpu_07_api_paths.py -m gpu -x -q 2>&1 | tail -2 || exit 1
for rep in 1 2; do
  examples/throughput --steps 3600 --warmup 180 > $O/hc_$rep.json 2> $O/hc.err || { tail -5 $O/hc.err; exit 1; }
  python - $O/hc_$rep.json <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); s = j["step_latency_ms"]
print("native 18/6 host: %.1f k  median %.2f p99 %.2f max %.2f  cores %.2f" % (j["proofs_per_s"] / 1e3, s["median"], s["p99"], s["max"], j["host_cpu_cores_busy"]))
PY
done
