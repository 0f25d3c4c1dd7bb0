#!/bin/bash
# device mode after the key-record job moved on to the caller's thread: workers per caller should no longer matter
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r6f; mkdir -p $O
python -m pytest tests/test_gpu_11_fs_device.py tests/test_gpu_10_combine.py -m gpu -x -q 2>&1 | tail -2 || exit 1
for cfg in "48 16 3" "48 16 1" "18 6 3" "18 6 1"; do
  set -- $cfg
  examples/throughput --fs device --callers $1 --combine $2 --threads $3 --steps 2400 --warmup 180 > $O/dev2_$1_$2_$3.json 2> $O/dev.err || { tail -5 $O/dev.err; exit 1; }
  python - $O/dev2_$1_$2_$3.json "$cfg" <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); s = j["step_latency_ms"]
print("device mode, callers/cohort/threads %-9s %7.1f k  median %.2f p99 %.2f  cores %.2f" % (sys.argv[2], j["proofs_per_s"] / 1e3, s["median"], s["p99"], j["host_cpu_cores_busy"]))
PY
done
