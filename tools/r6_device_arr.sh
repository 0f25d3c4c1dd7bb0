#!/bin/bash
# device Fiat-Shamir: callers / cohort size / workers per caller against throughput and busy host cores (native callers)
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r6f; mkdir -p $O
for cfg in "18 6 1" "27 9 1" "36 12 1" "48 16 1" "48 16 3" "30 10 1" "24 8 1"; do
  set -- $cfg
  examples/throughput --fs device --callers $1 --combine $2 --threads $3 --steps 2400 --warmup 180 > $O/dev_$1_$2_$3.json 2> $O/dev.err || { tail -5 $O/dev.err; exit 1; }
  python - $O/dev_$1_$2_$3.json "$cfg" <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); s = j["step_latency_ms"]
print("device mode, callers/cohort/threads %-9s %7.1f k  median %.2f p99 %.2f  cores %.2f  callers/run %.2f" % (sys.argv[2], j["proofs_per_s"] / 1e3, s["median"], s["p99"], j["host_cpu_cores_busy"], j["mean_callers_per_run"]))
PY
done
