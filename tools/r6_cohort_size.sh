#!/bin/bash
# Cohort size at three cohorts in flight (native callers, host Fiat-Shamir): 6 / 8 / 10 / 12 / 16 callers per merged run
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r6; mkdir -p $O
show() { python - $1 "$2" <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
s = j["step_latency_ms"]
print("%-34s %7.1f k  median %.2f p99 %.2f max %.2f  cores %.2f  callers/run %.2f" % (sys.argv[2], j["proofs_per_s"] / 1e3, s["median"], s["p99"], s["max"], j["host_cpu_cores_busy"], j["mean_callers_per_run"]))
PY
}
for rep in 1 2; do
  for cfg in "18 6 3" "24 8 3" "24 8 2" "30 10 2" "36 12 2" "48 16 1" "48 16 2"; do
    set -- $cfg
    examples/throughput --callers $1 --combine $2 --threads $3 --steps 3600 --warmup 180 > $O/cs_$1_$2_$3_$rep.json 2> $O/cs.err || { tail -5 $O/cs.err; exit 1; }
    show $O/cs_$1_$2_$3_$rep.json "callers $1 cohort $2 threads $3"
  done
done
