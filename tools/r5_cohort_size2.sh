#!/bin/bash
# Round 5: larger cohorts with the lean host settings of the final default (three workers per caller, no pre-wake), alternating on ONE box.
out=${1:-gpurun_out/r5/cohort_size2.txt}; reps=${2:-2}; mkdir -p $(dirname $out); : > $out
run() { name=$1; shift
  j=$(timeout -k 5 150 env "$@" 2>/dev/null | tail -1)
  python3 - "$name" "$j" >> $out <<'PY'
import json, sys
try:
    j = json.loads(sys.argv[2]); l = j["step_latency_ms"]
    print("%-30s %8.0f proofs/s drained %8.0f | latency ms median %.2f p90 %.2f p99 %.2f | cores %.2f | frac %.4f" % (
          sys.argv[1], j["value"], j["drained_run"]["value"], l["median"], l["p90"], l["p99"], j["host_cpu_cores_busy"], j["roofline"]["frac"]))
except Exception as e:
    print("%-30s failed: %r" % (sys.argv[1], e))
PY
  tail -1 $out; }
B="python bench.py --warmup 180 --no-kernels --no-cpu-baseline --steps 2520"
for rep in $(seq 1 $reps); do
run "18 callers, cohorts of 6" X=1 $B
run "21 callers, cohorts of 7" X=1 $B --slots 21 --combine 7
run "24 callers, cohorts of 8" X=1 $B --slots 24 --combine 8
run "30 callers, cohorts of 10" X=1 $B --slots 30 --combine 10
run "24 in 8s, 2 workers each" KOSK_HOST_THREADS=2 $B --slots 24 --combine 8
done
