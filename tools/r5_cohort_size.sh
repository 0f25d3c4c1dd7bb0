#!/bin/bash
# Round 5: callers per cohort (three cohorts each), alternating on ONE box.  usage: tools/r5_cohort_size.sh <outfile> <reps>
out=${1:-gpurun_out/r5/cohort_size.txt}; reps=${2:-3}; mkdir -p $(dirname $out); : > $out
run() { name=$1; shift
  j=$(timeout -k 5 120 env "$@" 2>/dev/null | tail -1)
  python3 - "$name" "$j" >> $out <<'PY'
import json, sys
try:
    j = json.loads(sys.argv[2]); l = j["step_latency_ms"]
    print("%-28s %8.0f proofs/s drained %8.0f | latency ms median %.2f p90 %.2f p99 %.2f | per cohort %s | cores %.2f" % (
          sys.argv[1], j["value"], j["drained_run"]["value"], l["median"], l["p90"], l["p99"], l["per_cohort_mean"], j["host_cpu_cores_busy"]))
except Exception as e:
    print("%-28s failed: %r" % (sys.argv[1], e))
PY
  tail -1 $out; }
B="python bench.py --warmup 90 --no-kernels --no-cpu-baseline"
for rep in $(seq 1 $reps); do
run "12 callers, cohorts of 4" X=1 $B --steps 720
run "15 callers, cohorts of 5" X=1 $B --steps 720 --slots 15 --combine 5
run "18 callers, cohorts of 6" X=1 $B --steps 720 --slots 18 --combine 6
run "18 in 6s, 4 threads each" KOSK_HOST_THREADS=4 $B --steps 720 --slots 18 --combine 6
run "15 in 5s, 4 threads each" KOSK_HOST_THREADS=4 $B --steps 720 --slots 15 --combine 5
done
