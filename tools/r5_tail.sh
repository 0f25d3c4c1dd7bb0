#!/bin/bash
# Round 5: the distribution of step latencies per arrangement (a mean far above the median = callers stalled somewhere).  usage: tools/r5_tail.sh <outfile>
out=${1:-gpurun_out/r5/tail.txt}; mkdir -p $(dirname $out); : > $out
run() { name=$1; shift
  j=$(env "$@" 2>/dev/null | tail -1)
  python3 - "$name" "$j" >> $out <<'PY'
import json, sys
try:
    j = json.loads(sys.argv[2]); l = j["step_latency_ms"]
    print("%-30s %8.0f proofs/s | latency ms: median %.2f mean %.2f p90 %.2f p99 %.2f max %.2f | in keygen call %.2f in verify call %.2f | callers/run %.2f | per cohort %s" % (
          sys.argv[1], j["value"], l["median"], l["mean"], l["p90"], l["p99"], l["max"], l["mean_in_keygen_call"], l["mean_in_verify_call"], (j.get("combining") or {}).get("mean_callers_per_run", 0), l.get("per_cohort_mean")))
except Exception as e:
    print("%-30s failed: %r" % (sys.argv[1], e))
PY
  tail -1 $out; }
B="python bench.py --steps 480 --warmup 48 --no-kernels --no-cpu-baseline"
for rep in 1 2; do
run "12 callers, 3 cohorts of 4" $B
run "12 callers, 4 cohorts of 3" $B --slots 12 --combine 3
run "9 callers, 3 cohorts of 3" $B --slots 9 --combine 3
run "8 callers, 2 cohorts of 4" $B --slots 8 --combine 4
done
