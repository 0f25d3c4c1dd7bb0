#!/bin/bash
# Round 5: two cohorts pipelined on ONE stream / hardware queue (KOSK_SHARE_STREAMS=n: the process's contexts are dealt round n streams).
# usage: tools/r5_share.sh <outfile>
out=${1:-gpurun_out/r5/share.txt}; mkdir -p $(dirname $out); : > $out
run() { name=$1; shift
  j=$(env "$@" 2>/dev/null | tail -1)
  python3 - "$name" "$j" >> $out <<'PY'
import json, sys
try:
    j = json.loads(sys.argv[2]); l = j["step_latency_ms"]; cg = j.get("cgroup_cpu") or {}
    print("%-44s %8.0f proofs/s drained %8.0f | latency ms median %.2f p90 %.2f p99 %.2f | per cohort %s | cores %.2f throttled %s ms" % (
          sys.argv[1], j["value"], j["drained_run"]["value"], l["median"], l["p90"], l["p99"], l["per_cohort_mean"], j["host_cpu_cores_busy"], cg.get("throttled_ms_in_run")))
except Exception as e:
    print("%-44s failed: %r" % (sys.argv[1], e))
PY
  tail -1 $out; }
B="python bench.py --steps 720 --warmup 72 --no-kernels --no-cpu-baseline"
for rep in 1 2; do
run "12 callers, 3 cohorts of 4, 3 streams (default)" X=1 $B
run "24 callers, 6 cohorts of 4 on 3 streams" KOSK_SHARE_STREAMS=3 $B --slots 24 --combine 4
run "18 callers, 6 cohorts of 3 on 3 streams" KOSK_SHARE_STREAMS=3 $B --slots 18 --combine 3
run "12 callers, 6 cohorts of 2 on 3 streams" KOSK_SHARE_STREAMS=3 $B --slots 12 --combine 2
run "16 callers, 4 cohorts of 4 on 2 streams" KOSK_SHARE_STREAMS=2 $B --slots 16 --combine 4
run "24 callers, 6 cohorts of 4 on 2 streams" KOSK_SHARE_STREAMS=2 $B --slots 24 --combine 4
run "18 callers, 6 cohorts of 3, own streams" X=1 $B --slots 18 --combine 3
done
