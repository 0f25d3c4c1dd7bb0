#!/bin/bash
# Round 5: other arrangements of callers / cohorts / hardware queues on the final kernels (one box).  usage: tools/r5_arrangements.sh <outfile>
out=${1:-gpurun_out/r5/arrangements.txt}; mkdir -p $(dirname $out); : > $out
run() { name=$1; shift
  j=$(env "$@" 2>/dev/null | tail -1)
  python3 - "$name" "$j" >> $out <<'PY'
import json, sys
try:
    j = json.loads(sys.argv[2])
    print("%-40s %8.0f proofs/s  drained %8.0f  latency %.2f ms  cores %5.2f  frac %.4f  callers/run %.2f" % (sys.argv[1], j["value"], j["drained_run"]["value"],
          j["step_latency_ms"]["median"], j["host_cpu_cores_busy"], j["roofline"]["frac"], (j.get("combining") or {}).get("mean_callers_per_run", 0)))
except Exception as e:
    print("%-40s failed: %r" % (sys.argv[1], e))
PY
  tail -1 $out; }
B="python bench.py --steps 450 --warmup 45 --no-kernels --no-cpu-baseline"
run "9 callers, 3 cohorts of 3 (default)" $B
run "12 callers, 4 cohorts of 3" $B --slots 12 --combine 3
run "12 callers, 4 cohorts, 8 hw queues" GPU_MAX_HW_QUEUES=8 $B --slots 12 --combine 3
run "12 callers, 3 cohorts of 4" $B --slots 12 --combine 4
run "8 callers, 2 cohorts of 4" $B --slots 8 --combine 4
run "6 callers, 2 cohorts of 3" $B --slots 6 --combine 3
run "9 callers, 3 cohorts, 8 hw queues" GPU_MAX_HW_QUEUES=8 $B
run "9 callers, 3 cohorts of 3 (default)" $B
