#!/bin/bash
# Round 5: BASELINE config 4 (Kyber-1024, 91 proofs per call, round hook per member) in other arrangements, alternating on ONE box.
# usage: tools/r5_config4.sh <outfile> <reps>
out=${1:-gpurun_out/r5/config4.txt}; reps=${2:-2}; mkdir -p $(dirname $out); : > $out
run() { name=$1; shift
  j=$("$@" 2>/dev/null | tail -1)
  python3 - "$name" "$j" >> $out <<'PY'
import json, sys
try:
    j = json.loads(sys.argv[2])
    print("%-34s %8.0f proofs/s  drained %8.0f  latency %.2f ms  p90 %.2f  cores %5.2f  callers/run %.2f" % (sys.argv[1], j["value"], j["drained_run"]["value"],
          j["step_latency_ms"]["median"], j["step_latency_ms"].get("p90", 0), j["host_cpu_cores_busy"], (j.get("combining") or {}).get("mean_callers_per_run", 0)))
except Exception as e:
    print("%-34s failed: %r" % (sys.argv[1], e))
PY
  tail -1 $out; }
B="python bench.py --config 4 --steps 300 --warmup 36 --no-kernels --no-cpu-baseline"
for rep in $(seq 1 $reps); do
  run "config 4, 9 callers in 3s" $B
  run "config 4, 12 callers in 4s" $B --slots 12 --combine 4
  run "config 4, 8 callers in 4s" $B --slots 8 --combine 4
  run "config 4, 6 callers in 2s" $B --slots 6 --combine 2
done
