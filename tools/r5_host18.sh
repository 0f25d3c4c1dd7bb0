#!/bin/bash
# Round 5: host cores of the default arrangement (eighteen callers in cohorts of six) under the wait / worker / pre-wake knobs, alternating
# on ONE box.   usage: tools/r5_host18.sh <outfile> <reps>
out=${1:-gpurun_out/r5/host18.txt}; reps=${2:-2}; mkdir -p $(dirname $out); : > $out
run() { name=$1; shift
  j=$(timeout -k 5 150 env "$@" 2>/dev/null | tail -1)
  python3 - "$name" "$j" >> $out <<'PY'
import json, sys
try:
    j = json.loads(sys.argv[2]); l = j["step_latency_ms"]
    print("%-40s %8.0f proofs/s drained %8.0f | latency ms median %.2f p99 %.2f | cores %.2f" % (
          sys.argv[1], j["value"], j["drained_run"]["value"], l["median"], l["p99"], j["host_cpu_cores_busy"]))
except Exception as e:
    print("%-40s failed: %r" % (sys.argv[1], e))
PY
  tail -1 $out; }
B="python bench.py --steps 1800 --warmup 180 --no-kernels --no-cpu-baseline"
for rep in $(seq 1 $reps); do
run "default (spin + nap, 4 workers, prewake)" X=1 $B
run "3 workers" KOSK_HOST_THREADS=3 $B
run "no pre-wake" KOSK_COMBINE_PREWAKE_US=0 $B
run "3 workers, no pre-wake" KOSK_HOST_THREADS=3 KOSK_COMBINE_PREWAKE_US=0 $B
run "sleeping waits, 3 workers, no pre-wake" KOSK_BLOCKING_SYNC=1 KOSK_HOST_THREADS=3 $B
run "sleeping waits, 4 workers, no pre-wake" KOSK_BLOCKING_SYNC=1 KOSK_HOST_THREADS=4 $B
run "sleeping waits, 2 workers, no pre-wake" KOSK_BLOCKING_SYNC=1 KOSK_HOST_THREADS=2 $B
done
