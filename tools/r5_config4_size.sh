#!/bin/bash
# Round 5: BASELINE config 4 (Kyber-1024, 91 proofs per call) at other cohort sizes, alternating on ONE box.  usage: tools/r5_config4_size.sh <outfile> <reps>
out=${1:-gpurun_out/r5/config4_size.txt}; reps=${2:-2}; mkdir -p $(dirname $out); : > $out
run() { name=$1; shift
  j=$(timeout -k 5 200 env "$@" 2>/dev/null | tail -1)
  python3 - "$name" "$j" >> $out <<'PY'
import json, sys
try:
    j = json.loads(sys.argv[2]); l = j["step_latency_ms"]
    print("%-40s %8.0f proofs/s drained %8.0f | latency ms median %.2f p99 %.2f | cores %.2f | callers/run %.2f" % (
          sys.argv[1], j["value"], j["drained_run"]["value"], l["median"], l["p99"], j["host_cpu_cores_busy"], (j.get("combining") or {}).get("mean_callers_per_run", 0)))
except Exception as e:
    print("%-40s failed: %r" % (sys.argv[1], e))
PY
  tail -1 $out; }
B="python bench.py --config 4 --steps 1440 --warmup 144 --no-kernels --no-cpu-baseline"
for rep in $(seq 1 $reps); do
run "9 callers in 3s (default, 6 workers)" X=1 $B
run "9 in 3s, 3 workers, no pre-wake" KOSK_HOST_THREADS=3 KOSK_COMBINE_PREWAKE_US=0 $B
run "12 in 4s, 3 workers, no pre-wake" KOSK_HOST_THREADS=3 KOSK_COMBINE_PREWAKE_US=0 $B --slots 12 --combine 4
run "15 in 5s, 3 workers" KOSK_HOST_THREADS=3 $B --slots 15 --combine 5
run "18 in 6s, 3 workers" KOSK_HOST_THREADS=3 $B --slots 18 --combine 6
done
