#!/bin/bash
# Round 5: alternating A/B runs on ONE box, any bench arguments.  usage: tools/r5_ab2.sh <outfile> <reps> "<bench args>" NAME=ENVVAR=VALUE ...  (NAME=- : no variable)
out=$1; reps=$2; bargs=$3; shift 3
mkdir -p $(dirname $out); : >> $out
for rep in $(seq 1 $reps); do
  for spec in "$@"; do
    name=${spec%%=*}; kv=${spec#*=}
    if [ "$kv" = "-" ]; then j=$(timeout -k 5 90 python bench.py $bargs --no-kernels --no-cpu-baseline 2>/dev/null | tail -1)
    else j=$(timeout -k 5 90 env $kv python bench.py $bargs --no-kernels --no-cpu-baseline 2>/dev/null | tail -1); fi
    python3 - "$name [$bargs]" "$j" >> $out <<'PY'
import json, sys
try:
    j = json.loads(sys.argv[2])
    print("%-60s %8.0f proofs/s  drained %8.0f  latency %.2f ms  p90 %.2f  cores %5.2f" % (sys.argv[1], j["value"], j["drained_run"]["value"],
          j["step_latency_ms"]["median"], j["step_latency_ms"].get("p90", 0), j["host_cpu_cores_busy"]))
except Exception as e:
    print("%-60s failed: %r" % (sys.argv[1], e))
PY
    tail -1 $out
  done
done
