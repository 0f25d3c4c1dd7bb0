#!/bin/bash
# Round 5: the digest-table copy in 1 / 2 / 3 / 4 pieces (KOSK_TABLE_CHUNKS), alternating on ONE box, at the default arrangement (twelve
# callers in cohorts of four) and at nine callers in cohorts of three.   usage: tools/r5_chunks.sh <outfile> <reps>
out=${1:-gpurun_out/r5/chunks.txt}; reps=${2:-2}; mkdir -p $(dirname $out); : > $out
run() { name=$1; shift
  j=$(env "$@" 2>/dev/null | tail -1)
  python3 - "$name" "$j" >> $out <<'PY'
import json, sys
try:
    j = json.loads(sys.argv[2])
    print("%-34s %8.0f proofs/s  drained %8.0f  latency %.2f ms  p90 %.2f  cores %5.2f" % (sys.argv[1], j["value"], j["drained_run"]["value"],
          j["step_latency_ms"]["median"], j["step_latency_ms"].get("p90", 0), j["host_cpu_cores_busy"]))
except Exception as e:
    print("%-34s failed: %r" % (sys.argv[1], e))
PY
  tail -1 $out; }
B="python bench.py --steps 450 --warmup 45 --no-kernels --no-cpu-baseline"
for rep in $(seq 1 $reps); do
  for ch in 1 3 2 4; do
    run "12x4 chunks=$ch" KOSK_TABLE_CHUNKS=$ch $B
    run "9x3  chunks=$ch" KOSK_TABLE_CHUNKS=$ch $B --slots 9 --combine 3
  done
done
