#!/bin/bash
# Round 5: alternating A/B runs of the default line on ONE box.  usage: tools/r5_ab.sh <outfile> <reps> NAME=ENVVAR=VALUE ...  (NAME=- : no variable)
out=$1; reps=$2; shift 2
mkdir -p $(dirname $out); : > $out
for rep in $(seq 1 $reps); do
  for spec in "$@"; do
    name=${spec%%=*}; kv=${spec#*=}
    if [ "$kv" = "-" ]; then j=$(python bench.py --steps 450 --warmup 45 --no-kernels --no-cpu-baseline 2>/dev/null | tail -1)
    else j=$(env $kv python bench.py --steps 450 --warmup 45 --no-kernels --no-cpu-baseline 2>/dev/null | tail -1); fi
    python3 - "$name" "$j" >> $out <<'PY'
import json, sys
j = json.loads(sys.argv[2])
k = j["kernels_in_pipeline"]
print("%-22s %8.0f proofs/s  drained %8.0f  latency %.2f ms  cores %5.2f  frac %.4f | assemble %.0f lincomb %.0f expand1 %.0f v_hash %.0f+%.0f us" % (
    sys.argv[1], j["value"], j["drained_run"]["value"], j["step_latency_ms"]["median"], j["host_cpu_cores_busy"], j["roofline"]["frac"],
    k["assemble"]["avg_us"], k["lincomb"]["avg_us"], k["gemm_expand1"]["avg_us"], k["v_hash_tcomm"]["avg_us"], k["v_hash_view"]["avg_us"]))
PY
    tail -1 $out
  done
done
