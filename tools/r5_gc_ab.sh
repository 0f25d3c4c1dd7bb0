#!/bin/bash
# Round 5: long runs (1200 steps) with the harness's cyclic collector off (default) / on (KOSK_BENCH_GC=1), alternating on one box.
out=${1:-gpurun_out/r5/gc_ab.txt}; mkdir -p $(dirname $out); : > $out
run() { name=$1; shift
  j=$(env "$@" 2>/dev/null | tail -1)
  python3 - "$name" "$j" >> $out <<'PY'
import json, sys
try:
    j = json.loads(sys.argv[2]); l = j["step_latency_ms"]
    print("%-22s %8.0f proofs/s drained %8.0f | latency ms median %.2f mean %.2f p90 %.2f p99 %.2f max %.2f | per cohort %s | cores %.2f" % (
          sys.argv[1], j["value"], j["drained_run"]["value"], l["median"], l["mean"], l["p90"], l["p99"], l["max"], l["per_cohort_mean"], j["host_cpu_cores_busy"]))
except Exception as e:
    print("%-22s failed: %r" % (sys.argv[1], e))
PY
  tail -1 $out; }
B="python bench.py --steps 1200 --warmup 120 --no-kernels --no-cpu-baseline"
for rep in 1 2 3 4; do
run "collector off" X=1 $B
run "collector on" KOSK_BENCH_GC=1 $B
done
run "driver flags" X=1 python bench.py --steps 20 --warmup 5 --no-kernels --no-cpu-baseline
run "driver flags" X=1 python bench.py --steps 20 --warmup 5 --no-kernels --no-cpu-baseline
