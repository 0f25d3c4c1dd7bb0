#!/bin/bash
# Round 5, host cost per GPU: the default line (spinning waits, pool workers spin 20 us) against napping waits (KOSK_WAIT_NAP=1),
# non-spinning pool workers (KOSK_POOL_SPIN_US=0), both, and sleeping waits (KOSK_BLOCKING_SYNC=1), alternating on ONE box.
# usage: tools/r5_sweep_host.sh <outfile>
out=${1:-gpurun_out/r5/sweep_host.txt}
mkdir -p $(dirname $out)
: > $out
run() { # name, env...
    name=$1; shift
    j=$(env "$@" python bench.py --steps 450 --warmup 45 --no-kernels --no-cpu-baseline 2>/dev/null | tail -1)
    python3 - "$name" "$j" >> $out <<'PY'
import json, sys
j = json.loads(sys.argv[2])
print("%-28s %8.0f proofs/s  drained %8.0f  latency %.2f ms  cores busy %5.2f  frac %.4f  callers/run %.2f" % (
    sys.argv[1], j["value"], j["drained_run"]["value"], j["step_latency_ms"]["median"], j["host_cpu_cores_busy"], j["roofline"]["frac"],
    (j.get("combining") or {}).get("mean_callers_per_run", 0)))
PY
    tail -1 $out
}
for rep in 1 2; do
    run "default" KOSK_X=0
    run "nap" KOSK_WAIT_NAP=1
    run "nap+poolspin0" KOSK_WAIT_NAP=1 KOSK_POOL_SPIN_US=0
    run "poolspin0" KOSK_POOL_SPIN_US=0
    run "nap+poolspin0+prewake100" KOSK_WAIT_NAP=1 KOSK_POOL_SPIN_US=0 KOSK_COMBINE_PREWAKE_US=100
    run "blocking" KOSK_BLOCKING_SYNC=1
done
