#!/bin/bash
# Round 5: HIP runtime tunables against the default line (one box, default run between the others).  usage: tools/r5_runtime_knobs.sh <outfile>
out=${1:-gpurun_out/r5/runtime_knobs.txt}; mkdir -p $(dirname $out); : > $out
run() { name=$1; shift
  j=$(timeout -k 5 90 env "$@" 2>/dev/null | tail -1)   # a knob that hangs the runtime (ROC_SYSTEM_SCOPE_SIGNAL=0 did) must not hang the sweep
  python3 - "$name" "$j" >> $out <<'PY'
import json, sys
try:
    j = json.loads(sys.argv[2]); l = j["step_latency_ms"]
    print("%-36s %8.0f proofs/s drained %8.0f | latency ms median %.2f p90 %.2f p99 %.2f | cores %.2f" % (
          sys.argv[1], j["value"], j["drained_run"]["value"], l["median"], l["p90"], l["p99"], j["host_cpu_cores_busy"]))
except Exception as e:
    print("%-36s failed: %r" % (sys.argv[1], e))
PY
  tail -1 $out; }
B="python bench.py --steps 720 --warmup 72 --no-kernels --no-cpu-baseline"
run "default" X=1 $B
for kv in DEBUG_CLR_LIMIT_BLIT_WG=4 DEBUG_CLR_LIMIT_BLIT_WG=16 DEBUG_CLR_LIMIT_BLIT_WG=64 DEBUG_CLR_LIMIT_BLIT_WG=256; do run "$kv" $kv $B; done
run "default" X=1 $B
for kv in HIP_FORCE_DEV_KERNARG=0 ROC_USE_FGS_KERNARG=0 DEBUG_CLR_BLIT_KERNARG_OPT=1 DEBUG_CLR_SKIP_RELEASE_SCOPE=1; do run "$kv" $kv $B; done
run "default" X=1 $B
for kv in ROC_ACTIVE_WAIT_TIMEOUT=0 ROC_ACTIVE_WAIT_TIMEOUT=100 ROC_CPU_WAIT_FOR_SIGNAL=0 ROC_AQL_QUEUE_SIZE=4096 AMD_DIRECT_DISPATCH=0 GPU_NUM_COMPUTE_RINGS=8; do run "$kv" $kv $B; done
run "default" X=1 $B
