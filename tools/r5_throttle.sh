#!/bin/bash
# Round 5: is the latency tail of long runs the container's CPU quota?  cgroup files + the line's cgroup_cpu object for runs of 1200 steps.
out=${1:-gpurun_out/r5/throttle.txt}; mkdir -p $(dirname $out); : > $out
{ echo "nproc $(nproc)"; for f in /sys/fs/cgroup/cpu.max /sys/fs/cgroup/cpu.stat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us /sys/fs/cgroup/cpu/cpu.stat /sys/fs/cgroup/cpuset.cpus.effective; do [ -r $f ] && { echo "== $f"; cat $f; }; done; cat /proc/self/cgroup; } >> $out 2>&1
run() { name=$1; shift
  j=$(env "$@" 2>/dev/null | tail -1)
  python3 - "$name" "$j" >> $out <<'PY'
import json, sys
try:
    j = json.loads(sys.argv[2]); l = j["step_latency_ms"]
    print("%-26s %8.0f proofs/s | median %.2f mean %.2f p90 %.2f p99 %.2f max %.2f | cores %.2f | cgroup %s" % (
          sys.argv[1], j["value"], l["median"], l["mean"], l["p90"], l["p99"], l["max"], j["host_cpu_cores_busy"], j.get("cgroup_cpu")))
except Exception as e:
    print("%-26s failed: %r" % (sys.argv[1], e))
PY
  tail -1 $out; }
B="python bench.py --steps 1200 --warmup 120 --no-kernels --no-cpu-baseline"
for rep in 1 2; do
run "default" $B
run "blocking sync" KOSK_BLOCKING_SYNC=1 $B
run "4 threads per caller" KOSK_HOST_THREADS=4 $B
run "no nap" KOSK_WAIT_NAP=0 $B
done
