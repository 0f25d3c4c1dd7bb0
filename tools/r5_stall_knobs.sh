#!/bin/bash
# Round 5: the periodic 4-9 ms stalls of long runs are the HIP runtime growing a system-memory pool inside a launch / copy call (an 8 MB
# KFD allocation + an SVM ioctl of ~4 ms under a runtime-wide lock; tools/slow_syscalls.c).  Which runtime knob moves them?
# usage: tools/r5_stall_knobs.sh <outfile> [steps per caller]
out=${1:-gpurun_out/r5/stall_knobs.txt}; n=${2:-200}; mkdir -p $(dirname $out); : > $out
gcc -shared -fPIC -O1 -o tools/libslow_syscalls.so tools/slow_syscalls.c -ldl || exit 1
run() { name=$1; shift
  env "$@" TAIL_GC=0 SLOW_KFD=1 SLOW_US=1500 LD_PRELOAD=$PWD/tools/libslow_syscalls.so timeout -k 10 120 python tools/tail_probe.py $n > /tmp/stall_run.txt 2>&1
  python3 - "$name" >> $out <<'PY'
import re, sys
t_s = None; allocs = []; slow = []; head = lat = ""
for l in open("/tmp/stall_run.txt", errors="replace"):
    m = re.match(r"run started at CLOCK_MONOTONIC ([\d.]+)", l)
    if m: t_s = float(m.group(1))
    if "callers x" in l: head = l.split(":")[1].strip()
    if l.startswith("latency ms"): lat = l.strip()
    m = re.match(r"\[kfd\] t=([\d.]+) ms\s+([\d.]+) ms tid \d+ ALLOC va \S+ size (\d+)", l)
    if m: allocs.append((float(m.group(1)), int(m.group(3))))
    m = re.match(r"\[slow\] t=([\d.]+) ms\s+([\d.]+) ms tid \d+ (\S+) (\S+)", l)
    if m: slow.append((float(m.group(1)), float(m.group(2)), m.group(3) + " " + m.group(4)))
if t_s is None:
    print("%-34s failed" % sys.argv[1]); sys.exit(0)
# the run proper: after its 20 untimed steps (~70 ms) and before the handles are closed
inrun = lambda t: t_s + 30 < t
t_end = t_s + 1e9
big = [(t - t_s, s) for t, s in allocs if inrun(t) and s >= (1 << 20)]
sl = [(round(t - t_s), round(d, 1), w) for t, d, w in slow if inrun(t) and "munmap" not in w]
print("%-34s %s | %s | in-run KFD allocations >= 1 MB: %d %s | slow ioctls in run: %s" % (sys.argv[1], head, lat, len(big), [(round(t), s >> 20) for t, s in big][:6], sl[:6]))
PY
  tail -1 $out | cut -c1-400; }
run "default" X=1
run "DEBUG_CLR_SYSMEM_POOL=0" DEBUG_CLR_SYSMEM_POOL=0
run "HIP_FORCE_DEV_KERNARG=1" HIP_FORCE_DEV_KERNARG=1
run "HIP_FORCE_DEV_KERNARG=0" HIP_FORCE_DEV_KERNARG=0
run "HSA_KERNARG_POOL_SIZE=64M" HSA_KERNARG_POOL_SIZE=67108864
run "DEBUG_CLR_BLIT_KERNARG_OPT=1" DEBUG_CLR_BLIT_KERNARG_OPT=1
run "DEBUG_HIP_KERNARG_COPY_OPT=0" DEBUG_HIP_KERNARG_COPY_OPT=0
run "ROC_SIGNAL_POOL_SIZE=4096" ROC_SIGNAL_POOL_SIZE=4096
run "GPU_STAGING_BUFFER_SIZE=64" GPU_STAGING_BUFFER_SIZE=64
run "default again" X=1
