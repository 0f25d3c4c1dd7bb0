#!/bin/bash
# round 6: the whole GPU suite (abort-trace helper preloaded when it exists), then the native caller harness in both Fiat-Shamir modes
set -o pipefail
O=gpurun_out/r6
mkdir -p $O
PRE=""
[ -f tools/libabort_trace.so ] && PRE="tools/libabort_trace.so"
LD_PRELOAD=$PRE timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gputest.log 2>&1 || { tail -30 $O/gputest.log; exit 1; }
tail -3 $O/gputest.log
for fs in host device; do
  examples/throughput --fs $fs --steps 1800 --warmup 180 > $O/native_$fs.json 2> $O/native_$fs.err || { cat $O/native_$fs.err; exit 1; }
  cat $O/native_$fs.json
done
examples/throughput --fs host --steps 1800 --warmup 180 --blocking 1 --threads 2 > $O/native_host_blocking2.json 2> $O/native_host_b.err || exit 1
cat $O/native_host_blocking2.json
examples/throughput --fs device --callers 24 --combine 8 --steps 1800 --warmup 180 > $O/native_device_c8.json 2> $O/native_dev_c8.err || exit 1
cat $O/native_device_c8.json
