#!/bin/bash
# Sample the driver's gpu_busy_percent while a multi-slot run is in flight (not product code).
slots=${1:-6}
python3 tools/slot_latency.py $slots 1500 > gpurun_out/busy_run.log 2>&1 &
pid=$!
sleep 12
for i in $(seq 1 20); do
  for f in /sys/class/drm/card*/device/gpu_busy_percent; do printf "%s " "$(cat $f 2>/dev/null)"; done; echo
  sleep 0.1
done
wait $pid
head -2 gpurun_out/busy_run.log
