#!/bin/bash
# round 6, final tree, part B: rocprof summaries (kernel stats of the bench command, GPU-busy per step in both Fiat-Shamir modes, PMC passes), soaks
set -o pipefail
O=gpurun_out/r6f
mkdir -p $O
tools/make_profiles.sh r06 > $O/make_profiles.log 2>&1 || { tail -20 $O/make_profiles.log; exit 1; }
cp gpurun_out/prof/r06_* gpurun_out/prof/traffic.json $O/ 2>/dev/null
echo "profiles done"
BUSY_STEPS=60 BUSY_ARGS="--slots 6 --combine 6 --fs device" tools/gpu_busy.sh gpurun_out/prof/busy6d 80 > $O/r06_gpu_busy_1cohort_fs_device.txt 2>&1 || exit 1
echo "device busy done"
python tools/stress_combine.py 18 3000 0 > $O/r06_soak_host.txt 2>&1 || { tail -5 $O/r06_soak_host.txt; exit 1; }
STRESS_FS=device python tools/stress_combine.py 18 2000 0 > $O/r06_soak_device.txt 2>&1 || { tail -5 $O/r06_soak_device.txt; exit 1; }
STRESS_FS=device python tools/stress_combine.py 18 400 50 > $O/r06_soak_device_checked.txt 2>&1 || { tail -5 $O/r06_soak_device_checked.txt; exit 1; }
tail -2 $O/r06_soak_host.txt $O/r06_soak_device.txt $O/r06_soak_device_checked.txt
python bench.py --config 4 --fs device --no-kernels --no-cpu-baseline > $O/r06_bench_config4_fs_device.json 2> $O/c4d.err || { tail -10 $O/c4d.err; exit 1; }
python -c "
import json; j=json.loads([l for l in open('gpurun_out/r6f/r06_bench_config4_fs_device.json') if l.startswith('{')][-1]); print('config 4 device: %.1f k, %.2f ms, %.2f cores' % (j['value']/1e3, j['step_latency_ms']['median'], j['host_cpu_cores_busy']))"
