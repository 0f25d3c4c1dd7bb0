#!/bin/bash
# round 6: the line of record's arrangement with the Fiat-Shamir hashes on the host / on the device, alternating on one box
set -o pipefail
O=gpurun_out/r6
mkdir -p $O
python tools/fs_chain_time.py $O/fs_chain_time_bperm.txt > $O/t_bperm.log 2>&1 || exit 1
KOSK_FS_SPONGE=lds python tools/fs_chain_time.py $O/fs_chain_time_lds.txt > $O/t_lds.log 2>&1 || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_11_fs_device.py -x -q -m gpu > $O/t2_tests.log 2>&1 || exit 1
B="python bench.py --no-kernels --no-cpu-baseline --steps 1800 --warmup 180"
for i in 1 2; do
  $B --fs host   > $O/ab_host_$i.json   2> $O/ab_host_$i.err   || exit 1
  $B --fs device > $O/ab_device_$i.json 2> $O/ab_device_$i.err || exit 1
done
$B --fs device --slots 24 --combine 8 > $O/ab_device_c8.json 2> $O/ab_device_c8.err || exit 1
$B --fs device --slots 6 --combine 6  > $O/ab_device_1cohort.json 2> $O/ab_device_1cohort.err || exit 1
$B --fs host --slots 6 --combine 6    > $O/ab_host_1cohort.json 2> $O/ab_host_1cohort.err || exit 1
GPU_MAX_HW_QUEUES=8 $B --fs device --slots 24 --combine 6 > $O/ab_device_4cohorts_q8.json 2> $O/ab_device_4cohorts_q8.err || exit 1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6/ab_*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "ERR", e); continue
    k = j.get("kernels_in_pipeline", {})
    print("%-44s %8.1f k/s  drained %8.1f  lat med %.2f p99 %.2f ms  cores %.2f  fs_alpha %s us  hash_view %s us" % (
        f.split("/")[-1], j["value"] / 1e3, j["drained_run"]["value"] / 1e3, j["step_latency_ms"]["median"], j["step_latency_ms"]["p99"],
        j["host_cpu_cores_busy"], round(k.get("fs_alpha", {}).get("avg_us", 0), 1), round(k.get("hash_view", {}).get("avg_us", 0), 1)))
PY
