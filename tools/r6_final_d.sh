#!/bin/bash
# after bench.py's device-mode arrangement: the sharding rehearsals (two ranks on one GPU: eight cores per rank -> device mode), the default
# line (host mode, unchanged) and python bench.py --fs device with its side legs
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r6f; mkdir -p $O
python -m pytest tests/test_sharding.py tests/test_gpu_11_fs_device.py -m gpu -x -q 2>&1 | tail -3 || exit 1
python bench.py --steps 20 --warmup 5 > $O/r06c_bench_driver_flags.json 2> $O/d.err || { tail -10 $O/d.err; exit 1; }
python bench.py --fs device --no-cpu-baseline > $O/r06c_bench_fs_device.json 2> $O/d.err || { tail -10 $O/d.err; exit 1; }
python - <<'PY'
import json
for f in ("gpurun_out/r6f/r06c_bench_driver_flags.json", "gpurun_out/r6f/r06c_bench_fs_device.json"):
    j = json.loads([l for l in open(f) if l.startswith("{")][-1]); s = j["step_latency_ms"]
    print("%-32s %.1f k drained %.1f k lat %.2f/%.2f cores %.2f fs=%s in flight %s" % (f.split("/")[-1], j["value"] / 1e3, j["drained_run"]["value"] / 1e3, s["median"], s["p99"], j["host_cpu_cores_busy"], j["config"]["fiat_shamir"][:6], j.get("proofs_in_flight_per_gpu")))
    for key in ("native_callers", "native_callers_fs_device", "native_callers_fs_host", "native_callers_fs_device_cohorts_of_16", "fiat_shamir_device", "fiat_shamir_host", "cohorts_of_three", "one_cohort_alone"):
        v = j.get(key)
        if v: print("    %-40s %s" % (key, {a: (round(v[a], 2) if isinstance(v.get(a), float) else v.get(a)) for a in ("proofs_per_s", "host_cpu_cores_busy", "step_latency_ms_median", "error") if a in v}))
PY
