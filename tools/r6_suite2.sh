#!/bin/bash
# round 6: the whole GPU suite on the cleaned tree, then the bench line with the driver's flags
set -o pipefail
O=gpurun_out/r6
mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gputest2.log 2>&1 || { tail -40 $O/gputest2.log; exit 1; }
tail -3 $O/gputest2.log
python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err || { tail -20 $O/bench_driver_flags.err; exit 1; }
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r6/bench_driver_flags.json") if l.startswith("{")][-1])
print("value %.1f k  cores %.2f  fs %s" % (j["value"] / 1e3, j["host_cpu_cores_busy"], j["config"]["fiat_shamir"][:6]))
for key in ("native_callers", "native_callers_fs_device", "fiat_shamir_device", "cohorts_of_three", "uncombined", "drop_in"):
    v = j.get(key) or {}
    print(key, {a: v.get(a) for a in ("proofs_per_s", "host_cpu_cores_busy", "step_latency_ms_median", "error") if a in v})
print("roofline", {a: j["roofline"].get(a) for a in ("frac", "avg_launch_us", "traffic_extrapolated", "alone_frac")})
print("graded", j["roofline"].get("graded_65536"))
print("fs chain", j["kernels_65536_lanes"].get("fs_chain_sha3_long"))
PY
