#!/bin/bash
# round 6, final tree, part C: the whole GPU suite, then the two lines of record once more with their wall time
set -o pipefail
O=gpurun_out/r6f
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/r06_gputest.log 2>&1 || { tail -40 $O/r06_gputest.log; exit 1; }
tail -2 $O/r06_gputest.log
SECONDS=0
python bench.py --steps 20 --warmup 5 > $O/r06_bench_driver_flags_2.json 2> $O/bench_driver2.err || { tail -20 $O/bench_driver2.err; exit 1; }
echo "driver-flags bench wall time: $SECONDS s"
SECONDS=0
python bench.py > $O/r06_bench_default_2.json 2> $O/bench_default2.err || { tail -20 $O/bench_default2.err; exit 1; }
echo "default bench wall time: $SECONDS s"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6f/r06_bench_d*_2.json")):
    j = json.loads([l for l in open(f) if l.startswith("{")][-1])
    g = j["roofline"]["graded_65536"]
    print("%-34s %7.1f k/s drained %7.1f  lat %.2f/%.2f/%.2f ms cores %.2f frac=%.4f alone=%.4f ntt=%.3f (%.1f us) sha3=%.4f chain=%.0f us" % (f.split("/")[-1], j["value"] / 1e3, j["drained_run"]["value"] / 1e3,
          j["step_latency_ms"]["median"], j["step_latency_ms"]["p99"], j["step_latency_ms"]["max"], j["host_cpu_cores_busy"], j["roofline"]["frac"], j["roofline"].get("alone_frac") or 0,
          g["ntt256"]["frac_hbm_peak"], g["ntt256"]["us"], g["sha3_view"]["frac_hbm_peak"], j["kernels_65536_lanes"]["fs_chain_sha3_long"]["us"]))
    for key in ("native_callers", "native_callers_fs_device", "native_callers_fs_device_cohorts_of_16", "fiat_shamir_device", "cohorts_of_three", "cohorts_of_four", "uncombined", "one_cohort_alone"):
        v = j.get(key)
        if v:
            print("    %-40s %s" % (key, {a: (round(v[a], 2) if isinstance(v.get(a), float) else v.get(a)) for a in ("proofs_per_s", "host_cpu_cores_busy", "step_latency_ms_median", "error") if a in v}))
    print("    drop_in", j["drop_in"].get("proofs_per_s"), "cpu", j["cpu_baseline"]["value"], "traffic_extrapolated", j["roofline"]["traffic_extrapolated"], j["roofline"]["traffic"])
PY
