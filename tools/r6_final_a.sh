#!/bin/bash
# round 6, final tree, part A: the line of record (default command and the driver's flags), configs 2 / 4 / 5, batch-of-1 latency
set -o pipefail
O=gpurun_out/r6f
mkdir -p $O
python bench.py > $O/r06_bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
echo "default done"
python bench.py --steps 20 --warmup 5 > $O/r06_bench_driver_flags.json 2> $O/bench_driver.err || { tail -20 $O/bench_driver.err; exit 1; }
echo "driver flags done"
for c in 2 4 5; do
  python bench.py --config $c --no-kernels --no-cpu-baseline > $O/r06_bench_config$c.json 2> $O/bench_c$c.err || { tail -20 $O/bench_c$c.err; exit 1; }
  echo "config $c done"
done
python tools/latency_one.py > $O/r06_latency_one.txt 2>&1 || exit 1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6f/r06_bench_*.json")):
    j = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print("%-34s %7.1f k/s drained %7.1f  lat %.2f/%.2f ms cores %.2f fs=%s frac=%s" % (f.split("/")[-1], j["value"] / 1e3, j["drained_run"]["value"] / 1e3,
          j["step_latency_ms"]["median"], j["step_latency_ms"]["p99"], j["host_cpu_cores_busy"], j["config"]["fiat_shamir"][:6], (j.get("roofline") or {}).get("frac")))
    for key in ("native_callers", "native_callers_fs_device", "native_callers_fs_device_cohorts_of_16", "fiat_shamir_device", "cohorts_of_three", "cohorts_of_four", "uncombined", "one_cohort_alone"):
        v = j.get(key)
        if v:
            print("    %-40s %s" % (key, {a: (round(v[a], 2) if isinstance(v.get(a), float) else v.get(a)) for a in ("proofs_per_s", "host_cpu_cores_busy", "step_latency_ms_median", "error") if a in v}))
PY
