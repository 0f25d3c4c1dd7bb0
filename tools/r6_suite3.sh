#!/bin/bash
# round 6: GPU suite on the cleaned tree, then larger cohorts (the combiner takes up to 16 callers since round 6) in both Fiat-Shamir modes
set -o pipefail
O=gpurun_out/r6
mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gputest2.log 2>&1 || { tail -40 $O/gputest2.log; exit 1; }
tail -3 $O/gputest2.log
for spec in "device 36 12" "device 48 16" "host 36 12" "host 24 8" "device 30 10"; do
  set -- $spec
  examples/throughput --fs $1 --callers $2 --combine $3 --steps 2400 --warmup 240 > $O/native_$1_c$3.json 2> $O/native_$1_c$3.err || { cat $O/native_$1_c$3.err; exit 1; }
  python - $O/native_$1_c$3.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read())
print("%s callers %d cohorts of %d fs %s: %.1f k proofs/s, step %.2f ms (p99 %.2f), %.2f cores, callers/run %.2f" % (sys.argv[1].split("/")[-1], j["callers"], j["handles_per_cohort"], j["fiat_shamir"], j["proofs_per_s"] / 1e3, j["step_latency_ms"]["median"], j["step_latency_ms"]["p99"], j["host_cpu_cores_busy"], j["mean_callers_per_run"]))
PY
done
