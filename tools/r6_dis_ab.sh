#!/bin/bash
# The tree before / after the verifier's image scatter skips the records nobody reads (and the beta / gamma product's clamp), on ONE box,
# alternating: native callers, the line of record's arrangement (18 callers, cohorts of six, host Fiat-Shamir), 3 600 steps each.
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r6; mkdir -p $O
show() { python - $1 "$2" <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
s = j["step_latency_ms"]
print("%-10s %7.1f k  median %.2f p99 %.2f max %.2f  cores %.2f" % (sys.argv[2], j["proofs_per_s"] / 1e3, s["median"], s["p99"], s["max"], j["host_cpu_cores_busy"]))
PY
}
for rep in 1 2 3 4; do
  (cd _prev && examples/throughput --steps 3600 --warmup 180) > $O/ab_prev_$rep.json 2> $O/ab.err || { tail -5 $O/ab.err; exit 1; }; show $O/ab_prev_$rep.json before
  examples/throughput --steps 3600 --warmup 180 > $O/ab_new_$rep.json 2> $O/ab.err || { tail -5 $O/ab.err; exit 1; }; show $O/ab_new_$rep.json after
done
