#!/bin/bash
# round 6: the latency tail of 3 600-step runs -- the round-5 tree against this one on ONE box, then this tree's native loop at 1 800 / 3 600 / 7 200 steps
set -o pipefail
O=gpurun_out/r6f
mkdir -p $O
B="bench.py --no-kernels --no-cpu-baseline --steps 3600 --warmup 180"
show() { python - $1 <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
s = j.get("step_latency_ms")
print("%-28s %7.1f k  median %.2f p99 %.2f max %.2f  cores %s" % (sys.argv[1].split("/")[-1], (j.get("value") or j.get("proofs_per_s")) / 1e3, s["median"], s["p99"], s["max"], j["host_cpu_cores_busy"]))
PY
}
for i in 1 2; do
  (cd _r5tree && python $B) > $O/tail_r5_$i.json 2> $O/tail.err || { tail -5 $O/tail.err; exit 1; }; show $O/tail_r5_$i.json
  python $B > $O/tail_r6_$i.json 2> $O/tail.err || { tail -5 $O/tail.err; exit 1; }; show $O/tail_r6_$i.json
done
for st in 1800 3600 7200; do
  examples/throughput --steps $st --warmup 180 > $O/tail_native_$st.json 2> $O/tail.err || exit 1; show $O/tail_native_$st.json
done
KOSK_WAIT_NAP=0 examples/throughput --steps 3600 --warmup 180 > $O/tail_native_3600_nonap.json 2> $O/tail.err || exit 1; show $O/tail_native_3600_nonap.json
examples/throughput --steps 3600 --warmup 180 --blocking 1 > $O/tail_native_3600_blocking.json 2> $O/tail.err || exit 1; show $O/tail_native_3600_blocking.json
