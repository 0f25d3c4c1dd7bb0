#!/bin/bash
# bench.py in device mode: the arrangement it now picks (sixteen callers per cohort, one worker per caller) and what a rank with two / three / four
# usable cores gets (taskset: usable_host_cores() follows the affinity mask)
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r6f; mkdir -p $O
show() { python - $1 "$2" <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); s = j["step_latency_ms"]; c = j["config"]
print("%-44s %7.1f k  drained %.1f k  lat %.2f/%.2f ms  cores %.2f  fs=%s callers=%s per cohort=%s in flight=%s" % (sys.argv[2], j["value"] / 1e3, j["drained_run"]["value"] / 1e3, s["median"], s["p99"],
      j["host_cpu_cores_busy"], c.get("fiat_shamir", "")[:6], c.get("caller_threads") or c.get("slots"), c.get("handles_per_cohort") or c.get("combine"), j.get("proofs_in_flight_per_gpu")))
PY
}
python bench.py --fs device --steps 1920 --warmup 96 --no-kernels --no-cpu-baseline > $O/devb_full.json 2> $O/devb.err || { tail -5 $O/devb.err; exit 1; }; show $O/devb_full.json "--fs device (all cores)"
for n in 2 3 4 6; do
  taskset -c 0-$((n-1)) python bench.py --steps 960 --warmup 96 --no-kernels --no-cpu-baseline > $O/devb_$n.json 2> $O/devb.err || { tail -5 $O/devb.err; exit 1; }; show $O/devb_$n.json "taskset $n cores (auto mode)"
done
taskset -c 0-1 python bench.py --steps 20 --warmup 5 --no-kernels --no-cpu-baseline > $O/devb_2_short.json 2> $O/devb.err || { tail -5 $O/devb.err; exit 1; }; show $O/devb_2_short.json "taskset 2 cores, driver's flags"
