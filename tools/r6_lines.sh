#!/bin/bash
# the two lines of record once more on this box (box-to-box spread of the final tree)
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r6f; mkdir -p $O
tag=${1:-x}
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/line_driver_$tag.json 2> $O/line.err || { tail -5 $O/line.err; exit 1; }
python bench.py --no-cpu-baseline > $O/line_default_$tag.json 2> $O/line.err || { tail -5 $O/line.err; exit 1; }
python - $O/line_driver_$tag.json $O/line_default_$tag.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    j = json.loads([l for l in open(f) if l.startswith("{")][-1]); s = j["step_latency_ms"]
    print("%-28s %.1f k drained %.1f k  lat %.2f/%.2f/%.2f  cores %.2f  native %.1f k  frac %.4f" % (f.split("/")[-1], j["value"] / 1e3, j["drained_run"]["value"] / 1e3, s["median"], s["p99"], s["max"],
          j["host_cpu_cores_busy"], (j.get("native_callers") or {}).get("proofs_per_s", 0) / 1e3, j["roofline"]["frac"]))
PY
