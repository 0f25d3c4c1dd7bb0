#!/bin/bash
# CU-partition sweep of the headline pipeline (run on the MI355X box from the repo root): proofs/s, view-hash launch time and
# step latency for slots x partitions (KOSK_CU_PARTITION per slot, bench.py --partitions).
# usage: tools/partition_sweep.sh <outfile> [extra bench args]
out=${1:-/dev/stdout}; shift
run() { # layout partitions slots
    line=$(KOSK_CU_MASK_LAYOUT=$1 timeout -k 10 200 python3 bench.py --partitions $2 --slots $3 --steps 300 --warmup 30 --no-kernels --no-cpu-baseline "${@:4}" 2>/dev/null | grep -a '^{"metric"' | tail -1)
    python3 - "$1" "$2" "$3" "$line" <<'PY' >> "$out"
import json, sys
lay, p, s, line = sys.argv[1:5]
try:
    j = json.loads(line)
    hv = j["kernels_in_pipeline"].get("hash_view", {}).get("avg_us", 0)
    ht = j["kernels_in_pipeline"].get("hash_tcomm", {}).get("avg_us", 0)
    g1 = j["kernels_in_pipeline"].get("gemm_expand1", {}).get("avg_us", 0)
    print("layout %s partitions %s slots %2s : %8.0f proofs/s  drained %8.0f  ms/step %.4f  hash_view %.1f us  hash_tcomm %.1f us  expand1 %.1f us  latency %.2f ms  frac %.4f"
          % (lay, p, s, j["value"], j["drained_run"]["value"], j["ms_per_step"], hv, ht, g1, j["step_latency_ms"]["median"], (j.get("roofline") or {}).get("frac") or 0))
except Exception as e:
    print("layout %s partitions %s slots %s : FAILED %s" % (lay, p, s, e))
PY
}
for cfg in "0 1 6" "0 2 6" "0 2 8" "0 4 8" "0 4 12" "1 4 8" "0 8 8" "0 8 16" "1 8 16" "1 2 8"; do run $cfg "$@"; done
