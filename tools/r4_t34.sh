cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4/t34_gputest.log 2>&1; rc=$?; tail -3 gpurun_out/r4/t34_gputest.log; [ $rc -eq 0 ] || { grep -v "^  File\|amdgpu.ids" gpurun_out/r4/t34_gputest.log | tail -60; exit $rc; }
timeout -k 10 900 python bench.py > gpurun_out/r4/t34_bench.json 2> gpurun_out/r4/t34_bench.err; echo "bench rc $?"
python3 - <<PY
import json
j=json.loads(open("gpurun_out/r4/t34_bench.json").read().strip().splitlines()[-1])
print(round(j["value"]), j["roofline"]["frac"], j["drop_in"].get("one_thread_pageable_image"), j["drop_in"].get("one_thread_pinned_image"), j["drop_in"].get("proofs_per_s"))
PY
