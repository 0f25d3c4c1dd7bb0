cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4/t24_gputest.log 2>&1; rc=$?; tail -4 gpurun_out/r4/t24_gputest.log; [ $rc -eq 0 ] || exit $rc
BUSY_ARGS="--slots 3 --combine 3" bash tools/gpu_busy.sh gpurun_out/r4/busy24 40 > gpurun_out/r4/t24_busy.txt 2>&1; cat gpurun_out/r4/t24_busy.txt | head -24
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python bench.py > gpurun_out/r4/t24_bench.json 2> gpurun_out/r4/t24_bench.err; python - <<PY
import json
j=json.loads(open("gpurun_out/r4/t24_bench.json").read().strip().splitlines()[-1])
print({k:j.get(k) for k in ("value","ms_per_step","roofline","uncombined","cohorts_of_five")})
print(j.get("kernels_65536_lanes"))
PY
