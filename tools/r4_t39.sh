cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4/t39_gputest.log 2>&1; rc=$?; tail -2 gpurun_out/r4/t39_gputest.log; [ $rc -eq 0 ] || { grep -v "^  File\|amdgpu.ids" gpurun_out/r4/t39_gputest.log | tail -60; exit $rc; }
BUSY_ARGS="--slots 3 --combine 3" bash tools/gpu_busy.sh gpurun_out/r4/busy39 3 > gpurun_out/r4/t39_busy.txt 2>&1; tail -1 gpurun_out/r4/t39_busy.txt
cd $GRAFT_REPO_ROOT
python3 tools/trace_gaps.py $(ls gpurun_out/r4/busy39/*/*kernel_trace.csv | head -1) > gpurun_out/r4/t39_gaps.txt 2>&1; grep -E "steps of|sum of|<<" gpurun_out/r4/t39_gaps.txt
for i in 1 2 3; do timeout -k 10 300 python bench.py --gpus 1 --no-kernels --no-cpu-baseline --steps 360 --warmup 36 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.readline())
print(json.dumps({'value':round(j['value']),'lat':round(j['step_latency_ms']['median'],2),'p90':round(j['step_latency_ms']['p90'],2),'in_keygen':round(j['step_latency_ms']['mean_in_keygen_call'],3),'in_verify':round(j['step_latency_ms']['mean_in_verify_call'],3)}))
"; done
