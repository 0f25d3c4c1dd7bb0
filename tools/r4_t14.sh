cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4; rm -f gpurun_out/r4/sweep14.txt
run() { # label, args..., env via caller
  echo "== $1" >> gpurun_out/r4/sweep14.txt; shift
  timeout -k 10 300 python bench.py --gpus 1 --no-kernels --no-cpu-baseline "$@" 2>>gpurun_out/r4/sweep14.err | python -c "
import sys,json
j=json.loads(sys.stdin.readline())
print(json.dumps({'value':round(j['value']),'lat':round(j['step_latency_ms']['median'],2),'p90':round(j['step_latency_ms']['p90'],2),'frac':round((j['roofline'] or {}).get('frac'),4),'hv':round(j['kernels_in_pipeline']['hash_view']['avg_us'],1)}))
" >> gpurun_out/r4/sweep14.txt
}
run "9 handles / 3 cohorts, default queues" --steps 360 --warmup 36
for q in 4 5 6 8; do
  GPU_MAX_HW_QUEUES=$q run "12 handles / 4 cohorts, GPU_MAX_HW_QUEUES=$q" --steps 480 --warmup 48 --slots 12
done
for q in 5 6; do
  GPU_MAX_HW_QUEUES=$q run "9 handles / 3 cohorts, GPU_MAX_HW_QUEUES=$q" --steps 360 --warmup 36
  GPU_MAX_HW_QUEUES=$q run "15 handles / 5 cohorts, GPU_MAX_HW_QUEUES=$q" --steps 600 --warmup 60 --slots 15
done
GPU_MAX_HW_QUEUES=3 run "9 handles / 3 cohorts, GPU_MAX_HW_QUEUES=3" --steps 360 --warmup 36
GPU_MAX_HW_QUEUES=2 run "6 handles / 2 cohorts, GPU_MAX_HW_QUEUES=2" --steps 240 --warmup 24 --slots 6
cat gpurun_out/r4/sweep14.txt
