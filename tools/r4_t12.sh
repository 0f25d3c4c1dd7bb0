cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4; rm -f gpurun_out/r4/sweep12.txt
timeout -k 10 300 python -m pytest tests/test_gpu_09_edges.py -q -k "knobs" 2>&1 | tail -2 >> gpurun_out/r4/sweep12.txt
run() { # label, args..., env via KOSK_*
  echo "== $1" >> gpurun_out/r4/sweep12.txt; shift
  timeout -k 10 300 python bench.py --gpus 1 --no-kernels --no-cpu-baseline "$@" 2>>gpurun_out/r4/sweep12.err | python -c "
import sys,json
j=json.loads(sys.stdin.readline())
print(json.dumps({'value':round(j['value']),'drained':round(j['drained_run']['value']),'lat':round(j['step_latency_ms']['median'],2),'p90':round(j['step_latency_ms']['p90'],2),'frac':round((j['roofline'] or {}).get('frac'),4),'hv':round(j['kernels_in_pipeline']['hash_view']['avg_us'],1),'cores':j['host_cpu_cores_busy']}))
" >> gpurun_out/r4/sweep12.txt
}
run "9 handles, 3 cohorts, own streams (default)" --steps 360 --warmup 36
KOSK_SHARED_STREAMS=3 run "18 handles, 6 cohorts on 3 shared streams" --steps 720 --warmup 72 --slots 18
KOSK_SHARED_STREAMS=3 run "12 handles, 4 cohorts on 3 shared streams" --steps 480 --warmup 48 --slots 12
KOSK_SHARED_STREAMS=3 run "9 handles, 3 cohorts on 3 shared streams" --steps 360 --warmup 36
KOSK_SHARED_STREAMS=2 run "12 handles, 4 cohorts on 2 shared streams" --steps 480 --warmup 48 --slots 12
KOSK_SHARED_STREAMS=3 run "12 handles, 6 cohorts of 2 on 3 shared streams" --steps 480 --warmup 48 --slots 12 --combine 2
KOSK_SHARED_STREAMS=3 run "27 handles, 9 cohorts on 3 shared streams" --steps 1080 --warmup 108 --slots 27
run "9 handles, 3 cohorts, own streams (again)" --steps 360 --warmup 36
cat gpurun_out/r4/sweep12.txt
