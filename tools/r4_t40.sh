cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_10_combine.py tests/test_gpu_01_prover.py -x -q > gpurun_out/r4/t40_tests.log 2>&1; rc=$?; tail -2 gpurun_out/r4/t40_tests.log; [ $rc -eq 0 ] || { tail -40 gpurun_out/r4/t40_tests.log; exit $rc; }
for pw in 0 400; do
KOSK_COMBINE_PREWAKE_US=$pw BUSY_ARGS="--slots 3 --combine 3" bash tools/gpu_busy.sh gpurun_out/r4/busy40_$pw 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
echo "prewake $pw: $(python3 tools/trace_gaps.py $(ls gpurun_out/r4/busy40_$pw/*/*kernel_trace.csv | head -1) | grep -E 'steps of|sum of|k_opened_setup' | tr '\n' ' ')"
done
O=gpurun_out/r4/sweep40.txt; rm -f $O
run() { echo "== $1" >> $O; shift
  timeout -k 10 300 python bench.py --gpus 1 --no-kernels --no-cpu-baseline "$@" 2>>gpurun_out/r4/sweep40.err | python -c "
import sys,json
j=json.loads(sys.stdin.readline())
print(json.dumps({'value':round(j['value']),'lat':round(j['step_latency_ms']['median'],2),'p90':round(j['step_latency_ms']['p90'],2),'frac':round((j['roofline'] or {}).get('frac'),4),'cores':j['host_cpu_cores_busy']}))
" >> $O
}
for i in 1 2 3 4; do
KOSK_COMBINE_PREWAKE_US=0 run "members sleep to the end of their run #$i" --steps 360 --warmup 36
run "members pre-woken for the end (400 us) #$i" --steps 360 --warmup 36
done
cat $O
