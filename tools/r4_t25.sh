cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4/t25_gputest.log 2>&1; rc=$?; tail -4 gpurun_out/r4/t25_gputest.log; [ $rc -eq 0 ] || exit $rc
BUSY_ARGS="--slots 3 --combine 3" bash tools/gpu_busy.sh gpurun_out/r4/busy25 12 > gpurun_out/r4/t25_busy.txt 2>&1; cat gpurun_out/r4/t25_busy.txt | head -14
cd $GRAFT_REPO_ROOT
python3 tools/trace_gaps.py $(ls gpurun_out/r4/busy25/*/*kernel_trace.csv | head -1) > gpurun_out/r4/t25_gaps.txt 2>&1; grep -E "steps of|sum of gaps|rocclr" gpurun_out/r4/t25_gaps.txt
O=gpurun_out/r4/sweep25.txt; rm -f $O
run() { echo "== $1" >> $O; shift
  timeout -k 10 300 python bench.py --gpus 1 --no-kernels --no-cpu-baseline "$@" 2>>gpurun_out/r4/sweep25.err | python -c "
import sys,json
j=json.loads(sys.stdin.readline())
print(json.dumps({'value':round(j['value']),'lat':round(j['step_latency_ms']['median'],2),'p90':round(j['step_latency_ms']['p90'],2),'frac':round((j['roofline'] or {}).get('frac'),4),'cores':j['host_cpu_cores_busy']}))
" >> $O
}
for i in 1 2 3; do
KOSK_SMALL_COPY_KERNEL=0 run "hipMemcpyAsync for the small copies #$i" --steps 360 --warmup 36
run "copy kernel (default) #$i" --steps 360 --warmup 36
done
cat $O
