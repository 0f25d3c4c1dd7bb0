cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_01_prover.py tests/test_gpu_03_split_api.py tests/test_gpu_04_configs.py tests/test_gpu_07_api_paths.py tests/test_gpu_10_combine.py -x -q > gpurun_out/r4/t38_tests.log 2>&1; rc=$?; tail -2 gpurun_out/r4/t38_tests.log; [ $rc -eq 0 ] || { tail -40 gpurun_out/r4/t38_tests.log; exit $rc; }
BUSY_ARGS="--slots 3 --combine 3" bash tools/gpu_busy.sh gpurun_out/r4/busy38 3 > gpurun_out/r4/t38_busy.txt 2>&1; tail -1 gpurun_out/r4/t38_busy.txt
cd $GRAFT_REPO_ROOT
python3 tools/trace_gaps.py $(ls gpurun_out/r4/busy38/*/*kernel_trace.csv | head -1) > gpurun_out/r4/t38_gaps.txt 2>&1; grep -E "steps of|sum of|<<" gpurun_out/r4/t38_gaps.txt
O=gpurun_out/r4/sweep38.txt; rm -f $O
run() { echo "== $1" >> $O; shift
  timeout -k 10 300 python bench.py --gpus 1 --no-kernels --no-cpu-baseline "$@" 2>>gpurun_out/r4/sweep38.err | python -c "
import sys,json
j=json.loads(sys.stdin.readline())
print(json.dumps({'value':round(j['value']),'lat':round(j['step_latency_ms']['median'],2),'p90':round(j['step_latency_ms']['p90'],2),'frac':round((j['roofline'] or {}).get('frac'),4),'cores':j['host_cpu_cores_busy']}))
" >> $O
}
for i in 1 2 3; do
KOSK_LIB_PATH=$PWD/tools/_ab/libkosk_prev.so run "host half of keygen in front of the first host round #$i" --steps 360 --warmup 36
run "host half of keygen under the GPU's first phase #$i" --steps 360 --warmup 36
done
cat $O
