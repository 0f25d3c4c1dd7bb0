cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_00_kernels.py tests/test_gpu_01_prover.py tests/test_gpu_03_split_api.py tests/test_gpu_04_configs.py -x -q > gpurun_out/r4/t28_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r4/t28_tests.log; [ $rc -eq 0 ] || exit $rc
BUSY_ARGS="--slots 3 --combine 3" bash tools/gpu_busy.sh gpurun_out/r4/busy28 6 > gpurun_out/r4/t28_busy.txt 2>&1; head -8 gpurun_out/r4/t28_busy.txt; tail -1 gpurun_out/r4/t28_busy.txt
cd $GRAFT_REPO_ROOT
bash tools/gpu_busy.sh gpurun_out/r4/busy28b 4 > gpurun_out/r4/t28_busy1.txt 2>&1; grep -E "prover_pre|GPU busy" gpurun_out/r4/t28_busy1.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4/sweep28.txt; rm -f $O
run() { echo "== $1" >> $O; shift
  timeout -k 10 300 python bench.py --gpus 1 --no-kernels --no-cpu-baseline "$@" 2>>gpurun_out/r4/sweep28.err | python -c "
import sys,json
j=json.loads(sys.stdin.readline())
print(json.dumps({'value':round(j['value']),'lat':round(j['step_latency_ms']['median'],2),'p90':round(j['step_latency_ms']['p90'],2),'frac':round((j['roofline'] or {}).get('frac'),4),'cores':j['host_cpu_cores_busy']}))
" >> $O
}
for i in 1 2 3; do
KOSK_LIB_PATH=$PWD/tools/_ab/libkosk_prev.so run "roles in the order A B G N #$i" --steps 360 --warmup 36
run "roles in the order G N A B #$i" --steps 360 --warmup 36
done
cat $O
