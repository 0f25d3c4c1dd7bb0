cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
KOSK_QUEUE_AHEAD=1 timeout -k 10 600 python -m pytest tests/test_gpu_01_prover.py tests/test_gpu_03_split_api.py tests/test_gpu_10_combine.py tests/test_gpu_09_edges.py -x -q > gpurun_out/r4/t46_tests.log 2>&1; rc=$?; tail -2 gpurun_out/r4/t46_tests.log; [ $rc -eq 0 ] || { grep -v "^  File\|amdgpu.ids" gpurun_out/r4/t46_tests.log | tail -50; exit $rc; }
O=gpurun_out/r4/sweep46.txt; rm -f $O
run() { echo "== $1" >> $O; shift
  timeout -k 10 300 python bench.py --gpus 1 --no-kernels --no-cpu-baseline "$@" 2>>gpurun_out/r4/sweep46.err | python -c "
import sys,json
j=json.loads(sys.stdin.readline())
print(json.dumps({'value':round(j['value']),'lat':round(j['step_latency_ms']['median'],2),'p90':round(j['step_latency_ms']['p90'],2),'in_keygen':round(j['step_latency_ms']['mean_in_keygen_call'],3),'in_verify':round(j['step_latency_ms']['mean_in_verify_call'],3),'cores':j['host_cpu_cores_busy']}))
" >> $O
}
for i in 1 2 3 4; do
run "launches issued after each host round (default) #$i" --steps 720 --warmup 72
KOSK_QUEUE_AHEAD=1 run "prover's later launches queued ahead behind gates #$i" --steps 720 --warmup 72
done
cat $O
