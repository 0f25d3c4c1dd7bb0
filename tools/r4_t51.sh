cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
O=gpurun_out/r4/sweep51.txt; rm -f $O
run() { echo "== $1" >> $O; f=$2; shift; shift
  timeout -k 10 300 python $f --gpus 1 --no-kernels --no-cpu-baseline "$@" 2>>gpurun_out/r4/sweep51.err | python -c "
import sys,json
j=json.loads(sys.stdin.readline())
print(json.dumps({'value':round(j['value']),'lat':round(j['step_latency_ms']['median'],2),'p90':round(j['step_latency_ms']['p90'],2),'in_keygen':round(j['step_latency_ms']['mean_in_keygen_call'],3),'in_verify':round(j['step_latency_ms']['mean_in_verify_call'],3),'cores':j['host_cpu_cores_busy']}))
" >> $O
}
for i in 1 2 3 4; do
run "harness through the api.py wrappers #$i" tools/_ab/bench_old.py --steps 1200 --warmup 120
run "harness on the raw entry points with prebuilt arguments #$i" bench.py --steps 1200 --warmup 120
done
cat $O
