set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4; rm -f gpurun_out/r4/sweep7.txt
run() { # label, env...
  echo "== $1" >> gpurun_out/r4/sweep7.txt; shift
  env "$@" timeout -k 10 300 python bench.py --gpus 1 --steps 360 --warmup 36 --no-kernels --no-cpu-baseline 2>>gpurun_out/r4/sweep7.err | python -c "
import sys,json
j=json.loads(sys.stdin.readline())
print(json.dumps({'value':round(j['value']),'lat':round(j['step_latency_ms']['median'],2),'frac':round((j['roofline'] or {}).get('frac'),4),'kern':{k:round(v['avg_us'],1) for k,v in j['kernels_in_pipeline'].items()}}))
" >> gpurun_out/r4/sweep7.txt
}
run "prev lib" KOSK_LIB_PATH=$PWD/mpcith_kyber_kosk_amd/libkosk_prev.so
run "new lib, split tables" X=1
run "new lib, whole tables" KOSK_VERIFY_TABLES=1
run "prev lib again" KOSK_LIB_PATH=$PWD/mpcith_kyber_kosk_amd/libkosk_prev.so
run "new lib, split tables again" X=1
cat gpurun_out/r4/sweep7.txt
