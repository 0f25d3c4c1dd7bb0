set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4; rm -f gpurun_out/r4/sweep8.txt
run() { # label, env...
  echo "== $1" >> gpurun_out/r4/sweep8.txt; shift
  env "$@" timeout -k 10 300 python bench.py --gpus 1 --steps 360 --warmup 36 --no-kernels --no-cpu-baseline 2>>gpurun_out/r4/sweep8.err | python -c "
import sys,json
j=json.loads(sys.stdin.readline())
print(json.dumps({'value':round(j['value']),'lat':round(j['step_latency_ms']['median'],2),'frac':round((j['roofline'] or {}).get('frac'),4)}))
" >> gpurun_out/r4/sweep8.txt
}
run "prev lib" KOSK_LIB_PATH=$PWD/mpcith_kyber_kosk_amd/libkosk_prev.so
run "new lib, whole tables (no side stream)" KOSK_VERIFY_TABLES=1
run "new lib, split tables (null stream)" X=1
run "new lib, split tables (own side stream)" KOSK_SIDE_STREAM=1
run "new lib, split tables (null stream) again" X=1
run "new lib, whole tables again" KOSK_VERIFY_TABLES=1
run "prev lib again" KOSK_LIB_PATH=$PWD/mpcith_kyber_kosk_amd/libkosk_prev.so
cat gpurun_out/r4/sweep8.txt
