cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4; rm -f gpurun_out/r4/sweep19.txt
run() { # label, args..., env via KOSK_*
  echo "== $1" >> gpurun_out/r4/sweep19.txt; shift
  timeout -k 10 300 python bench.py --gpus 1 --no-kernels --no-cpu-baseline "$@" 2>>gpurun_out/r4/sweep19.err | python -c "
import sys,json
j=json.loads(sys.stdin.readline())
print(json.dumps({'value':round(j['value']),'lat':round(j['step_latency_ms']['median'],2),'p90':round(j['step_latency_ms']['p90'],2),'frac':round((j['roofline'] or {}).get('frac'),4),'cores':j['host_cpu_cores_busy'],'waits':j['config']['host_waits']}))
" >> gpurun_out/r4/sweep19.txt
}
run "default 9/3 (spin 20 us, spinning waits)" --steps 360 --warmup 36
KOSK_POOL_SPIN_US=0 run "9/3, KOSK_POOL_SPIN_US=0" --steps 360 --warmup 36
KOSK_BLOCKING_SYNC=1 run "9/3, KOSK_BLOCKING_SYNC=1" --steps 360 --warmup 36
KOSK_POOL_SPIN_US=0 KOSK_BLOCKING_SYNC=1 run "9/3, spin 0 + blocking" --steps 360 --warmup 36
KOSK_POOL_SPIN_US=0 KOSK_BLOCKING_SYNC=1 run "15/5, spin 0 + blocking" --steps 600 --warmup 60 --slots 15 --combine 5
run "15/5 default" --steps 600 --warmup 60 --slots 15 --combine 5
KOSK_POOL_SPIN_US=0 KOSK_BLOCKING_SYNC=1 run "18/6, spin 0 + blocking" --steps 720 --warmup 72 --slots 18 --combine 6
run "18/6 default" --steps 720 --warmup 72 --slots 18 --combine 6
run "default 9/3 again" --steps 360 --warmup 36
KOSK_POOL_SPIN_US=0 run "9/3, KOSK_POOL_SPIN_US=0 again" --steps 360 --warmup 36
cat gpurun_out/r4/sweep19.txt
