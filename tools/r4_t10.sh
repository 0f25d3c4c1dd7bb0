cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4; rm -f gpurun_out/r4/sweep10.txt
run() { # label, args..., env via KOSK_*
  echo "== $1" >> gpurun_out/r4/sweep10.txt; shift
  timeout -k 10 300 python bench.py --gpus 1 --no-kernels --no-cpu-baseline "$@" 2>>gpurun_out/r4/sweep10.err | python -c "
import sys,json
j=json.loads(sys.stdin.readline())
print(json.dumps({'value':round(j['value']),'drained':round(j['drained_run']['value']),'steps':j['steps'],'lat':round(j['step_latency_ms']['median'],2),'frac':round((j['roofline'] or {}).get('frac'),4),'ms_per_step':round(j['ms_per_step'],4)}))
" >> gpurun_out/r4/sweep10.txt
}
run "360 steps" --steps 360 --warmup 36
for i in 1 2 3 4 5 6; do run "driver flags #$i" --steps 20 --warmup 5; done
run "360 steps again" --steps 360 --warmup 36
run "uncombined driver flags" --steps 20 --warmup 5 --combine 1 --slots 6
cat gpurun_out/r4/sweep10.txt
