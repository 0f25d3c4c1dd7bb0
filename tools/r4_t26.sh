# which default: 9 callers in cohorts of 3, or 15 in cohorts of 5 -- with the DRIVER's flags (--steps 20 --warmup 5) and with the bench's own defaults
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
O=gpurun_out/r4/sweep26.txt; rm -f $O
run() { echo "== $1" >> $O; shift
  t0=$(date +%s.%N); timeout -k 10 300 python bench.py --gpus 1 --no-kernels --no-cpu-baseline "$@" 2>>gpurun_out/r4/sweep26.err | python -c "
import sys,json
j=json.loads(sys.stdin.readline())
print(json.dumps({'value':round(j['value']),'steps':j['steps'],'ms_per_step':round(j['ms_per_step'],4),'lat':round(j['step_latency_ms']['median'],2),'p90':round(j['step_latency_ms']['p90'],2),'frac':round((j['roofline'] or {}).get('frac'),4),'cores':j['host_cpu_cores_busy'],'drained':round(j['drained_run']['value'])}))
" >> $O; python3 -c "import time,sys; print(\"%.1f s wall\" % (time.time() - float(sys.argv[1])))" $t0 >> $O
}
for i in 1 2 3 4 5; do
run "9/3 driver flags #$i" --steps 20 --warmup 5
run "15/5 driver flags #$i" --steps 20 --warmup 5 --slots 15 --combine 5
done
run "9/3 default" 
run "15/5 600 steps" --steps 600 --warmup 60 --slots 15 --combine 5
cat $O
