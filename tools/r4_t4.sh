set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4; rm -f gpurun_out/r4/sweep4.txt
for cfg in "9 3" "12 3" "8 2" "12 4" "16 4" "6 2" "10 2" "15 5"; do
  set -- $cfg
  echo "== slots $1 combine $2" >> gpurun_out/r4/sweep4.txt
  timeout -k 10 300 python bench.py --gpus 1 --slots $1 --combine $2 --steps 360 --warmup 36 --no-kernels --no-cpu-baseline 2>>gpurun_out/r4/sweep4.err | python -c "
import sys,json
j=json.loads(sys.stdin.readline())
print(json.dumps({'value':round(j['value']),'drained':round(j['drained_run']['value']),'lat':round(j['step_latency_ms']['median'],2),'p90':round(j['step_latency_ms']['p90'],2),'frac':round((j['roofline'] or {}).get('frac'),4),'hv':round(j['kernels_in_pipeline'].get('hash_view',{}).get('avg_us'),1),'ppl':j['kernels_in_pipeline'].get('hash_view',{}).get('proofs_per_launch'),'cores':j['host_cpu_cores_busy'],'comb':round(j['combining']['mean_callers_per_run'],2)}))
" >> gpurun_out/r4/sweep4.txt
done
cat gpurun_out/r4/sweep4.txt
