set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
for cfg in "46 6" "92 3" "92 4" "138 2" "138 3" "138 4" "184 2" "184 3" "276 2" "276 3" "368 2"; do
  set -- $cfg
  steps=$(( 9200 / $1 ))
  echo "== batch $1 slots $2 steps $steps" >> gpurun_out/r4/sweep1.txt
  python bench.py --gpus 1 --batch $1 --slots $2 --steps $steps --warmup 6 --no-kernels --no-cpu-baseline 2>>gpurun_out/r4/sweep1.err | python -c "
import sys,json
j=json.loads(sys.stdin.readline())
print(json.dumps({'value':j['value'],'ms_per_step':j['ms_per_step'],'lat':j['step_latency_ms']['median'],'frac':(j['roofline'] or {}).get('frac'),'hv':j['kernels_in_pipeline'].get('hash_view',{}).get('avg_us'),'cores':j['host_cpu_cores_busy']}))
" >> gpurun_out/r4/sweep1.txt
done
cat gpurun_out/r4/sweep1.txt
