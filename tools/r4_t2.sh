set -e; rm -f gpurun_out/r4/sweep2.txt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_gpu_10_combine.py -q > gpurun_out/r4/t_combine.log 2>&1 || { grep -E "AssertionError|Error|FAILED" gpurun_out/r4/t_combine.log | head -20; }
tail -3 gpurun_out/r4/t_combine.log
for cfg in "9 3" "6 2" "12 4" "12 3" "8 2"; do
  set -- $cfg
  echo "== slots $1 combine $2" >> gpurun_out/r4/sweep2.txt
  timeout -k 10 300 python bench.py --gpus 1 --slots $1 --combine $2 --steps 240 --warmup 24 --no-kernels --no-cpu-baseline 2>>gpurun_out/r4/sweep2.err | python -c "
import sys,json
j=json.loads(sys.stdin.readline())
print(json.dumps({'value':j['value'],'ms_per_step':j['ms_per_step'],'lat':j['step_latency_ms'],'frac':(j['roofline'] or {}).get('frac'),'hv':j['kernels_in_pipeline'].get('hash_view',{}).get('avg_us'),'ppl':j['kernels_in_pipeline'].get('hash_view',{}).get('proofs_per_launch'),'cores':j['host_cpu_cores_busy'],'comb':j['combining'],'drained':j['drained_run']['value']}))
" >> gpurun_out/r4/sweep2.txt
done
cat gpurun_out/r4/sweep2.txt
