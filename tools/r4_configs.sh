cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4; rm -f gpurun_out/r4/configs.txt
for c in 2 4 5; do
  echo "== config $c" >> gpurun_out/r4/configs.txt
  timeout -k 10 500 python bench.py --config $c --steps $([ $c = 5 ] && echo 24 || echo 200) --warmup $([ $c = 5 ] && echo 4 || echo 20) --no-kernels --no-cpu-baseline 2>>gpurun_out/r4/configs.err | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(json.dumps({'metric':j['metric'],'value':round(j['value']),'ms_per_step':round(j['ms_per_step'],4),'lat':round(j['step_latency_ms']['median'],2),'frac':round((j['roofline'] or {}).get('frac'),4),'slots':j['config']['pipeline_slots_per_gpu'],'cohort':j['config']['handles_per_cohort'],'comb':(j.get('combining') or {}).get('mean_callers_per_run'),'latency':j.get('latency'),'gather':(j.get('digest_allgather') or {}).get('collective')}))
" >> gpurun_out/r4/configs.txt
done
cat gpurun_out/r4/configs.txt
