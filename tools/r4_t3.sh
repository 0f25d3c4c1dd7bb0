set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
for cfg in "9 3" "8 4"; do
  set -- $cfg
  echo "== slots $1 combine $2" >> gpurun_out/r4/trace3.txt
  KOSK_COMBINE_TRACE=1 timeout -k 10 300 python bench.py --gpus 1 --slots $1 --combine $2 --steps 240 --warmup 24 --no-kernels --no-cpu-baseline --phase-stats 2>gpurun_out/r4/trace3.err | python -c "
import sys,json
j=json.loads(sys.stdin.readline())
print(json.dumps({'value':j['value'],'lat':j['step_latency_ms']['median'],'phase':j['phase_means_ms']}))
" >> gpurun_out/r4/trace3.txt
  grep "kosk combine" gpurun_out/r4/trace3.err >> gpurun_out/r4/trace3.txt
done
cat gpurun_out/r4/trace3.txt
