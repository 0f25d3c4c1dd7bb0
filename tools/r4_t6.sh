set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4; rm -f gpurun_out/r4/sweep6.txt
timeout -k 10 900 python -m pytest tests/test_gpu_02_verify.py tests/test_gpu_10_combine.py tests/test_gpu_04_configs.py tests/test_gpu_07_api_paths.py tests/test_gpu_09_edges.py -x -q > gpurun_out/r4/t6.log 2>&1 || { tail -40 gpurun_out/r4/t6.log; exit 1; }
tail -2 gpurun_out/r4/t6.log
for rep in 1 2; do
for vt in 0 1; do
  echo "== KOSK_VERIFY_TABLES=$vt (slots 9 combine 3)" >> gpurun_out/r4/sweep6.txt
  KOSK_VERIFY_TABLES=$vt timeout -k 10 300 python bench.py --gpus 1 --steps 360 --warmup 36 --no-kernels --no-cpu-baseline --phase-stats 2>>gpurun_out/r4/sweep6.err | python -c "
import sys,json
j=json.loads(sys.stdin.readline())
print(json.dumps({'value':round(j['value']),'drained':round(j['drained_run']['value']),'lat':round(j['step_latency_ms']['median'],2),'frac':round((j['roofline'] or {}).get('frac'),4),'phase':j['phase_means_ms']}))
" >> gpurun_out/r4/sweep6.txt
done
done
for vt in 0 1; do
  echo "== uncombined 6 slots KOSK_VERIFY_TABLES=$vt" >> gpurun_out/r4/sweep6.txt
  KOSK_VERIFY_TABLES=$vt timeout -k 10 300 python bench.py --gpus 1 --slots 6 --combine 1 --steps 360 --warmup 36 --no-kernels --no-cpu-baseline 2>>gpurun_out/r4/sweep6.err | python -c "
import sys,json
j=json.loads(sys.stdin.readline())
print(json.dumps({'value':round(j['value']),'lat':round(j['step_latency_ms']['median'],2),'frac':round((j['roofline'] or {}).get('frac'),4)}))
" >> gpurun_out/r4/sweep6.txt
done
cat gpurun_out/r4/sweep6.txt
