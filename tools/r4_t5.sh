set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4; rm -f gpurun_out/r4/sweep5.txt
timeout -k 10 600 python -m pytest tests/test_gpu_00_kernels.py tests/test_gpu_01_prover.py tests/test_gpu_10_combine.py tests/test_gpu_08_sizes.py -x -q > gpurun_out/r4/t5.log 2>&1 || { tail -40 gpurun_out/r4/t5.log; exit 1; }
tail -2 gpurun_out/r4/t5.log
for rep in 1 2; do
for dd in 1 0; do
  echo "== digest_direct $dd (slots 9 combine 3)" >> gpurun_out/r4/sweep5.txt
  KOSK_DIGEST_DIRECT=$dd timeout -k 10 300 python bench.py --gpus 1 --steps 360 --warmup 36 --no-kernels --no-cpu-baseline --phase-stats 2>>gpurun_out/r4/sweep5.err | python -c "
import sys,json
j=json.loads(sys.stdin.readline())
print(json.dumps({'value':round(j['value']),'drained':round(j['drained_run']['value']),'lat':round(j['step_latency_ms']['median'],2),'frac':round((j['roofline'] or {}).get('frac'),4),'hv':round(j['kernels_in_pipeline'].get('hash_view',{}).get('avg_us'),1),'ht':round(j['kernels_in_pipeline'].get('hash_tcomm',{}).get('avg_us'),1),'phase':j['phase_means_ms']}))
" >> gpurun_out/r4/sweep5.txt
done
done
echo "== uncombined 6 slots direct 1 / 0" >> gpurun_out/r4/sweep5.txt
for dd in 1 0; do
  KOSK_DIGEST_DIRECT=$dd timeout -k 10 300 python bench.py --gpus 1 --slots 6 --combine 1 --steps 360 --warmup 36 --no-kernels --no-cpu-baseline 2>>gpurun_out/r4/sweep5.err | python -c "
import sys,json
j=json.loads(sys.stdin.readline())
print(json.dumps({'value':round(j['value']),'lat':round(j['step_latency_ms']['median'],2),'frac':round((j['roofline'] or {}).get('frac'),4),'hv':round(j['kernels_in_pipeline'].get('hash_view',{}).get('avg_us'),1)}))
" >> gpurun_out/r4/sweep5.txt
done
cat gpurun_out/r4/sweep5.txt
