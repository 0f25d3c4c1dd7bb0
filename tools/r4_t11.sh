cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4; rm -f gpurun_out/r4/sweep11.txt
timeout -k 10 600 python -m pytest tests/test_gpu_01_prover.py tests/test_gpu_02_verify.py tests/test_gpu_10_combine.py tests/test_gpu_04_configs.py -x -q 2>&1 | tail -2 >> gpurun_out/r4/sweep11.txt
run() { # label, args..., env via KOSK_*
  echo "== $1" >> gpurun_out/r4/sweep11.txt; shift
  timeout -k 10 300 python bench.py --gpus 1 --no-kernels --no-cpu-baseline "$@" 2>>gpurun_out/r4/sweep11.err | python -c "
import sys,json
j=json.loads(sys.stdin.readline())
print(json.dumps({'value':round(j['value']),'drained':round(j['drained_run']['value']),'lat':round(j['step_latency_ms']['median'],2),'frac':round((j['roofline'] or {}).get('frac'),4),'hv':round(j['kernels_in_pipeline']['hash_view']['avg_us'],1),'gemm1':round(j['kernels_in_pipeline']['gemm_expand1']['avg_us'],1),'lincomb':round(j['kernels_in_pipeline']['lincomb']['avg_us'],1)}))
" >> gpurun_out/r4/sweep11.txt
}
for w in 512 0 256 1024 128 2048 512 0; do KOSK_COPY_WAVES=$w run "KOSK_COPY_WAVES=$w" --steps 360 --warmup 36; done
KOSK_COPY_WAVES=512 run "uncombined 6 slots, KOSK_COPY_WAVES=512" --steps 360 --warmup 36 --combine 1 --slots 6
KOSK_COPY_WAVES=0 run "uncombined 6 slots, KOSK_COPY_WAVES=0" --steps 360 --warmup 36 --combine 1 --slots 6
cat gpurun_out/r4/sweep11.txt
