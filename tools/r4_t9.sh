cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4; rm -f gpurun_out/r4/sweep9.txt
timeout -k 5 120 python tools/ntt_time.py 2>&1 | grep -E "polys|Error|assert" >> gpurun_out/r4/sweep9.txt
timeout -k 10 300 python -m pytest tests/test_gpu_00_kernels.py tests/test_gpu_08_sizes.py tests/test_gpu_01_prover.py tests/test_gpu_02_verify.py -q -x 2>&1 | tail -2 >> gpurun_out/r4/sweep9.txt
run() { # label, args..., env via KOSK_*
  echo "== $1" >> gpurun_out/r4/sweep9.txt; shift
  timeout -k 10 300 python bench.py --gpus 1 --no-kernels --no-cpu-baseline "$@" 2>>gpurun_out/r4/sweep9.err | python -c "
import sys,json
j=json.loads(sys.stdin.readline())
print(json.dumps({'value':round(j['value']),'drained':round(j['drained_run']['value']),'steps':j['steps'],'lat':round(j['step_latency_ms']['median'],2),'frac':round((j['roofline'] or {}).get('frac'),4),'ppl':round(j['kernels_in_pipeline']['hash_view']['proofs_per_launch'],1),'gemm1':round(j['kernels_in_pipeline']['gemm_expand1']['avg_us'],1),'comb':round(j['combining']['mean_callers_per_run'],2) if j.get('combining') else None}))
" >> gpurun_out/r4/sweep9.txt
}
run "wide GEMM on, 360 steps" --steps 360 --warmup 36
KOSK_TG_WIDE=0 run "wide GEMM off, 360 steps" --steps 360 --warmup 36
run "wide GEMM on, 360 steps (again)" --steps 360 --warmup 36
KOSK_TG_WIDE=0 run "wide GEMM off, 360 steps (again)" --steps 360 --warmup 36
for i in 1 2 3 4; do run "driver flags #$i" --steps 20 --warmup 5; done
cat gpurun_out/r4/sweep9.txt
