cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4; rm -f gpurun_out/r4/sweep15.txt
KOSK_TG_NREG=1 timeout -k 10 600 python -m pytest tests/test_gpu_00_kernels.py tests/test_gpu_01_prover.py tests/test_gpu_02_verify.py -x -q 2>&1 | tail -2 >> gpurun_out/r4/sweep15.txt
KOSK_TG_NREG=2 timeout -k 10 600 python -m pytest tests/test_gpu_00_kernels.py tests/test_gpu_01_prover.py -x -q 2>&1 | tail -2 >> gpurun_out/r4/sweep15.txt
for nr in 0 1 2; do
  echo "== KOSK_TG_NREG=$nr one handle alone (46 proofs per launch)" >> gpurun_out/r4/sweep15.txt
  KOSK_TG_NREG=$nr BUSY_STEPS=30 bash tools/gpu_busy.sh gpurun_out/busyrb 40 2>&1 | grep -E "table_gemm<7|GPU busy" >> gpurun_out/r4/sweep15.txt
  echo "== KOSK_TG_NREG=$nr one cohort alone (138 proofs per launch)" >> gpurun_out/r4/sweep15.txt
  KOSK_TG_NREG=$nr BUSY_STEPS=60 BUSY_ARGS="--slots 3 --combine 3" bash tools/gpu_busy.sh gpurun_out/busyrb 40 2>&1 | grep -E "table_gemm<7|GPU busy" >> gpurun_out/r4/sweep15.txt
done
run() { # label, args..., env via KOSK_*
  echo "== $1" >> gpurun_out/r4/sweep15.txt; shift
  timeout -k 10 300 python bench.py --gpus 1 --no-kernels --no-cpu-baseline "$@" 2>>gpurun_out/r4/sweep15.err | python -c "
import sys,json
j=json.loads(sys.stdin.readline())
k=j['kernels_in_pipeline']
print(json.dumps({'value':round(j['value']),'lat':round(j['step_latency_ms']['median'],2),'frac':round((j['roofline'] or {}).get('frac'),4),'gemm1':round(k['gemm_expand1']['avg_us'],1),'v_recon':round(k['v_gemm_recon']['avg_us'],1)}))
" >> gpurun_out/r4/sweep15.txt
}
for nr in 0 1 2 0 1 2; do KOSK_TG_NREG=$nr run "bench default, KOSK_TG_NREG=$nr" --steps 360 --warmup 36; done
cat gpurun_out/r4/sweep15.txt
