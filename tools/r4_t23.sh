cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
for p in 0 1; do echo "== KOSK_TG_PIPE=$p"; KOSK_TG_STAMP=1 KOSK_TG_PIPE=$p timeout -k 10 120 python3 tools/gemm_time.py 9982 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r4/stamps23.txt 2>&1
cat gpurun_out/r4/stamps23.txt | cut -c1-600
