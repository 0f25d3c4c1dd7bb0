cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_01_prover.py tests/test_gpu_02_verify.py tests/test_gpu_04_configs.py tests/test_gpu_06_compact.py tests/test_gpu_08_sizes.py tests/test_gpu_10_combine.py -x -q > gpurun_out/r4/t50_tests.log 2>&1; rc=$?; tail -2 gpurun_out/r4/t50_tests.log; [ $rc -eq 0 ] || { grep -v "^  File\|amdgpu.ids" gpurun_out/r4/t50_tests.log | tail -50; exit $rc; }
BUSY_ARGS="--slots 3 --combine 3" bash tools/gpu_busy.sh gpurun_out/r4/busy50 12 > gpurun_out/r4/t50_busy.txt 2>&1; grep -E "assemble|GPU busy" gpurun_out/r4/t50_busy.txt
cd $GRAFT_REPO_ROOT
bash tools/gpu_busy.sh gpurun_out/r4/busy50b 12 > gpurun_out/r4/t50_busy1.txt 2>&1; grep -E "assemble|GPU busy" gpurun_out/r4/t50_busy1.txt
