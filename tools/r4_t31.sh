cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_01_prover.py tests/test_gpu_02_verify.py -x -q > gpurun_out/r4/t31_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r4/t31_tests.log; [ $rc -eq 0 ] || exit $rc
BUSY_ARGS="--slots 3 --combine 3" bash tools/gpu_busy.sh gpurun_out/r4/busy31 40 > gpurun_out/r4/t31_busy.txt 2>&1; grep -E "disassemble|prover_pre|GPU busy" gpurun_out/r4/t31_busy.txt
cd $GRAFT_REPO_ROOT
python3 tools/trace_gaps.py $(ls gpurun_out/r4/busy31/*/*kernel_trace.csv | head -1) > gpurun_out/r4/t31_gaps.txt 2>&1; grep -E "steps of|sum of" gpurun_out/r4/t31_gaps.txt
