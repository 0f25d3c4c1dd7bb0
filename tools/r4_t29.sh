cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_02_verify.py tests/test_gpu_04_configs.py tests/test_gpu_06_compact.py tests/test_gpu_10_combine.py -x -q > gpurun_out/r4/t29_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r4/t29_tests.log; [ $rc -eq 0 ] || exit $rc
BUSY_ARGS="--slots 3 --combine 3" bash tools/gpu_busy.sh gpurun_out/r4/busy29 40 > gpurun_out/r4/t29_busy.txt 2>&1; grep -E "disassemble|assemble|GPU busy" gpurun_out/r4/t29_busy.txt
cd $GRAFT_REPO_ROOT
python3 tools/trace_gaps.py $(ls gpurun_out/r4/busy29/*/*kernel_trace.csv | head -1) > gpurun_out/r4/t29_gaps.txt 2>&1; grep -E "steps of|sum of" gpurun_out/r4/t29_gaps.txt
