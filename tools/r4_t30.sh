cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_02_verify.py tests/test_gpu_03_split_api.py tests/test_gpu_04_configs.py tests/test_gpu_05_compat_example.py tests/test_gpu_06_compact.py tests/test_gpu_07_api_paths.py tests/test_gpu_09_edges.py tests/test_gpu_10_combine.py -x -q > gpurun_out/r4/t30_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r4/t30_tests.log; [ $rc -eq 0 ] || exit $rc
BUSY_ARGS="--slots 3 --combine 3" bash tools/gpu_busy.sh gpurun_out/r4/busy30 40 > gpurun_out/r4/t30_busy.txt 2>&1; grep -E "disassemble|gen_matrix|decode_pk|GPU busy" gpurun_out/r4/t30_busy.txt
cd $GRAFT_REPO_ROOT
python3 tools/trace_gaps.py $(ls gpurun_out/r4/busy30/*/*kernel_trace.csv | head -1) > gpurun_out/r4/t30_gaps.txt 2>&1; grep -E "steps of|sum of" gpurun_out/r4/t30_gaps.txt
