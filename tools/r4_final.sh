cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
rm -rf gpurun_out/r4f; mkdir -p gpurun_out/r4f
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r4f/r04_gputest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r4f/r04_gputest.log; tail -3 gpurun_out/r4f/r04_gputest.log
timeout -k 10 900 python bench.py > gpurun_out/r4f/r04_bench_default.json 2> gpurun_out/r4f/bench_default.err; echo "bench exit $?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r4f/r04_bench_driver_flags.json 2>/dev/null; echo "bench (driver flags) exit $?"
bash tools/make_profiles.sh r04 > gpurun_out/r4f/make_profiles.log 2>&1; tail -8 gpurun_out/r4f/make_profiles.log
cp gpurun_out/prof/r04_* gpurun_out/prof/traffic.json gpurun_out/r4f/ 2>/dev/null
ls gpurun_out/r4f
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ovl; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ovl -- python3 bench.py --steps 120 --warmup 12 --no-cpu-baseline --no-kernels > /dev/null 2>&1
python3 tools/trace_overlap.py $(find gpurun_out/ovl -name "*kernel_trace.csv" | head -1) > gpurun_out/r4f/r04_trace_overlap.txt 2>&1; head -8 gpurun_out/r4f/r04_trace_overlap.txt
