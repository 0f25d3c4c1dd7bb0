# expansion product alone: pipelined epilogue (KOSK_TG_PIPE) and 16-byte stores (KOSK_TG_STORE16); kernel time from rocprofv3 --kernel-trace --stats
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4/sweep22.txt; mkdir -p gpurun_out/r4; rm -f $O
timeout -k 10 600 python -m pytest tests/test_gpu_00_kernels.py tests/test_gpu_01_prover.py -x -q > gpurun_out/r4/t22_tests.log 2>&1; echo "kernel tests rc=$?" >> $O; tail -3 gpurun_out/r4/t22_tests.log >> $O
for n in 9982 29946; do
for cfg in "1 1" "0 0" "1 0" "0 1" "1 1" "0 0"; do
  set -- $cfg
  rm -rf gpurun_out/r4/p22
  KOSK_TG_PIPE=$1 KOSK_TG_STORE16=$2 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4/p22 -- python3 tools/gemm_time.py $n > gpurun_out/r4/p22.log 2>&1
  f=$(find gpurun_out/r4/p22 -name "*kernel_stats.csv" | head -1)
  python3 - $f $n "pipe=$1 store16=$2" >> $O <<PY
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_table_gemm" in r["Name"]:
        print("n=%s %s: %s calls, avg %.2f us, min %.2f us  %s" % (sys.argv[2], sys.argv[3], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, r["Name"][:40]))
PY
done; done
cat $O
