# expansion product alone, waves of a SIMD in opposite phases (KOSK_TG_PHASE): kernel time from rocprofv3 --kernel-trace --stats
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4/sweep21.txt; mkdir -p gpurun_out/r4; rm -f $O
for n in 9982 29946; do
for ph in 0 1 2 3 4 6 8 -3 0; do
  rm -rf gpurun_out/r4/p21
  KOSK_TG_PHASE=$ph rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4/p21 -- python3 tools/gemm_time.py $n > gpurun_out/r4/p21.log 2>&1
  f=$(find gpurun_out/r4/p21 -name "*kernel_stats.csv" | head -1)
  python3 - $f $n $ph >> $O <<PY
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_table_gemm" in r["Name"]:
        print("n=%s KOSK_TG_PHASE=%s: %s calls, avg %.2f us, min %.2f us" % (sys.argv[2], sys.argv[3], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
done; done
cat $O
