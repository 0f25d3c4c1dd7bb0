#!/bin/bash
# Kernel time of the expansion product alone (rocprofv3 kernel trace; not product code).
# usage (GPU box, repo root): tools/gemm_prof.sh <outdir> [rows ...]
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for n in "${@:-9982}"; do
  rm -rf $out/g$n
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/g$n -- python3 tools/gemm_time.py $n > /dev/null 2>&1
  f=$(find $out/g$n -name "*kernel_stats.csv" | head -1)
  python3 - $f $n <<PY
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm" in r["Name"] or "limbs" in r["Name"]:
        print("rows %6s  %-44s calls %3s avg %7.1f us" % (sys.argv[2], r["Name"].split("(")[0].replace("void ","").replace("kosk::","")[:44], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
