#!/bin/bash
# SQ counters of the expansion product alone (what its waves wait on). Not product code.
# usage (GPU box, repo root): tools/gemm_pmc.sh <outdir> [rows]
out=$1; n=${2:-9982}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $out -- python3 tools/gemm_time.py $n > /dev/null 2>&1
python3 - $(find $out -name "*counter_collection.csv" | head -1) <<PY
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kosk::", "")
    if "gemm" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    print(k)
    for name, v in sorted(c.items()): print("   %-28s %14.0f  (median of %d dispatches)" % (name, sorted(v)[len(v)//2], len(v)))
PY
