#!/bin/bash
# counters of the view-hash dispatches: back to back (dispatches 3-8) vs right behind an expansion product (9-14)
mkdir -p gpurun_out/r2/hcp
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_IFETCH SQ_INSTS_VALU GRBM_GUI_ACTIVE" "SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM"; do
  i=$((i+1)); rm -rf gpurun_out/r2/hcp/p$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/r2/hcp/p$i -- python3 tools/hash_clock_wl.py > gpurun_out/r2/hcp/p$i.log 2>&1
  python3 - gpurun_out/r2/hcp/p$i <<'PY'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not fs: print("no counters collected in", sys.argv[1]); sys.exit(0)
rows = [r for r in csv.DictReader(open(fs[0])) if "k_commit_hash" in r["Kernel_Name"]]
by = collections.OrderedDict()
for r in rows: by.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for name, v in by.items():
    print("%-22s back to back: %s | behind the product: %s" % (name, " ".join("%.0f" % x for x in v[2:8]), " ".join("%.0f" % x for x in v[8:14])))
PY
done
