#!/bin/bash
# A/B of the commitment-hash launch split (KOSK_HASH_SPLIT) on one slot: per-launch durations from a rocprofv3 kernel trace
mkdir -p gpurun_out/r2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for sp in 0 1; do
export KOSK_HASH_SPLIT=$sp
rm -rf gpurun_out/r2/hs$sp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2/hs$sp -- python3 bench.py --steps 8 --warmup 2 --slots 1 --no-cpu-baseline --no-kernels > /dev/null 2>&1
python3 - $sp <<'PY'
import csv,glob,collections,sys
f=glob.glob("gpurun_out/r2/hs%s/**/*kernel_trace.csv"%sys.argv[1],recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_commit_hash" in r["Kernel_Name"]:
        d[(r["Kernel_Name"].split("(")[0][-28:], r["Grid_Size_X"], r["Grid_Size_Y"])].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in d.items(): print("split", sys.argv[1], k, len(v), round(sum(v)/len(v),1))
PY
done
