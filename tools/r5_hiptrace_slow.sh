#!/bin/bash
# Round 5: which HIP API call blocks during the rare process-wide stalls?  rocprofv3 --hip-trace of tools/tail_probe.py, then every API call
# longer than 2 ms with its thread, start and duration.   usage: tools/r5_hiptrace_slow.sh <outfile> [steps per caller]
out=${1:-gpurun_out/r5/hiptrace_slow.txt}; n=${2:-120}; mkdir -p $(dirname $out)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/hiptrace; 
rocprofv3 --hip-trace --output-format csv -d /tmp/hiptrace -- python3 tools/tail_probe.py $n > $out.probe 2>&1
grep -v "^t=" $out.probe | cut -c1-300
f=$(find /tmp/hiptrace -name "*hip_api_trace.csv" | head -1)
python3 - "$f" > $out <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print("# %d HIP API calls traced; columns of the trace: %s" % (len(rows), list(rows[0].keys())))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
slow = [r for r in rows if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 2_000_000]
slow.sort(key=lambda r: int(r["Start_Timestamp"]))
byname = collections.Counter(r["Function"] for r in slow)
print("# calls longer than 2 ms by function:", dict(byname))
for r in slow:
    print("t=%9.2f ms  %8.2f ms  tid %s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Thread_Id"), r["Function"]))
PY
head -60 $out
