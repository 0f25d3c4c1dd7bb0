#!/bin/bash
# GPU-busy time per bench step (sum of kernel durations, 1 slot) -- the stable optimisation metric.
# usage (on the GPU box, from the repo root): [BUSY_ARGS="--slots 3 --combine 3"] tools/gpu_busy.sh [outdir] [rows]
# BUSY_ARGS: the slots of the run (default one handle on its own; "--slots 3 --combine 3" = one cohort, every launch serves 138 proofs);
# a 'step' in the output is one verifier run (one k_opened_setup launch), whatever number of callers it served
out=${1:-gpurun_out/busy}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --steps ${BUSY_STEPS:-8} --warmup 2 ${BUSY_ARGS:---slots 1 --combine 1} --no-cpu-baseline --no-kernels > /dev/null 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - $f <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
steps=[int(r["Calls"]) for r in rows if "k_opened_setup" in r["Name"]][0]  # one k_opened_setup launch per verify = per step
for r in rows[:int("${2:-16}")]:
    print("%-40s calls/step %5.1f  avg %7.1f us  per-step %6.1f us"%(r["Name"].replace("void ","").replace("kosk::","").replace("(anonymous namespace)::","").split("(")[0][:44], int(r["Calls"])/steps, float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e3/steps))
print("GPU busy per step: %.0f us over %d steps"%(tot/1e3/steps, steps))
PY
