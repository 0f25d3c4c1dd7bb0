#!/bin/bash
# Issue-side accounting of one pipeline step (GPU box, repo root): SQ counters per kernel over a 1-slot bench run, summed per step.
# Tells what the kernels of a step need of the SIMDs' vector issue / matrix pipe, independent of how they overlap in the pipeline.
# usage: tools/pipeline_pmc.sh <outdir>
out=${1:-gpurun_out/pmc_pipe}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out -- python3 bench.py --steps 8 --warmup 2 --slots 1 --no-cpu-baseline --no-kernels > /dev/null 2>&1
python3 - $(find $out -name "*counter_collection.csv" | head -1) <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kosk::", "")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"], k)
    if key not in seen:
        seen.add(key); calls[k] += 1
steps = max(1, calls.get("k_opened_setup", 1))
names = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE"]
print("per step (sum over the step's dispatches), %d steps; counters as reported (quad-cycles for SQ_*_CYCLES / ACTIVE_INST_*, see MI355X_MICROARCH.md)" % steps)
print("%-34s %6s " % ("kernel", "calls") + " ".join("%14s" % n.replace("SQ_", "").replace("GRBM_", "")[:14] for n in names))
tot = collections.defaultdict(float)
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0)):
    print("%-34s %6.1f " % (k[:34], calls[k] / steps) + " ".join("%14.0f" % (c.get(n, 0) / steps) for n in names))
    for n in names: tot[n] += c.get(n, 0) / steps
print("%-34s %6s " % ("TOTAL", "") + " ".join("%14.0f" % tot[n] for n in names))
PY
