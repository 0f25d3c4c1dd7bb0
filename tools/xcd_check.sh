#!/bin/bash
# full GPU suite, one-slot busy time, per-kernel HBM traffic (two PMC passes) and two bench runs: the check after a change of
# workgroup order
mkdir -p gpurun_out/r2/pmcx
python -m pytest tests -m gpu -x -q 2>&1 | tail -2 || exit 1
tools/gpu_busy.sh gpurun_out/r2/busy2 40 > gpurun_out/r2/busy2.txt 2>&1; grep -E "table_gemm|k_lincomb |assemble|GPU busy" gpurun_out/r2/busy2.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; rm -rf gpurun_out/r2/pmcx/*
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r2/pmcx/f -- python3 tools/pmc_workload.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r2/pmcx/w -- python3 tools/pmc_workload.py > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
acc={}
for d,n in (("f","FETCH_SIZE"),("w","WRITE_SIZE")):
    f=glob.glob("gpurun_out/r2/pmcx/%s/**/*counter_collection.csv"%d,recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]!=n: continue
        k=(r["Kernel_Name"].split("(")[0].replace("void kosk::",""), r["Grid_Size"])
        a=acc.setdefault(k,{"FETCH_SIZE":[0,0.0],"WRITE_SIZE":[0,0.0]})[n]; a[0]+=1; a[1]+=float(r["Counter_Value"])
tot=0
for k,v in acc.items():
    if "at::" in k[0] or "rows_copy" in k[0] or "fillBuffer" in k[0]: continue
    f=v["FETCH_SIZE"]; w=v["WRITE_SIZE"]
    if f[0]<2: continue
    mb=(2*f[1]/f[0]+w[1]/max(1,w[0]))/1024
    tot+=mb*f[0]/2
    if mb > 8 or "table_gemm" in k[0] or k[0].startswith("k_lincomb"): print(k, f[0], round(mb,1),"MB")
print("per-step total MB", round(tot))
PY
for i in 1 2; do python bench.py --steps 400 --warmup 40 --no-kernels --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],4))"; done
