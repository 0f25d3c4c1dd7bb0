# SQ counters of the wire-image kernels (k_assemble_fields, k_disassemble_fields) at 138 proofs per launch (tools/pmc_workload.py), separate passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4; rm -rf gpurun_out/wirepmc*
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_LEVEL_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/wirepmc$i -- python3 tools/pmc_workload.py > /dev/null 2>&1
done
python3 - <<'PY' > gpurun_out/r4/wire_pmc.txt
import csv, glob, collections
for kn in ("k_assemble_fields", "k_disassemble_fields"):
    acc = collections.OrderedDict()
    for f in sorted(glob.glob("gpurun_out/wirepmc*/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            if kn not in r["Kernel_Name"]: continue
            a = acc.setdefault(r["Counter_Name"], [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
    print("%s, 138 proofs per launch, per dispatch, summed over the device:" % kn)
    for k, (n, v) in acc.items():
        print("  %-28s %16.0f   (%d dispatches)" % (k, v / n, n))
PY
cat gpurun_out/r4/wire_pmc.txt
