cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4; rm -rf gpurun_out/nttpmc*
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" "GRBM_GUI_ACTIVE SQ_CYCLES SQ_LEVEL_WAVES SQ_INSTS_VALU_ADD_F16"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/nttpmc$i -- python3 tools/ntt_time.py > /dev/null 2>&1
done
python3 - <<'PY' > gpurun_out/r4/ntt_pmc.txt
import csv, glob, collections
acc = collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/nttpmc*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "k_ntt256" not in r["Kernel_Name"] or int(r["Grid_Size"]) != 65536 * 16: continue
        a = acc.setdefault(r["Counter_Name"], [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
print("k_ntt256 at 65 536 polynomials (grid 1 048 576 threads), per dispatch, summed over the device:")
for k, (n, v) in acc.items():
    print("%-28s %16.0f   (%d dispatches)" % (k, v / n, n))
PY
cat gpurun_out/r4/ntt_pmc.txt
