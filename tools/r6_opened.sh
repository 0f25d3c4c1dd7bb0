#!/bin/bash
# review item 4: the opened parties' records by recomputation (k_opened_gemm + k_assemble_groups) against the gathering form, one binary
set -o pipefail
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r6; mkdir -p $O
python -m pytest tests/test_gpu_01_prover.py tests/test_gpu_02_verify.py tests/test_gpu_04_configs.py tests/test_gpu_10_combine.py -m gpu -x -q 2>&1 | tail -3 || exit 1
KOSK_OREC=0 python -m pytest tests/test_gpu_01_prover.py tests/test_gpu_04_configs.py -m gpu -x -q 2>&1 | tail -3 || exit 1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for orec in 1 0 1 0; do
  rm -rf gpurun_out/prof/asmdbg
  KOSK_OREC=$orec rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/asmdbg -- python3 tools/asm_ablate.py > /dev/null 2>&1 || exit 1
  f=$(find gpurun_out/prof/asmdbg -name "*kernel_stats.csv" | head -1)
  python3 - $f $orec <<'PY'
import csv, sys
out = []
for r in csv.DictReader(open(sys.argv[1])):
    if "k_assemble_groups" in r["Name"] or "k_opened_gemm" in r["Name"]:
        out.append("%s avg %.1f us (%s calls)" % (r["Name"].split("(")[0].replace("kosk::", ""), float(r["AverageNs"]) / 1e3, r["Calls"]))
print("orec=%s  " % sys.argv[2] + "   ".join(sorted(out)))
PY
done
for orec in 1 0; do
KOSK_OREC=$orec BUSY_STEPS=60 BUSY_ARGS="--slots 6 --combine 6" tools/gpu_busy.sh gpurun_out/prof/busy6 40 > $O/og_busy_$orec.txt 2>&1 || exit 1
echo "orec=$orec: $(grep -E 'assemble|opened_gemm|disassemble|GPU busy' $O/og_busy_$orec.txt | sed 's/  */ /g' | tr '\n' '|')"
done
