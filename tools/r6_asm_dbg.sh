#!/bin/bash
# what bounds k_assemble_groups?  ablations (results wrong by construction): rocprofv3 kernel stats of six 276-proof provers per switch
# bit 0: no opened groups, 1: no unopened groups, 2: no digest blocks, 3: unopened groups without their write-out, 4: without their gather loads
set -o pipefail
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r6; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for dbg in 0 1 2 4 8 16 24 3 7; do
  rm -rf gpurun_out/prof/asmdbg
  KOSK_ASM_DBG=$dbg rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/asmdbg -- python3 tools/asm_ablate.py > /dev/null 2>&1 || exit 1
  f=$(find gpurun_out/prof/asmdbg -name "*kernel_stats.csv" | head -1)
  python3 - $f $dbg <<'PY'
import csv, sys
out = []
for r in csv.DictReader(open(sys.argv[1])):
    if "k_assemble_groups" in r["Name"] or "k_opened_gemm" in r["Name"]:
        out.append("%s avg %.1f us (%s calls)" % (r["Name"].split("(")[0].replace("kosk::", ""), float(r["AverageNs"]) / 1e3, r["Calls"]))
print("dbg=%s  " % sys.argv[2] + "   ".join(sorted(out)))
PY
done
