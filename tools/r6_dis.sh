#!/bin/bash
# k_disassemble_fields without the windows the reference never reads (and k_lincomb_stream with the padding-row slots clamped):
# the verifier suites, kernel times in a cohort of six alone, PMC bytes.
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r6; mkdir -p $O
python -m pytest tests/test_gpu_00_kernels.py tests/test_gpu_02_verify.py tests/test_gpu_04_configs.py tests/test_gpu_07_api_paths.py -m gpu -x -q 2>&1 | tail -3 || exit 1
BUSY_STEPS=60 BUSY_ARGS="--slots 6 --combine 6" tools/gpu_busy.sh gpurun_out/prof/busy6 40 > $O/dis_busy.txt 2>&1 || exit 1
grep -E "lincomb|assemble|gather_frags|check_|interp_setup|coef_limbs|pow_table|GPU busy" $O/dis_busy.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/lc_f -- python3 tools/pmc_workload.py > /dev/null 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/lc_w -- python3 tools/pmc_workload.py > /dev/null 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, os, collections
def load(d, name):
    f = max(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != name: continue
        k = (r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size"]))
        a = acc.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
    return acc
F, W = load("gpurun_out/prof/lc_f", "FETCH_SIZE"), load("gpurun_out/prof/lc_w", "WRITE_SIZE")
for k, (n, v) in F.items():
    if "lincomb_stream" in k[0] or "assemble" in k[0]:
        w = W.get(k, [1, 0.0])
        print(k, "dispatches", n, "FETCH_KB %.0f WRITE_KB %.0f  HBM MB (2F+W, decimal) %.1f" % (v / n, w[1] / w[0], (2 * v / n + w[1] / w[0]) * 1024 / 1e6))
PY
