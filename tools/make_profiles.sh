#!/bin/bash
# Regenerates the files under profiles/ (run on the MI355X box from the repo root; outputs under gpurun_out/prof).
# usage: tools/make_profiles.sh <tag>        e.g. r01b
set -o pipefail
tag=${1:-r05}
out=gpurun_out/prof
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# 1. kernel stats of the bench command itself
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-kernels > $out/bench_under_rocprof.log 2>&1
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/${tag}_bench_kernel_stats.csv
grep -a '^{"metric"' $out/bench_under_rocprof.log | tail -1 > $out/${tag}_bench_under_rocprof.json
echo "stats done"
# 2. one slot: GPU-busy per step
tools/gpu_busy.sh $out/busy 80 > $out/${tag}_gpu_busy_1slot.txt 2>&1
BUSY_STEPS=60 BUSY_ARGS="--slots 6 --combine 6" tools/gpu_busy.sh $out/busy6 80 > $out/${tag}_gpu_busy_1cohort.txt 2>&1
BUSY_STEPS=60 BUSY_ARGS="--slots 3 --combine 3" tools/gpu_busy.sh $out/busy3 80 > $out/${tag}_gpu_busy_1cohort_of_three.txt 2>&1
echo "busy done"
# 3. PMC passes (separate, no other tracing domains)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_f -- python3 tools/pmc_workload.py > /dev/null 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_w -- python3 tools/pmc_workload.py > /dev/null 2>&1
echo "write done"
python3 - $out $tag <<'PY'
import csv, sys, glob, collections, json, os
out, tag = sys.argv[1], sys.argv[2]
def load(d, name):
    f = max(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != name: continue
        k = (r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size"]))
        a = acc.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
    return acc
F, W = load(out + "/pmc_f", "FETCH_SIZE"), load(out + "/pmc_w", "WRITE_SIZE")
with open("%s/%s_pmc_fetch_write.csv" % (out, tag), "w", newline="") as fo:
    wr = csv.writer(fo)  # kernel names carry commas (template arguments): quoted
    wr.writerow(["kernel", "grid_threads", "dispatches", "FETCH_SIZE_KB_avg", "WRITE_SIZE_KB_avg", "HBM_MB_corrected_2F_plus_W"])
    for k, (n, v) in F.items():
        w = W.get(k, [1, 0.0])
        f_kb, w_kb = v / n, w[1] / max(1, w[0])
        wr.writerow([k[0], k[1], n, "%.0f" % f_kb, "%.0f" % w_kb, "%.1f" % ((2 * f_kb + w_kb) / 1024)])
# in-pipeline view hash: the k_commit_hash_dma<16,220,...> dispatch of the 276-proof merged run (1472 threads per proof); the
# 65 536-lane calibration dispatch has a different grid
hv = [(k, v) for k, v in F.items() if "k_commit_hash" in k[0] and "<16, 220" in k[0] and k[1] % 1472 == 0 and k[1] // 1472 <= 276]
if hv:
    k, (n, v) = max(hv, key=lambda kv: kv[0][1])
    w = W[k]
    proofs = k[1] // 1472
    traffic = int(round((2 * v / n + w[1] / w[0]) * 1024))
    json.dump({"source": "profiles/%s_pmc_fetch_write.csv (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; FETCH_SIZE doubled per MI355X_MICROARCH.md, calibrated on k_ntt256)" % tag,
               "hash_view_lanes_per_launch": proofs * 1454, "hash_view_hbm_bytes_per_lane": traffic / (proofs * 1454.0),
               "hash_view_hbm_bytes_per_launch": traffic, "hash_view_algorithmic_bytes_per_launch": proofs * 1454 * 504}, open(out + "/traffic.json", "w"), indent=1)
    print("view hash traffic per launch:", traffic, "algorithmic", proofs * 1454 * 504, "proofs", proofs)
PY
echo "pmc done"
