mkdir -p gpurun_out/r2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "0 0" "1 0" "1 1" "0 1"; do
set -- $cfg
export KOSK_HASH_PRIMER=$1 KOSK_HASH_SPLIT=$2
rm -rf gpurun_out/r2/ab
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2/ab -- python3 bench.py --steps 8 --warmup 2 --slots 1 --no-cpu-baseline --no-kernels > /dev/null 2>&1
python3 - $1 $2 <<'PY'
import csv,glob,collections,sys
f=glob.glob("gpurun_out/r2/ab/**/*kernel_trace.csv",recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_commit_hash" in r["Kernel_Name"] or "primer" in r["Kernel_Name"]:
        d[(r["Kernel_Name"].split("(")[0][-28:], r["Grid_Size_X"])].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
print("primer", sys.argv[1], "split", sys.argv[2], " | ".join("%s %s: %.1f" % (k[0], k[1], sum(v)/len(v)) for k,v in d.items()))
PY
for rep in 1 2; do python bench.py --steps 400 --warmup 40 --no-kernels --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('   6 slots:', round(d['value']), round(d['ms_per_step'],4), 'frac', round(r['frac'],4), 'view us', round(r['avg_launch_us'],1))"; done
done
