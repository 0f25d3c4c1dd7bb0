cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
t0=$(date +%s); timeout -k 10 900 python bench.py > gpurun_out/r4/t35_bench.json 2> gpurun_out/r4/t35_bench.err; echo "bench rc $? in $(( $(date +%s) - t0 )) s"
python3 - <<PY
import json
s=open("gpurun_out/r4/t35_bench.json").read().strip().splitlines()
print(len(s), "line(s)")
j=json.loads(s[-1])
print(round(j["value"]), j["roofline"]["frac"], {k:(v if not isinstance(v,dict) else "...") for k,v in j["drop_in"].items()})
print(j["uncombined"].get("proofs_per_s"), j["cohorts_of_five"].get("proofs_per_s"), j["cpu_baseline"]["value"])
PY
