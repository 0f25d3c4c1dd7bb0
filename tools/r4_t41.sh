cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
for i in 1 2 3; do t0=$(date +%s); timeout -k 10 900 python bench.py > gpurun_out/r4/t41_$i.json 2> gpurun_out/r4/t41_$i.err; echo "bench rc $? in $(( $(date +%s) - t0 )) s"; python3 - $i <<PY
import json,sys
j=json.loads(open("gpurun_out/r4/t41_%s.json"%sys.argv[1]).read().strip().splitlines()[-1])
print(round(j["value"]), j["steps"], round(j["ms_per_step"],4), round(j["roofline"]["frac"],4), round(j["step_latency_ms"]["median"],2), j["uncombined"].get("proofs_per_s"), j["cohorts_of_five"].get("proofs_per_s"), j["drop_in"].get("proofs_per_s"), j["cpu_baseline"]["value"])
PY
done
