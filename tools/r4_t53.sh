cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
for a in "" "--steps 20 --warmup 5"; do timeout -k 10 600 python bench.py $a 2>/dev/null | python -c "
import sys,json
s=[l for l in sys.stdin if l.startswith('{')]
j=json.loads(s[-1]); print(len(s), round(j['value']), j['steps'], round(j['roofline']['frac'],4), round(j['step_latency_ms']['median'],2), j['uncombined'].get('proofs_per_s'), j['cohorts_of_five'].get('proofs_per_s'), j['drop_in'].get('proofs_per_s'), j['cpu_baseline']['value'])"; done
