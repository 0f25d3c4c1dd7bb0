mkdir -p gpurun_out/r2
for cfg_slots in "4 3" "4 4" "4 6" "5 3" "5 4" "3 6" "3 8"; do
set -- $cfg_slots
python bench.py --config $1 --slots $2 --steps 120 --warmup 12 --no-kernels --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('config', $1, 'slots', $2, round(d['value']), round(d['ms_per_step'],3), d['roofline']['frac'], d['step_latency_ms']['median'])"
done
