cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_sharding.py tests/test_gpu_04_configs.py -m gpu -x -q > gpurun_out/r4/t52_tests.log 2>&1; rc=$?; tail -2 gpurun_out/r4/t52_tests.log; [ $rc -eq 0 ] || { tail -40 gpurun_out/r4/t52_tests.log; exit $rc; }
timeout -k 10 600 python bench.py > gpurun_out/r4/t52_bench.json 2>/dev/null; echo "bench rc $?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r4/t52_bench_driver.json 2>/dev/null; echo "bench (driver flags) rc $?"
for c in 2 4 5; do timeout -k 10 300 python bench.py --config $c --steps $([ $c = 5 ] && echo 24 || echo 200) --warmup $([ $c = 5 ] && echo 4 || echo 20) --no-kernels --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('config', $c, round(j['value']), round(j['ms_per_step'],4))"; done
python3 - <<PY
import json
for f in ("t52_bench.json","t52_bench_driver.json"):
    j=json.loads(open("gpurun_out/r4/"+f).read().strip().splitlines()[-1])
    print(f, round(j["value"]), j["steps"], round(j["roofline"]["frac"],4), round(j["step_latency_ms"]["median"],2), j["uncombined"].get("proofs_per_s"), j["cohorts_of_five"].get("proofs_per_s"), j["drop_in"].get("proofs_per_s"), j["cpu_baseline"]["value"])
PY
