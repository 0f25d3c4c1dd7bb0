cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4/gputest_b.log 2>&1; echo "pytest exit $?" >> gpurun_out/r4/gputest_b.log; tail -4 gpurun_out/r4/gputest_b.log
timeout -k 10 900 python bench.py > gpurun_out/r4/bench_default.json 2> gpurun_out/r4/bench_default.err; echo "bench exit $?"
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r4/bench_default.json').read().strip().splitlines()[-1])
print({k:j[k] for k in ('value','ms_per_step','steps')}, j['step_latency_ms']['median'], j['roofline']['frac'], j['combining'])
print({k:(v.get('frac_hbm_peak'), v.get('us')) for k,v in j['kernels_65536_lanes'].items()} if 'error' not in j['kernels_65536_lanes'] else j['kernels_65536_lanes'])
print({k:v for k,v in j['drop_in'].items() if k in ('proofs_per_s','one_thread_pageable_image','one_thread_pinned_image','two_threads_pinned_image','two_threads_pinned_compact','error')})
print(j.get('uncombined'))
print(j.get('cpu_baseline'))
PY
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-kernels --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.readline()); print('driver flags:', round(j['value']), j['ms_per_step'], j['roofline']['frac'], round(j['drained_run']['value']))"
