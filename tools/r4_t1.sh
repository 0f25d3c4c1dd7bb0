set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_gpu_10_combine.py -x -q > gpurun_out/r4/t_combine.log 2>&1 || { tail -60 gpurun_out/r4/t_combine.log; exit 1; }
tail -5 gpurun_out/r4/t_combine.log
timeout -k 10 300 python bench.py --gpus 1 --steps 100 --warmup 10 --no-kernels --no-cpu-baseline > gpurun_out/r4/b1.json 2> gpurun_out/r4/b1.err || { tail -30 gpurun_out/r4/b1.err; exit 1; }
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r4/b1.json').read().strip().splitlines()[-1])
print({k:j[k] for k in ('value','ms_per_step','step_latency_ms','combining')})
print(j['roofline'])
print({k:(round(v['avg_us'],1),round(v['proofs_per_launch'],1)) for k,v in j['kernels_in_pipeline'].items()})
PY
