cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4; rm -f gpurun_out/r4/sweep18.txt
run() { # label, args..., env via KOSK_*
  echo "== $1" >> gpurun_out/r4/sweep18.txt; shift
  timeout -k 10 300 python bench.py --gpus 1 --no-kernels --no-cpu-baseline --phase-stats "$@" 2>>gpurun_out/r4/sweep18.err | python -c "
import sys,json
j=json.loads(sys.stdin.readline())
p=j['phase_means_ms']
print(json.dumps({'value':round(j['value']),'lat':round(j['step_latency_ms']['median'],2),'frac':round((j['roofline'] or {}).get('frac'),4),'fs':[p['fs_alpha_host'],p['fs_open_host'],p['v_fs_alpha_host'],p['v_fs_open_host_and_masks']],'cores':j['host_cpu_cores_busy']}))
" >> gpurun_out/r4/sweep18.txt
}
for sp in 20 200 1000 3000 20 1000; do KOSK_POOL_SPIN_US=$sp run "KOSK_POOL_SPIN_US=$sp" --steps 360 --warmup 36; done
KOSK_POOL_SPIN_US=1000 run "uncombined 6 slots, KOSK_POOL_SPIN_US=1000" --steps 360 --warmup 36 --combine 1 --slots 6
KOSK_POOL_SPIN_US=20 run "uncombined 6 slots, KOSK_POOL_SPIN_US=20" --steps 360 --warmup 36 --combine 1 --slots 6
cat gpurun_out/r4/sweep18.txt
