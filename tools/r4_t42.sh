cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
run() { echo "== $1"; shift
  timeout -k 10 300 python bench.py --gpus 1 --no-kernels --no-cpu-baseline --phase-stats "$@" 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.readline())
p=j['phase_means_ms']
print(json.dumps({'value':round(j['value']),'lat':round(j['step_latency_ms']['median'],2),'cores':j['host_cpu_cores_busy'],'phases':{k:round(v,4) for k,v in p.items()}}))
"
}
run "one cohort alone (3 callers)" --steps 360 --warmup 36 --slots 3 --combine 3
run "three cohorts (default)" --steps 1200 --warmup 120
KOSK_POOL_SPIN_US=300 run "one cohort alone, pool workers spin 300 us" --steps 360 --warmup 36 --slots 3 --combine 3
