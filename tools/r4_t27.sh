cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
O=gpurun_out/r4/sweep27.txt; rm -f $O
echo "## eight-way AVX-512 SHA3 of the host: generic template (old) against the in-place schedule (new), one core, 8 x 46 528 bytes in cache" >> $O
for i in 1 2 3; do echo -n "old: " >> $O; tools/_ab/bench_sha3_old >> $O; echo -n "new: " >> $O; tools/_ab/bench_sha3_new >> $O; done
run() { echo "== $1" >> $O; shift
  timeout -k 10 300 python bench.py --gpus 1 --no-kernels --no-cpu-baseline --phase-stats "$@" 2>>gpurun_out/r4/sweep27.err | python -c "
import sys,json
j=json.loads(sys.stdin.readline())
p=j['phase_means_ms']
print(json.dumps({'value':round(j['value']),'lat':round(j['step_latency_ms']['median'],2),'p90':round(j['step_latency_ms']['p90'],2),'frac':round((j['roofline'] or {}).get('frac'),4),'cores':j['host_cpu_cores_busy'],'fs':[p['fs_alpha_host'],p['fs_open_host'],p['v_fs_alpha_host'],p['v_fs_open_host_and_masks']]}))
" >> $O
}
for i in 1 2 3; do
KOSK_LIB_PATH=$PWD/tools/_ab/libkosk_old.so run "old SHA3 #$i" --steps 360 --warmup 36
run "new SHA3 #$i" --steps 360 --warmup 36
done
cat $O
