cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4; rm -f gpurun_out/r4/sweep20.txt
O=gpurun_out/r4/sweep20.txt
echo "## harness alone on the box (18 threads, 138 proofs; 6 threads, 46 proofs), idle gap before every job" >> $O
for v in cv futex cv futex; do for idle in 50 300 1000; do echo "-- $v idle $idle" >> $O; tools/_ab/pool_wake_$v 18 138 $idle >> $O; done; done
for v in cv futex; do echo "-- $v 6 threads idle 300" >> $O; tools/_ab/pool_wake_$v 6 46 300 >> $O; done
run() { # label, args..., env via KOSK_*
  echo "== $1" >> $O; shift
  timeout -k 10 300 python bench.py --gpus 1 --no-kernels --no-cpu-baseline --phase-stats "$@" 2>>gpurun_out/r4/sweep20.err | python -c "
import sys,json
j=json.loads(sys.stdin.readline())
p=j['phase_means_ms']
print(json.dumps({'value':round(j['value']),'lat':round(j['step_latency_ms']['median'],2),'p90':round(j['step_latency_ms']['p90'],2),'frac':round((j['roofline'] or {}).get('frac'),4),'cores':j['host_cpu_cores_busy'],'fs':[p['fs_alpha_host'],p['fs_open_host'],p['v_fs_alpha_host'],p['v_fs_open_host_and_masks']]}))
" >> $O
}
for i in 1 2 3; do
KOSK_LIB_PATH=$PWD/tools/_ab/libkosk_cv.so run "cv pool 9/3 #$i" --steps 360 --warmup 36
run "futex pool 9/3 #$i" --steps 360 --warmup 36
done
KOSK_LIB_PATH=$PWD/tools/_ab/libkosk_cv.so run "cv pool uncombined 6" --steps 360 --warmup 36 --combine 1 --slots 6
run "futex pool uncombined 6" --steps 360 --warmup 36 --combine 1 --slots 6
KOSK_LIB_PATH=$PWD/tools/_ab/libkosk_cv.so run "cv pool 15/5" --steps 600 --warmup 60 --slots 15 --combine 5
run "futex pool 15/5" --steps 600 --warmup 60 --slots 15 --combine 5
cat $O
