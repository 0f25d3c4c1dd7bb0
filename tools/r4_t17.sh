cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4; rm -f gpurun_out/r4/sweep17.txt
lscpu | grep -E "Model name|Socket|Core|Thread|Flags" | cut -c1-400 | sed 's/Flags.*avx512f.*/Flags: ... avx512f present/' >> gpurun_out/r4/sweep17.txt
python - >> gpurun_out/r4/sweep17.txt <<'PY'
import ctypes as C, time, os, sys
sys.path.insert(0, ".")
from mpcith_kyber_kosk_amd import api
lib = api.lib
n, L = 8, 1454 * 32
buf = os.urandom(n * L); out = C.create_string_buffer(32 * n)
for env in (None, "4", "1"):
    # width is a process static: measure through separate interpreters below instead; here only the default
    pass
t0 = time.perf_counter()
for _ in range(200):
    w = lib.kosk_host_sha3_256_multi(out, buf, L, L, n, 1)
dt = (time.perf_counter() - t0) / 200
print("host multi-buffer sha3_256, %d messages of %d B on one thread: width %d, %.1f us per call (%.3f us per Keccak-f of the chain)" % (n, L, w, dt * 1e6, dt * 1e6 / 343))
n = 4
t0 = time.perf_counter()
for _ in range(200):
    w = lib.kosk_host_sha3_256_multi(out, buf, L, L, n, 1)
dt = (time.perf_counter() - t0) / 200
print("  same with 4 messages: width %d, %.1f us per call" % (w, dt * 1e6))
PY
for w in 8 4; do KOSK_FS_WIDTH=$w python - >> gpurun_out/r4/sweep17.txt <<'PY'
import ctypes as C, time, os, sys
sys.path.insert(0, ".")
from mpcith_kyber_kosk_amd import api
lib = api.lib
n, L = int(os.environ["KOSK_FS_WIDTH"]), 1454 * 32
buf = os.urandom(n * L); out = C.create_string_buffer(32 * n)
# through the batched Fiat-Shamir path (honours KOSK_FS_WIDTH): kosk_host_sha3_256_multi with nthreads > 1 uses groups of the forced width
t0 = time.perf_counter()
for _ in range(200):
    w = lib.kosk_host_sha3_256_multi(out, buf, L, L, n, 1)
dt = (time.perf_counter() - t0) / 200
print("KOSK_FS_WIDTH=%s: %d messages, one thread: %.1f us per call" % (os.environ["KOSK_FS_WIDTH"], n, dt * 1e6))
PY
done
run() { # label, args..., env via KOSK_*
  echo "== $1" >> gpurun_out/r4/sweep17.txt; shift
  timeout -k 10 300 python bench.py --gpus 1 --no-kernels --no-cpu-baseline --phase-stats "$@" 2>>gpurun_out/r4/sweep17.err | python -c "
import sys,json
j=json.loads(sys.stdin.readline())
p=j['phase_means_ms']
print(json.dumps({'value':round(j['value']),'lat':round(j['step_latency_ms']['median'],2),'frac':round((j['roofline'] or {}).get('frac'),4),'fs':[p['fs_alpha_host'],p['fs_open_host'],p['v_fs_alpha_host'],p['v_fs_open_host_and_masks']],'cores':j['host_cpu_cores_busy'],'threads':j['config']['host_threads_per_slot']}))
" >> gpurun_out/r4/sweep17.txt
}
run "default (width auto, 6 threads per handle)" --steps 360 --warmup 36
KOSK_FS_WIDTH=4 KOSK_HOST_THREADS=12 run "KOSK_FS_WIDTH=4, 12 threads per handle" --steps 360 --warmup 36
KOSK_FS_WIDTH=4 KOSK_HOST_THREADS=6 run "KOSK_FS_WIDTH=4, 6 threads per handle" --steps 360 --warmup 36
KOSK_HOST_THREADS=12 run "width auto, 12 threads per handle" --steps 360 --warmup 36
run "default again" --steps 360 --warmup 36
KOSK_FS_WIDTH=4 KOSK_HOST_THREADS=12 run "KOSK_FS_WIDTH=4, 12 threads per handle (again)" --steps 360 --warmup 36
cat gpurun_out/r4/sweep17.txt
