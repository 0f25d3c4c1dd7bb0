#!/bin/bash
# Round 5: what a rank of an 8-GPU job does on its host share -- bench.py's scarce-cores branch (sleeping waits, three Fiat-Shamir workers per
# caller, no pre-wake spinning) forced on a one-GPU box, beside the default; then the two-rank gloo rehearsal of the default configuration
# with its default eighteen callers per rank (both ranks on this one GPU).   usage: tools/r5_n8_host.sh <outfile>
out=${1:-gpurun_out/r5/n8_host.txt}; mkdir -p $(dirname $out); : > $out
run() { name=$1; shift
  j=$(timeout -k 5 150 env "$@" 2>/dev/null | tail -1)
  python3 - "$name" "$j" >> $out <<'PY'
import json, sys
try:
    j = json.loads(sys.argv[2]); l = j["step_latency_ms"]
    print("%-44s n_gpus %d %8.0f proofs/s drained %8.0f | latency ms median %.2f p99 %.2f | cores %.2f | waits %s threads %s" % (
          sys.argv[1], j["n_gpus"], j["value"], j["drained_run"]["value"], l["median"], l["p99"], j["host_cpu_cores_busy"], j["config"]["host_waits"], j["config"]["host_threads_per_slot"]))
except Exception as e:
    print("%-44s failed: %r" % (sys.argv[1], e))
PY
  tail -1 $out; }
B="python bench.py --steps 1800 --warmup 180 --no-kernels --no-cpu-baseline"
for rep in 1 2; do
run "default (spin + nap, 4 workers per caller)" X=1 $B
run "as a rank of eight (sleep, 3 workers)" KOSK_BLOCKING_SYNC=1 KOSK_HOST_THREADS=3 $B
done
port=$((20000 + RANDOM % 20000))
run "two ranks on this GPU over gloo (rehearsal)" KOSK_BENCH_REHEARSE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 2 --steps 360 --warmup 72
