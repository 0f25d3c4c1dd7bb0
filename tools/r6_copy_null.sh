#!/bin/bash
# round 6: the four digest-table copies of a step on the device's legacy null stream (KOSK_COPY_NULL=1) against the cohort's own stream, alternating
set -o pipefail
O=gpurun_out/r6
mkdir -p $O
for i in 1 2 3; do
  for v in 0 1; do
    KOSK_COPY_NULL=$v examples/throughput --fs host --steps 1800 --warmup 180 > $O/cn_${v}_$i.json 2> $O/cn.err || { cat $O/cn.err; exit 1; }
    python - $O/cn_${v}_$i.json $v <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read())
print("copy_null=%s: %.1f k proofs/s, step %.2f ms (p99 %.2f), %.2f cores" % (sys.argv[2], j["proofs_per_s"] / 1e3, j["step_latency_ms"]["median"], j["step_latency_ms"]["p99"], j["host_cpu_cores_busy"]))
PY
  done
done
KOSK_COPY_NULL=1 timeout -k 10 600 python -m pytest tests/test_gpu_01_prover.py tests/test_gpu_02_verify.py tests/test_gpu_10_combine.py -x -q -m gpu > $O/cn_tests.log 2>&1 || { tail -30 $O/cn_tests.log; exit 1; }
tail -2 $O/cn_tests.log
KOSK_COPY_NULL=1 examples/throughput --fs host --callers 24 --combine 8 --steps 1800 --warmup 180 > $O/cn_1_c8.json 2>> $O/cn.err || exit 1
python -c "
import json; j=json.loads(open('gpurun_out/r6/cn_1_c8.json').read()); print('copy_null=1 c8: %.1f k, %.2f ms, %.2f cores' % (j['proofs_per_s']/1e3, j['step_latency_ms']['median'], j['host_cpu_cores_busy']))"
