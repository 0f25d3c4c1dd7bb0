#!/bin/bash
# Round 5: the final-tree evidence in one call: whole GPU suite (abort-trace helper preloaded), the bench lines (default flags, the driver's
# flags, configs 2 / 4 / 5), batch-of-1 latency, then the rocprof / PMC profiles.   usage: tools/r5_final.sh <outdir>
out=${1:-gpurun_out/r5/final}; mkdir -p $out
tools/r5_suite.sh $out/suite.log || exit 1
timeout -k 10 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err || exit 1
echo "default: $(python3 -c "import json,sys; j=json.loads(open('$out/bench_default.json').read().strip().splitlines()[-1]); print(j['value'], j['step_latency_ms']['median'], j['roofline']['frac'])")"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $out/bench_driver_flags.json 2> $out/bench_driver_flags.err || exit 1
echo "driver flags: $(python3 -c "import json,sys; j=json.loads(open('$out/bench_driver_flags.json').read().strip().splitlines()[-1]); print(j['value'], j['step_latency_ms']['median'], j['roofline']['frac'])")"
for c in 2 4 5; do
  timeout -k 10 300 python bench.py --config $c --no-kernels --no-cpu-baseline > $out/bench_config$c.json 2> $out/bench_config$c.err || exit 1
  echo "config $c: $(python3 -c "import json,sys; j=json.loads(open('$out/bench_config$c.json').read().strip().splitlines()[-1]); print(j['value'], j['step_latency_ms']['median'])")"
done
timeout -k 10 120 python tools/latency_one.py 3 > $out/latency_one.txt 2>&1; cat $out/latency_one.txt
KOSK_HOST_THREADS=4 timeout -k 10 200 python tools/stress_combine.py 18 6000 0 > $out/soak.txt 2>&1; tail -2 $out/soak.txt
KOSK_HOST_THREADS=4 timeout -k 10 200 python tools/stress_combine.py 18 1000 50 >> $out/soak.txt 2>&1; tail -1 $out/soak.txt
timeout -k 10 900 tools/make_profiles.sh r05 > $out/make_profiles.log 2>&1; tail -3 $out/make_profiles.log
cp -r gpurun_out/prof/r05_* gpurun_out/prof/traffic.json $out/ 2>/dev/null
echo final done
