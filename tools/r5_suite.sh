#!/bin/bash
# Round 5: the whole GPU suite with the fatal-signal backtrace helper preloaded (tools/abort_trace.c), so that an abort anywhere
# in a suite run leaves a C backtrace in the run's own log.   usage (GPU box, repo root): tools/r5_suite.sh <log>
log=${1:-gpurun_out/r5/suite.log}
mkdir -p $(dirname $log)
gcc -shared -fPIC -O1 -o tools/libabort_trace.so tools/abort_trace.c -ldl || exit 1
LD_PRELOAD=$PWD/tools/libabort_trace.so timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $log 2>&1
rc=$?
echo "suite rc=$rc"
tail -4 $log
exit $rc
