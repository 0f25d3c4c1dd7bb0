#!/bin/bash
# four-minute soak of the default arrangement (host Fiat-Shamir), then two minutes in device mode with byte checks every 200th step
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r6f; mkdir -p $O
python tools/stress_combine.py 18 45000 0 > $O/soak_long_host.txt 2>&1 || { tail -5 $O/soak_long_host.txt; exit 1; }
tail -1 $O/soak_long_host.txt
STRESS_FS=device python tools/stress_combine.py 18 12000 200 > $O/soak_long_device.txt 2>&1 || { tail -5 $O/soak_long_device.txt; exit 1; }
tail -1 $O/soak_long_device.txt
