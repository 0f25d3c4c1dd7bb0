cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
KOSK_COMBINE_IDLE_US=20000 timeout -k 10 500 python tools/stress_combine.py 9 12000 3000 > gpurun_out/r4/soak2.txt 2>&1; echo "rc $?"; tail -2 gpurun_out/r4/soak2.txt
timeout -k 10 300 python tools/stress_combine.py 9 2000 50 >> gpurun_out/r4/soak2.txt 2>&1; echo "rc $?"; tail -1 gpurun_out/r4/soak2.txt
