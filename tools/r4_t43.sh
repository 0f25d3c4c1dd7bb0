cd $GRAFT_REPO_ROOT
bash tools/r4_configs.sh
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python tools/latency_one.py 2>/dev/null | tail -6
