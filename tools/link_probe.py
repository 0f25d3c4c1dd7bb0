"""PCIe link probe (GPU box): page-locked host <-> HBM copy rates, each direction alone and both at once, with the runtime's copy
path forced either way (HSA_ENABLE_SDMA=0: every copy is a blit kernel; default: the runtime's own choice).
    python3 tools/link_probe.py            (spawns itself once per setting)"""
import json
import os
import subprocess
import sys
import time

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out = bench.link_rate(torch, "cuda:0", mib=int(sys.argv[2]))
    # the same with four streams per direction (several copies in flight each way)
    n = int(sys.argv[2]) << 20
    hs = [torch.empty(n // 4, dtype=torch.uint8).pin_memory() for _ in range(8)]
    ds = [torch.empty(n // 4, dtype=torch.uint8, device="cuda:0") for _ in range(8)]
    st = [torch.cuda.Stream("cuda:0") for _ in range(8)]

    def both4():
        for i in range(4):
            with torch.cuda.stream(st[i]):
                ds[i].copy_(hs[i], non_blocking=True)
            with torch.cuda.stream(st[4 + i]):
                hs[4 + i].copy_(ds[4 + i], non_blocking=True)
    both4(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        both4()
    torch.cuda.synchronize()
    out["both_4_streams_each_GBps_each"] = 4 * n / (time.perf_counter() - t0) / 1e9
    print(json.dumps(out))
    sys.exit(0)

for env in ({}, {"HSA_ENABLE_SDMA": "0"}, {"HSA_ENABLE_SDMA": "1", "GPU_FORCE_BLIT_COPY_SIZE": "0"}):
    for mib in (64, 256):
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(mib)], env=e, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=300)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        j = json.loads(line[-1]) if line else {}
        print("%-55s %4d MiB: h2d %5.1f  d2h %5.1f  both (each) %5.1f  both, 4 streams per direction (each) %5.1f GB/s"
              % (env or "default", mib, j.get("h2d_GBps", 0), j.get("d2h_GBps", 0), j.get("both_directions_GBps_each", 0), j.get("both_4_streams_each_GBps_each", 0)))
