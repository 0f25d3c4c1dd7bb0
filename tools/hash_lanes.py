#!/usr/bin/env python3
"""View-hash kernel time vs lane count, LDS-DMA staged load path (the gather path of rounds 1-2 is what unaligned rows still get: profiles/r02_hash_lanes.txt).
Not product code.  usage: hash_lanes.py [lanes ...]"""
import os, sys, torch
sys.path.insert(0, ".")
from mpcith_kyber_kosk_amd import api
ctx = api.Kosk(kyber_k=3, max_batch=46, device=0)
g = torch.Generator(device="cuda"); g.manual_seed(1)
for lanes in [int(x) for x in sys.argv[1:]] or [32768, 65536, 66880, 131072, 262144, 1048576]:
    rows = torch.randint(0, 3329, (220, lanes), dtype=torch.int16, device="cuda", generator=g)
    pre = torch.randint(0, 256, (lanes, 32), dtype=torch.uint8, device="cuda", generator=g)
    dig = torch.zeros((lanes, 32), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for _ in range(3): ctx.commit_hash_lanes(rows.data_ptr(), lanes, lanes, pre.data_ptr(), 1, dig.data_ptr())
    ctx.synchronize(); ctx.timer_start()
    for _ in range(20): ctx.commit_hash_lanes(rows.data_ptr(), lanes, lanes, pre.data_ptr(), 1, dig.data_ptr())
    ms = ctx.timer_stop_ms() / 20
    import hashlib
    d0 = bytes(dig[7].tolist())
    msg = bytes(pre[7].tolist()) + rows[:, 7].contiguous().cpu().numpy().astype("<u2").tobytes()
    ok = hashlib.sha3_256(msg).digest() == d0
    print("lanes %7d waves %5d  %.1f us  %.1f GB/s  %.2f G Keccak-f/s  %s" % (lanes, (lanes + 63) // 64, ms * 1e3, lanes * 504 / ms / 1e6, lanes * 4 / ms / 1e6, "ok" if ok else "WRONG"))
