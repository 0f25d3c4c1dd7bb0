#!/usr/bin/env python3
"""View-hash kernel time vs lane count (tail effects around one wave per SIMD). Not product code."""
import sys, torch
sys.path.insert(0, ".")
from mpcith_kyber_kosk_amd import api
ctx = api.Kosk(kyber_k=3, max_batch=46, device=0)
g = torch.Generator(device="cuda"); g.manual_seed(1)
for lanes in [int(x) for x in sys.argv[1:]] or [32768, 49152, 61440, 65536, 66884, 67712, 69632, 73728, 81920, 98304, 131072, 196608, 262144]:
    rows = torch.randint(0, 3329, (220, lanes), dtype=torch.int16, device="cuda", generator=g)
    pre = torch.randint(0, 256, (lanes, 32), dtype=torch.uint8, device="cuda", generator=g)
    dig = torch.zeros((lanes, 32), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for _ in range(3): ctx.commit_hash_lanes(rows.data_ptr(), lanes, lanes, pre.data_ptr(), 1, dig.data_ptr())
    ctx.synchronize(); ctx.timer_start()
    for _ in range(20): ctx.commit_hash_lanes(rows.data_ptr(), lanes, lanes, pre.data_ptr(), 1, dig.data_ptr())
    ms = ctx.timer_stop_ms() / 20
    print("lanes %7d waves %5d  %.1f us  %.1f ns/wave-slot  %.1f GB/s" % (lanes, (lanes + 63) // 64, ms * 1e3, ms * 1e6 / ((lanes + 63) // 64) * 1024, lanes * 504 / ms / 1e6))
