"""Workload for counter passes (tools/hash_clock_pmc.sh): six view-hash launches back to back, then six each right behind an
expansion product.  Not product code."""
import sys, torch
sys.path.insert(0, ".")
from mpcith_kyber_kosk_amd import api
ctx = api.Kosk(kyber_k=3, max_batch=46, device=0)
g = torch.Generator(device="cuda"); g.manual_seed(1)
lanes, n = 65536, 9982
rows = torch.randint(0, 3329, (220, lanes), dtype=torch.int16, device="cuda", generator=g)
pre = torch.randint(0, 256, (lanes, 32), dtype=torch.uint8, device="cuda", generator=g)
dig = torch.zeros((lanes, 32), dtype=torch.uint8, device="cuda")
y = torch.randint(0, 3329, (n, 407), dtype=torch.int16, device="cuda", generator=g)
sh = torch.zeros((n, 1454), dtype=torch.int16, device="cuda")
for _ in range(8):
    ctx.commit_hash_lanes(rows.data_ptr(), lanes, lanes, pre.data_ptr(), 1, dig.data_ptr())
ctx.synchronize()
for _ in range(6):
    ctx.lagrange_expand(y.data_ptr(), sh.data_ptr(), n)
    ctx.commit_hash_lanes(rows.data_ptr(), lanes, lanes, pre.data_ptr(), 1, dig.data_ptr())
ctx.synchronize()
