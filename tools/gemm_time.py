"""Timing experiment for the expansion GEMM at the GEMM1 size of the bench (9982 sharings)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpcith_kyber_kosk_amd import api
n = int(sys.argv[1]) if len(sys.argv) > 1 else 9982
ctx = api.Kosk(kyber_k=3, max_batch=46)
y = torch.randint(0, 3329, (n, 407), dtype=torch.int16, device="cuda")
sh = torch.zeros((n, 1454), dtype=torch.int16, device="cuda")
torch.cuda.synchronize()
for _ in range(3):
    ctx.lagrange_expand(y.data_ptr(), sh.data_ptr(), n)
ctx.synchronize()
ctx.timer_start()
for _ in range(10):
    ctx.lagrange_expand(y.data_ptr(), sh.data_ptr(), n)
print("dbg=%s n=%d: %.1f us per expand (incl. 2 row copies + limb conversion)" % (os.environ.get("KOSK_GEMM_DBG", "0"), n, ctx.timer_stop_ms() * 100))
