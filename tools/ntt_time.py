"""k_ntt256 at 65 536 (and 262 144) polynomials: time, TB/s, fraction of the 8 TB/s HBM peak; output checked against the oracle on samples."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpcith_kyber_kosk_amd import api
from tests import oracle_lib as oracle
import numpy as np
ctx = api.Kosk(kyber_k=3, max_batch=1)
for lanes in (65536, 262144, 1003):
    g = torch.Generator(device="cuda"); g.manual_seed(lanes)
    polys = torch.randint(0, 3329, (lanes, 256), dtype=torch.int16, device="cuda", generator=g)
    outp = torch.zeros_like(polys)
    torch.cuda.synchronize()
    for _ in range(5):
        ctx.ntt256_batch(polys.data_ptr(), outp.data_ptr(), lanes)
    ctx.synchronize()
    best = 1e9
    for rep in range(5):
        ctx.timer_start()
        for _ in range(20):
            ctx.ntt256_batch(polys.data_ptr(), outp.data_ptr(), lanes)
        best = min(best, ctx.timer_stop_ms() / 20)
    smp = sorted(set([0, 1, 15, 16, 17, lanes - 1, lanes - 16, lanes - 17] + np.random.default_rng(1).integers(0, lanes, 40).tolist()))
    hin, hout = polys[smp].cpu().numpy(), outp[smp].cpu().numpy()
    for j, i in enumerate(smp):
        assert np.array_equal(hout[j], oracle.poly_ntt(hin[j])), i
    print("polys %7d: %.2f us  %.2f TB/s  %.3f of HBM peak (sampled outputs == oracle)" % (lanes, best * 1e3, lanes * 1024 / best / 1e9, lanes * 1024 / best / 1e9 / 8.0))
ctx.close()
