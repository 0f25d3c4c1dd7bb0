"""NTT-256 kernel time, integer Montgomery (default) and packed fp32 (KOSK_NTT_FP32=1). Not product code."""
import os, sys, torch
sys.path.insert(0, ".")
from mpcith_kyber_kosk_amd import api
for fp32 in ("0", "1"):
    os.environ["KOSK_NTT_FP32"] = fp32
    ctx = api.Kosk(kyber_k=3, max_batch=4, device=0)
    for lanes in (65536, 262144):
        polys = torch.randint(0, 3329, (lanes, 256), dtype=torch.int16, device="cuda")
        outp = torch.zeros_like(polys)
        for _ in range(3): ctx.ntt256_batch(polys.data_ptr(), outp.data_ptr(), lanes)
        ctx.synchronize(); ctx.timer_start()
        for _ in range(20): ctx.ntt256_batch(polys.data_ptr(), outp.data_ptr(), lanes)
        ms = ctx.timer_stop_ms() / 20
        print("ntt256 %s %d polys: %.1f us, %.0f GB/s (%.1f %% of 8 TB/s)" % ("packed-fp32" if fp32 == "1" else "integer    ", lanes, ms * 1e3, lanes * 1024 / ms / 1e6, lanes * 1024 / ms / 1e6 / 80))
    ctx.close()
