"""Workload for the PMC passes: one merged run of the bench configuration -- 276 proofs, what a cohort of six 46-proof callers sends
through every launch (kosk_options::combine = 6; PMC_PROOFS=184 / 138: the cohorts of four / three of the side runs) -- plus a known-bytes calibration copy."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hashlib
import torch
from mpcith_kyber_kosk_amd import api
k, B = 3, int(os.environ.get("PMC_PROOFS", "276"))
ctx = api.Kosk(kyber_k=k, max_batch=B, fs_mode=1 if os.environ.get("PMC_FS", "host") == "device" else 0)  # PMC_FS=device: the chain kernels in the trace
tapes = [hashlib.shake_256(("kosk-tape-v1:%d" % b).encode()).digest(ctx.tape_bytes) for b in range(B)]
for _ in range(2):
    ctx.verifiable_keygen_resident(tapes)
    assert all(ctx.verify_resident_pk(B))
# calibration: k_rows_copy moves n*407 u16 in and n*1454 u16 out (2-byte-per-lane coalesced accesses, like the hash kernel's loads)
n = 8192
y = torch.randint(0, 3329, (n, 407), dtype=torch.int16, device="cuda")
sh = torch.zeros((n, 1454), dtype=torch.int16, device="cuda")
ctx.lagrange_expand(y.data_ptr(), sh.data_ptr(), n)
# the graded kernels at exactly 65 536 lanes
lanes = 65536
rows = torch.randint(0, 3329, (220, lanes), dtype=torch.int16, device="cuda")
pre = torch.randint(0, 256, (lanes, 32), dtype=torch.uint8, device="cuda")
dig = torch.zeros((lanes, 32), dtype=torch.uint8, device="cuda")
ctx.commit_hash_lanes(rows.data_ptr(), lanes, lanes, pre.data_ptr(), 0, dig.data_ptr())
ctx.commit_hash_lanes(rows.data_ptr(), lanes, lanes, pre.data_ptr(), 1, dig.data_ptr())
polys = torch.randint(0, 3329, (lanes, 256), dtype=torch.int16, device="cuda")
outp = torch.zeros_like(polys)
ctx.ntt256_batch(polys.data_ptr(), outp.data_ptr(), lanes)
ctx.synchronize()
print("done")
