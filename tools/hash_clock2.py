#!/usr/bin/env python3
"""Follow-up to hash_clock.py: the series of hash launch times right behind an int8-MFMA product, and the other way round.
Not product code."""
import sys, time, torch
sys.path.insert(0, ".")
from mpcith_kyber_kosk_amd import api
ctx = api.Kosk(kyber_k=3, max_batch=46, device=0)
g = torch.Generator(device="cuda"); g.manual_seed(1)
lanes = 65536
rows = torch.randint(0, 3329, (220, lanes), dtype=torch.int16, device="cuda", generator=g)
pre = torch.randint(0, 256, (lanes, 32), dtype=torch.uint8, device="cuda", generator=g)
dig = torch.zeros((lanes, 32), dtype=torch.uint8, device="cuda")
n = 9982
y = torch.randint(0, 3329, (n, 407), dtype=torch.int16, device="cuda", generator=g)
sh = torch.zeros((n, 1454), dtype=torch.int16, device="cuda")
def hash_once():
    ctx.timer_start(); ctx.commit_hash_lanes(rows.data_ptr(), lanes, lanes, pre.data_ptr(), 1, dig.data_ptr()); return ctx.timer_stop_ms() * 1e3
def exp_once(m=n):
    ctx.timer_start(); ctx.lagrange_expand(y.data_ptr(), sh.data_ptr(), m); return ctx.timer_stop_ms() * 1e3
for _ in range(5): hash_once()
print("hash, warm:", " ".join("%.1f" % hash_once() for _ in range(6)))
for rep in range(3):
    e = exp_once()
    print("expansion %.1f us, then hash x 8:" % e, " ".join("%.1f" % hash_once() for _ in range(8)))
print("expansion x 8 back to back:", " ".join("%.1f" % exp_once() for _ in range(8)))
print("alternating expansion / hash:", " ".join("%.1f/%.1f" % (exp_once(), hash_once()) for _ in range(6)))
e = exp_once(414)
print("small expansion (414 rows) %.1f us, then hash x 6:" % e, " ".join("%.1f" % hash_once() for _ in range(6)))
# does the order of the two kernels inside ONE timed pair matter?  (events around both)
def pair(first_hash):
    ctx.timer_start()
    if first_hash:
        ctx.commit_hash_lanes(rows.data_ptr(), lanes, lanes, pre.data_ptr(), 1, dig.data_ptr()); ctx.lagrange_expand(y.data_ptr(), sh.data_ptr(), n)
    else:
        ctx.lagrange_expand(y.data_ptr(), sh.data_ptr(), n); ctx.commit_hash_lanes(rows.data_ptr(), lanes, lanes, pre.data_ptr(), 1, dig.data_ptr())
    return ctx.timer_stop_ms() * 1e3
print("pair hash+expansion:", " ".join("%.1f" % pair(True) for _ in range(5)), "| expansion+hash:", " ".join("%.1f" % pair(False) for _ in range(5)))
# is it the instruction cache?  a tiny different kernel (one workgroup) in front of each hash launch
p1 = torch.randint(0, 3329, (4, 256), dtype=torch.int16, device="cuda", generator=g); po1 = torch.zeros_like(p1)
def tiny_then_hash():
    ctx.ntt256_batch(p1.data_ptr(), po1.data_ptr(), 4)
    return hash_once()
print("tiny NTT launch (4 polynomials) then hash:", " ".join("%.1f" % tiny_then_hash() for _ in range(6)))
t = torch.zeros(16, device="cuda")
def torch_then_hash():
    t.add_(1.0); torch.cuda.synchronize()
    return hash_once()
print("tiny torch kernel (other stream) then hash:", " ".join("%.1f" % torch_then_hash() for _ in range(6)))
# the Tcomm flavour (3 blocks, shorter code)
def hash0_once():
    ctx.timer_start(); ctx.commit_hash_lanes(rows.data_ptr(), lanes, lanes, pre.data_ptr(), 0, dig.data_ptr()); return ctx.timer_stop_ms() * 1e3
print("Tcomm hash warm:", " ".join("%.1f" % hash0_once() for _ in range(4)), "| behind the view hash:", " ".join("%.1f" % (hash_once() * 0 + hash0_once()) for _ in range(4)))
# placement, not clocks or caches (tools/hash_clock_pmc.sh: same wave-cycles, 54 % more busy time): does a launch of the same
# shape in between put the dispatcher back into its even pattern?
def seq():
    exp_once()
    a = hash0_once()
    b = hash_once()
    return a, b
print("expansion, Tcomm hash, view hash:", " ".join("%.1f/%.1f" % seq() for _ in range(6)))
