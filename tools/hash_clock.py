#!/usr/bin/env python3
"""Is the in-pipeline time of the commitment hash (65 us at 1 012 waves, 43 us back to back at 1 024) a clock effect?
Times single launches of the 65 536-lane view hash (a) back to back, (b) each after an idle gap, (c) each right after an int8
MFMA product (the expansion of 9 982 rows).  Not product code."""
import sys, time, torch
sys.path.insert(0, ".")
from mpcith_kyber_kosk_amd import api
ctx = api.Kosk(kyber_k=3, max_batch=46, device=0)
g = torch.Generator(device="cuda"); g.manual_seed(1)
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rows = torch.randint(0, 3329, (220, lanes), dtype=torch.int16, device="cuda", generator=g)
pre = torch.randint(0, 256, (lanes, 32), dtype=torch.uint8, device="cuda", generator=g)
dig = torch.zeros((lanes, 32), dtype=torch.uint8, device="cuda")
n = 9982
y = torch.randint(0, 3329, (n, 407), dtype=torch.int16, device="cuda", generator=g)
sh = torch.zeros((n, 1454), dtype=torch.int16, device="cuda")
def hash_once():
    ctx.timer_start(); ctx.commit_hash_lanes(rows.data_ptr(), lanes, lanes, pre.data_ptr(), 1, dig.data_ptr()); return ctx.timer_stop_ms() * 1e3
for _ in range(5): hash_once()
ctx.timer_start()
for _ in range(20): ctx.commit_hash_lanes(rows.data_ptr(), lanes, lanes, pre.data_ptr(), 1, dig.data_ptr())
print("back to back, 20 launches: %.1f us each" % (ctx.timer_stop_ms() * 1e3 / 20))
for gap in (0.0, 0.0002, 0.001, 0.005, 0.02):
    ts = []
    for _ in range(12):
        ctx.synchronize(); time.sleep(gap); ts.append(hash_once())
    ts.sort(); print("single launch after %5.1f ms idle: median %.1f us (min %.1f)" % (gap * 1e3, ts[len(ts) // 2], ts[0]))
ts = []
for _ in range(12):
    ctx.synchronize()
    for _ in range(3): ctx.lagrange_expand(y.data_ptr(), sh.data_ptr(), n)
    ts.append(hash_once())
ts.sort(); print("single launch right behind three expansion products (int8 MFMA): median %.1f us (min %.1f)" % (ts[len(ts) // 2], ts[0]))
# how long does it last, what triggers it?
big = torch.empty(64 << 20, dtype=torch.uint8, device="cuda"); big2 = torch.empty_like(big)
def after(what, fn, gap=0.0):
    ts = []
    for _ in range(12):
        ctx.synchronize(); fn(); 
        if gap: ctx.synchronize(); time.sleep(gap)
        ts.append(hash_once())
    ts.sort(); print("single launch behind %s%s: median %.1f us (min %.1f)" % (what, (" + %.1f ms idle" % (gap * 1e3)) if gap else "", ts[len(ts) // 2], ts[0]))
after("one expansion product", lambda: ctx.lagrange_expand(y.data_ptr(), sh.data_ptr(), n))
after("a 64 MB device copy (no MFMA; evicts the caches)", lambda: (big2.copy_(big), torch.cuda.synchronize()))
for gap in (0.0001, 0.0005, 0.002, 0.01):
    after("three expansion products", lambda: [ctx.lagrange_expand(y.data_ptr(), sh.data_ptr(), n) for _ in range(3)], gap)
ys = torch.randint(0, 3329, (414, 407), dtype=torch.int16, device="cuda", generator=g); shs = torch.zeros((414, 1454), dtype=torch.int16, device="cuda")
after("one small expansion product (414 rows)", lambda: ctx.lagrange_expand(ys.data_ptr(), shs.data_ptr(), 414))
# and the other way round: the product's own time behind the hash
def exp_once():
    ctx.timer_start(); ctx.lagrange_expand(y.data_ptr(), sh.data_ptr(), n); return ctx.timer_stop_ms() * 1e3
ts = sorted(exp_once() for _ in range(12)); print("expansion product (9 982 rows) back to back: median %.1f us" % ts[6])
p = torch.randint(0, 3329, (65536, 256), dtype=torch.int16, device="cuda", generator=g); po = torch.zeros_like(p)
def ntt_once():
    ctx.timer_start(); ctx.ntt256_batch(p.data_ptr(), po.data_ptr(), 65536); return ctx.timer_stop_ms() * 1e3
ts = sorted(ntt_once() for _ in range(12)); print("NTT-256 x 65 536 alone: median %.1f us" % ts[6])
ts = []
for _ in range(12):
    ctx.synchronize(); [ctx.lagrange_expand(y.data_ptr(), sh.data_ptr(), n) for _ in range(3)]; ts.append(ntt_once())
ts.sort(); print("NTT-256 x 65 536 right behind three expansion products: median %.1f us" % ts[6])
