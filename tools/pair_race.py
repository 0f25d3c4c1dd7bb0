#!/usr/bin/env python3
"""k448 (small expansion GEMM) next to one other kernel family on another stream: which pairing corrupts results?"""
import sys, threading, numpy as np, torch, hashlib
sys.path.insert(0, ".")
from mpcith_kyber_kosk_amd import api
other = sys.argv[1]; N = int(sys.argv[2])
ca = api.Kosk(kyber_k=3, max_batch=12, device=0); cb = api.Kosk(kyber_k=3, max_batch=12, device=0)
rows = 108
y = torch.randint(0, 3329, (rows, 407), dtype=torch.int16, device="cuda"); o = torch.zeros((rows, 1454), dtype=torch.int16, device="cuda")
ca.lagrange_expand(y.data_ptr(), o.data_ptr(), rows); ca.synchronize(); ref = o.clone()
bad = {"k448": 0, other: 0}
stop = False
def loop_a():
    if len(sys.argv) > 3 and sys.argv[3] == "pipeline":   # load generator: the whole prove+verify pipeline on 46 proofs
        cc = api.Kosk(kyber_k=3, max_batch=46, device=0)
        tp = [hashlib.shake_256(b"kosk-tape-v1:%d" % i).digest(cc.tape_bytes) for i in range(46)]
        cc.stage_prover_inputs(tp)
        for it in range(N):
            cc.prove_resident(46)
            if not all(cc.verify_resident(46)): bad["k448"] += 1
        return
    for it in range(N):
        o.zero_(); torch.cuda.synchronize()
        ca.lagrange_expand(y.data_ptr(), o.data_ptr(), rows); ca.synchronize()
        if (o != ref).any().item(): bad["k448"] += 1
def loop_b():
    if other == "ntt":
        p = torch.randint(0, 3329, (4096, 256), dtype=torch.int16, device="cuda"); q = torch.zeros_like(p)
        cb.ntt256_batch(p.data_ptr(), q.data_ptr(), 4096); cb.synchronize(); r = q.clone()
        while not stop:
            q.zero_(); torch.cuda.synchronize(); cb.ntt256_batch(p.data_ptr(), q.data_ptr(), 4096); cb.synchronize()
            if (q != r).any().item(): bad[other] += 1
    elif other == "hash":
        lanes = 16384
        rowsd = torch.randint(0, 3329, (220, lanes), dtype=torch.int16, device="cuda"); pre = torch.randint(0, 256, (lanes, 32), dtype=torch.uint8, device="cuda")
        dig = torch.zeros((lanes, 32), dtype=torch.uint8, device="cuda")
        cb.commit_hash_lanes(rowsd.data_ptr(), lanes, lanes, pre.data_ptr(), 1, dig.data_ptr()); cb.synchronize(); r = dig.clone()
        while not stop:
            dig.zero_(); torch.cuda.synchronize(); cb.commit_hash_lanes(rowsd.data_ptr(), lanes, lanes, pre.data_ptr(), 1, dig.data_ptr()); cb.synchronize()
            if (dig != r).any().item(): bad[other] += 1
    elif other == "copy":   # plain torch kernels only
        p = torch.randint(0, 3329, (4096, 256), dtype=torch.int16, device="cuda"); q = torch.zeros_like(p)
        while not stop:
            q.zero_(); torch.cuda.synchronize(); q.copy_(p); torch.cuda.synchronize()
            if (q != p).any().item(): bad[other] += 1
    elif other == "big":   # the k-loop GEMM kernel (48 KB LDS per workgroup)
        yy = torch.randint(0, 3329, (4000, 407), dtype=torch.int16, device="cuda"); oo = torch.zeros((4000, 1454), dtype=torch.int16, device="cuda")
        cb.lagrange_expand(yy.data_ptr(), oo.data_ptr(), 4000); cb.synchronize(); r = oo.clone()
        while not stop:
            oo.zero_(); torch.cuda.synchronize(); cb.lagrange_expand(yy.data_ptr(), oo.data_ptr(), 4000); cb.synchronize()
            if (oo != r).any().item(): bad[other] += 1
    elif other == "prove":
        tapes = [hashlib.shake_256(b"kosk-tape-v1:%d" % i).digest(cb.tape_bytes) for i in range(12)]
        cb.stage_prover_inputs(tapes); cb.prove_resident(12); assert all(cb.verify_resident(12))
        while not stop:
            cb.prove_resident(12)
            if not all(cb.verify_resident(12)): bad[other] += 1
ta = threading.Thread(target=loop_a); tb = threading.Thread(target=loop_b)
tb.start(); ta.start(); ta.join(); stop = True; tb.join()
print(other, bad)
