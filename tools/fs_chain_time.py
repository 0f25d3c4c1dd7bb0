#!/usr/bin/env python3
"""Latency of the device Fiat-Shamir chains (round 6): sha3_256 of n digest tables of 46 528 bytes, one wave per table
(kosk_sha3_256_batch_wave = k_fs_chain), against the one-state-per-lane sponge on the same messages (kosk_sha3_256_batch) and the lane-pair
sponge; then the two derivation kernels.  343 permutations per table.  Usage: python tools/fs_chain_time.py [out.txt]"""
import hashlib
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mpcith_kyber_kosk_amd import api

L = 1454 * 32
NPERM = L // 136 + 1
out = []


def say(s):
    print(s, flush=True)
    out.append(s)


c = api.Kosk(kyber_k=3, max_batch=8)
rng = np.random.default_rng(1)
say("# sha3_256 of n tables of %d bytes (%d permutations each); best of 5; us per launch, us per permutation of one chain" % (L, NPERM))
say("# n      wave-sponge us   per-perm us   | one-state-per-lane us  per-perm us | lane-pair us  per-perm us")
for n in (1, 8, 46, 138, 276, 552, 1024, 2048, 4096):
    msgs = rng.integers(0, 256, size=(n, L), dtype=np.uint8)
    d_in = torch.from_numpy(msgs).cuda()
    d_out = torch.zeros((n, 32), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    res = []
    for idx, fn in enumerate((c.sha3_256_batch_wave, c.sha3_256_batch, c.sha3_256_batch_pair)):
        if idx and n > 1024:  # the per-lane sponges take 16 ms per launch: not beyond 1 024 tables
            res.append(float("nan"))
            continue
        best = 1e9
        for it in range(5):
            c.timer_start()
            fn(d_in.data_ptr(), L, L, d_out.data_ptr(), n)
            best = min(best, c.timer_stop_ms() * 1e3)
        res.append(best)
        got = d_out.cpu().numpy()
        for i in (0, n - 1):
            assert got[i].tobytes() == hashlib.sha3_256(msgs[i].tobytes()).digest()
    say("%5d  %12.1f  %10.3f   | %14.1f  %10.3f     | %10.1f  %10.3f" % (n, res[0], res[0] / NPERM, res[1], res[1] / NPERM, res[2], res[2] / NPERM))

say("# derivations (k_fs_chain<FS_ALPHA>: chain + 2 PRF permutations; <FS_OPENED>: chain + 3 + probing + complement), us per launch")
for n in (1, 46, 276, 552):
    tabs = torch.from_numpy(rng.integers(0, 256, size=(n, L), dtype=np.uint8)).cuda()
    d_a = torch.zeros((n, 80), dtype=torch.int16, device="cuda")
    d_sel = torch.zeros((n, 1312), dtype=torch.int16, device="cuda")
    d_rest = torch.zeros((n, 1312), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    ba = bo = 1e9
    for it in range(5):
        c.timer_start(); c.fs_alpha_device(tabs.data_ptr(), L, n, d_a.data_ptr()); ba = min(ba, c.timer_stop_ms() * 1e3)
        c.timer_start(); c.fs_opened_device(tabs.data_ptr(), L, n, d_sel.data_ptr(), d_rest.data_ptr(), 1312); bo = min(bo, c.timer_stop_ms() * 1e3)
    say("%5d  alpha %9.1f   opened %9.1f" % (n, ba, bo))
c.close()
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(out) + "\n")
