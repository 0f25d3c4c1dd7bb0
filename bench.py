#!/usr/bin/env python3
"""Headline benchmark: Kyber KOSK proofs/s (key generation + prove + verify) on N MI355X.

One "step" = one pass of the hot path over one batch of synthetic input per GPU, through the two reference calls as
resident library calls:
    kosk_verifiable_keygen_resident   kyber_verifiable_keygen  (kosk.cpp:72-86): GPU key generation, offline + online
                                      prover; pk/sk come back to the host, the proofs stay in HBM
    kosk_verify_resident_pk           kyber_kosk_verify        (kosk.cpp:88-117): pk decoding + gen_matrix, verifier
Inputs (randomness tapes) are resident in HBM when the timed region starts; every step reads a different tape set
(--tape-sets per slot, rotated).  Every verify bit of every step is asserted.

--config picks the BASELINE.json workload (1-based, as VERDICT/SURVEY number them; configs[c-1] of BASELINE.json):
    2  Kyber-512,  46 proofs  = 66 884 party lanes per GPU per step
    3  Kyber-768,  46 proofs  = 66 884 party lanes per GPU per step            (default: the metric's configuration)
    4  Kyber-1024, 91 proofs  = 132 314 party lanes per GPU per step (2^20 lanes over 8 GPUs, proof-aligned), with the
       Tcomm and view-commitment digest tables all-gathered (RCCL) from HBM after each commitment round
    5  Kyber-768, 512 proofs per GPU per step (4096 keygens over 8 GPUs, throughput mode): proofs/s, batch latency,
       batch-of-1 latency

Several whole batches are kept in flight per GPU (--slots, one library context = one HIP stream + host threads each):
the protocol has two host Fiat-Shamir round trips per prove and per verify, and a slot's host phase hides under another
slot's kernels.  Timing: the slots run continuously; W warm-up steps, then EXACTLY K steps are timed from the completion
of step W to the completion of step W+K (a steady-state window: the pipeline is full on both edges), then S more steps
drain untimed.  barrier + torch.cuda.synchronize() bracket the whole run; `drained_run` reports the same run timed
from the first issue to the last completion (fill and drain included).  MAX over ranks; ranks shard by proof (no
data-path collective): scaling = weak.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import socket
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# NOTE: the library refuses AMD_DIRECT_DISPATCH=0 (stream synchronisation does not cover D2H copies in that mode on
# ROCm 7.2: the host would hash stale digests; tools/stress.py reproduces it).

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s spec, ~6.3 TB/s achievable)

CONFIGS = {  # 1-based config number -> (kyber_k, proofs per GPU per step, default slots, host threads per slot, handles per cohort)
    # configs 2 and 3: eighteen caller threads, each with its own handle and its own 46-proof calls; the library serves the calls of
    # a cohort of six handles with one pipeline run (kosk_options::combine = 6, include/kosk_mi355x.h), i.e. three merged runs of 276 proofs in
    # flight.  The default line is bound by the GPU, and the kernels are the more efficient the more proofs a launch
    # serves: on one box, alternating, cohorts of three / four / five / six give 150 / 161-164 / 170-175 / 179-184 k proofs/s at
    # 2.7 / 3.3 / 3.9 / 4.5 ms per call (profiles/r05_cohort_size.txt); eight per cohort is bound by the container's 16 host cores.
    # Round 4's line was cohorts of three, the first half of round 5 cohorts of four: `cohorts_of_three` and `cohorts_of_four` in
    # the line are those arrangements on the same box, for a like-for-like comparison.  Three Fiat-Shamir workers per caller (18 per
    # merged run: its 35 groups of eight tables in two rounds) instead of six, and no pre-wake spinning (Slot.__init__): 10.6-10.9 busy
    # host cores instead of 13.7-14.3 at the same rate (profiles/r05_host18.txt).
    2: dict(k=2, batch=46, slots=18, threads=3, combine=6, what="Kyber-512 (KYBER_K=2), 46 proofs = 66 884 party lanes per GPU per step"),
    3: dict(k=3, batch=46, slots=18, threads=3, combine=6, what="Kyber-768 (KYBER_K=3), 46 proofs = 66 884 party lanes per GPU per step"),
    4: dict(k=4, batch=91, slots=9, threads=3, combine=3, prewake_us=0, what="Kyber-1024 (KYBER_K=4), 91 proofs = 132 314 party lanes per GPU per step "
                                                    "(2^20 lanes over 8 GPUs, proof-aligned), digest tables all-gathered after each commitment round"),
    5: dict(k=3, batch=512, slots=4, threads=8, what="Kyber-768 (KYBER_K=3), 512 verifiable keygens per GPU per step (4096 over 8 GPUs, throughput mode)"),
}
VIEW_MSG = {2: 452, 3: 472, 4: 524}
TCOMM_MSG = {2: 308, 3: 320, 4: 332}


def cgroup_cpu():
    """(quota in cores or None, nr_throttled, throttled seconds) of this process's CPU cgroup (v2 or v1), None where unreadable: a
    container with a CPU quota stalls EVERY thread for the rest of a 100 ms period once the quota is spent (CFS bandwidth control), which
    shows in the line as a latency tail (step_latency_ms p99 / max) far above the median"""
    quota = nthr = thr_s = None
    try:
        for base in ("/sys/fs/cgroup", "/sys/fs/cgroup/cpu", "/sys/fs/cgroup/cpu,cpuacct"):
            if os.path.exists(base + "/cpu.max"):
                q, per = open(base + "/cpu.max").read().split()[:2]
                quota = None if q == "max" else float(q) / float(per)
            elif os.path.exists(base + "/cpu.cfs_quota_us"):
                q, per = int(open(base + "/cpu.cfs_quota_us").read()), int(open(base + "/cpu.cfs_period_us").read())
                quota = None if q <= 0 else q / per
            if os.path.exists(base + "/cpu.stat"):
                st = dict(l.split()[:2] for l in open(base + "/cpu.stat").read().splitlines() if len(l.split()) >= 2)
                if "nr_throttled" in st:
                    nthr = int(st["nr_throttled"])
                    thr_s = int(st["throttled_usec"]) * 1e-6 if "throttled_usec" in st else int(st.get("throttled_time", 0)) * 1e-9
                    break
    except Exception:  # noqa: BLE001
        pass
    return quota, nthr, thr_s


def tapes_for(first, count, nbytes):
    return [hashlib.shake_256(("kosk-tape-v1:%d" % (first + b)).encode()).digest(nbytes) for b in range(count)]


def cpu_baseline(k, tapes, nproofs):
    """The oracle (kind "port": single-thread scalar C restatement) timed on this host, main.cpp style."""
    lib_path = os.path.join(ROOT, "oracle", "libkosk_oracle.so")
    if not os.path.exists(lib_path):
        return None
    lib = C.CDLL(lib_path)
    tp, tv = C.c_double(), C.c_double()
    blob = b"".join(tapes[:nproofs])
    lib.ko_bench.restype = C.c_int
    good = lib.ko_bench(k, nproofs, C.c_char_p(blob), C.c_size_t(len(tapes[0])), C.byref(tp), C.byref(tv))
    if good != nproofs:
        raise RuntimeError("oracle rejected its own proofs")
    tot = tp.value + tv.value
    return {"value": nproofs / tot, "unit": "proofs/s", "cores": 1, "kind": "port",
            "sample": "%d of the first batch's proofs (same tapes), kyber_verifiable_keygen + kyber_kosk_verify, "
                      "single thread, clock(): prove %.2f s + verify %.2f s; host reports %d cores, %d usable by this process"
                      % (nproofs, tp.value, tv.value, os.cpu_count(), len(os.sched_getaffinity(0)))}


def _sample(n, count=64):
    import numpy as np
    rng = np.random.default_rng(n)
    return sorted(set([0, 1, 63, 64, n - 65, n - 1] + rng.integers(0, n, count - 6).tolist()))


def _ntt_by_definition(f):
    """Kyber's NTT from its definition (ntt.c:80-95 computes exactly this): outputs 2i, 2i+1 = the even / odd coefficients'
    polynomials evaluated at zeta^(2 br7(i) + 1), zeta = 17, mod q.  f: 256 ints; returns 256 ints in [0, q)."""
    import numpy as np
    q = 3329
    f = np.asarray(f, dtype=np.int64) % q
    out = np.zeros(256, np.int64)
    for i in range(128):
        z = pow(17, 2 * int("{:07b}".format(i)[::-1], 2) + 1, q)
        pw = np.array([pow(z, j, q) for j in range(128)], dtype=np.int64)
        out[2 * i] = int((f[0::2] * pw % q).sum() % q)
        out[2 * i + 1] = int((f[1::2] * pw % q).sum() % q)
    return out


def kernel_microbench(ctx, torch, k, lanes=65536, reps=20):
    """Graded kernels at exactly `lanes` lanes (BASELINE.json configs[1..2]); hipEvent pairs on the ctx stream around `reps` launches, the
    best of three such groups.  The outputs of the launches that were timed are checked on 64 sampled lanes (hashlib.sha3_256; the NTT's
    definition mod q)."""
    import numpy as np
    out = {}

    def timed(launch, n, groups=3):
        """ms per launch: `n` launches between two HIP events on the library's stream, the best of `groups` such groups -- one group alone is
        at the mercy of the runtime's rare multi-millisecond stalls inside a launch call (its system-memory pool growing): one
        of them in twenty 20 us launches reads as 500 us per launch (it did, in a round-6 run)"""
        best = None
        for _ in range(groups):
            ctx.timer_start()
            for _ in range(n):
                launch()
            t = ctx.timer_stop_ms() / n
            best = t if best is None or t < best else best
        return best
    tc_words = {2: 154, 3: 160, 4: 166}[k]
    vw_words = {2: 210, 3: 220, 4: 246}[k]
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    rows = torch.randint(0, 3329, (vw_words, lanes), dtype=torch.int16, device="cuda", generator=g)
    pre = torch.randint(0, 256, (lanes, 32), dtype=torch.uint8, device="cuda", generator=g)
    dig = torch.zeros((lanes, 32), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for name, words, wp in (("sha3_tcomm", tc_words, 0), ("sha3_view", vw_words, 1)):
        for _ in range(3):
            ctx.commit_hash_lanes(rows.data_ptr(), lanes, lanes, pre.data_ptr(), wp, dig.data_ptr())
        ctx.synchronize()
        dig.zero_()
        torch.cuda.synchronize()
        ms = timed(lambda: ctx.commit_hash_lanes(rows.data_ptr(), lanes, lanes, pre.data_ptr(), wp, dig.data_ptr()), reps)
        smp = _sample(lanes)
        h_rows = rows[:words][:, smp].cpu().numpy().astype("<u2")
        h_pre, h_dig = pre[smp].cpu().numpy(), dig[smp].cpu().numpy()
        for j, l in enumerate(smp):
            msg = (h_pre[j].tobytes() if wp else b"") + h_rows[:, j].tobytes()
            if h_dig[j].tobytes() != hashlib.sha3_256(msg).digest():
                raise RuntimeError("%s: digest of lane %d differs from hashlib.sha3_256" % (name, l))
        nbytes = lanes * (2 * words + 32 * wp + 32)
        perms = lanes * ((2 * words + 32 * wp) // 136 + 1)
        out[name] = {"lanes": lanes, "msg_bytes": 2 * words + 32 * wp, "us": ms * 1e3, "GBps": nbytes / ms / 1e6,
                     "frac_hbm_peak": nbytes / ms / 1e6 / HBM_PEAK_GBS, "keccak_f_per_s": perms / ms * 1e3,
                     "checked": "%d sampled lanes of the timed launches' output == hashlib.sha3_256" % len(smp)}
    for big in (262144, 1048576):  # the same view-hash kernel with several waves per SIMD (its saturated rate)
        rows_b = torch.randint(0, 3329, (vw_words, big), dtype=torch.int16, device="cuda", generator=g)
        pre_b = torch.randint(0, 256, (big, 32), dtype=torch.uint8, device="cuda", generator=g)
        dig_b = torch.zeros((big, 32), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        for _ in range(2):
            ctx.commit_hash_lanes(rows_b.data_ptr(), big, big, pre_b.data_ptr(), 1, dig_b.data_ptr())
        ctx.synchronize()
        ms = timed(lambda: ctx.commit_hash_lanes(rows_b.data_ptr(), big, big, pre_b.data_ptr(), 1, dig_b.data_ptr()), 5)
        nbytes = big * (2 * vw_words + 64)
        out["sha3_view_%d_lanes" % big] = {"us": ms * 1e3, "GBps": nbytes / ms / 1e6, "frac_hbm_peak": nbytes / ms / 1e6 / HBM_PEAK_GBS,
                                           "keccak_f_per_s": big * ((2 * vw_words + 32) // 136 + 1) / ms * 1e3}
        del rows_b, pre_b, dig_b
    polys = torch.randint(0, 3329, (lanes, 256), dtype=torch.int16, device="cuda", generator=g)
    outp = torch.zeros_like(polys)
    for _ in range(3):
        ctx.ntt256_batch(polys.data_ptr(), outp.data_ptr(), lanes)
    ctx.synchronize()
    ms = timed(lambda: ctx.ntt256_batch(polys.data_ptr(), outp.data_ptr(), lanes), reps)

    def check_ntt(what):
        smp = _sample(lanes, 24)
        h_in, h_out = polys[smp].cpu().numpy(), outp[smp].cpu().numpy()
        for j, l in enumerate(smp):
            if not np.array_equal(h_out[j].astype(np.int64) % 3329, _ntt_by_definition(h_in[j])) or abs(int(h_out[j].min())) > 1664 or int(h_out[j].max()) > 1664:
                raise RuntimeError("%s: polynomial %d differs from the NTT's definition" % (what, l))
        return "%d sampled polynomials of the timed launches' output == the NTT by definition (mod q, centred range)" % len(smp)
    out["ntt256"] = {"polys": lanes, "us": ms * 1e3, "GBps": lanes * 1024 / ms / 1e6,
                     "frac_hbm_peak": lanes * 1024 / ms / 1e6 / HBM_PEAK_GBS, "arithmetic": "integer Montgomery (default)",
                     "checked": check_ntt("ntt256")}
    # the Fiat-Shamir chain: sha3_256 of 276 digest tables of 46 528 bytes, one wave per table (k_fs_chain: 343 sequential permutations)
    nt, L = 276, 1454 * 32
    tabs = torch.randint(0, 256, (nt, L), dtype=torch.uint8, device="cuda", generator=g)
    dg = torch.zeros((nt, 32), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.sha3_256_batch_wave(tabs.data_ptr(), L, L, dg.data_ptr(), nt)
    ctx.synchronize()
    ms = timed(lambda: ctx.sha3_256_batch_wave(tabs.data_ptr(), L, L, dg.data_ptr(), nt), 5)
    for i in (0, nt - 1):
        if dg[i].cpu().numpy().tobytes() != hashlib.sha3_256(tabs[i].cpu().numpy().tobytes()).digest():
            raise RuntimeError("fs chain: digest %d differs from hashlib.sha3_256" % i)
    out["fs_chain_sha3_long"] = {"tables": nt, "bytes_per_table": L, "permutations_per_chain": L // 136 + 1, "us": ms * 1e3,
                                 "us_per_permutation": ms * 1e3 / (L // 136 + 1),
                                 "note": "one Keccak state per WAVE (a word per lane, DPP column sums + ds_bpermute exchanges): the latency of one "
                                         "sequential chain, not a throughput figure", "checked": "first and last digest == hashlib.sha3_256"}
    return out


def link_rate(torch, device, mib=256):
    """Measured PCIe rate of this box (GB/s): page-locked host memory <-> HBM, each direction alone and both at once."""
    n = mib << 20
    host_a = torch.empty(n, dtype=torch.uint8).pin_memory()
    host_b = torch.empty(n, dtype=torch.uint8).pin_memory()
    dev_a = torch.empty(n, dtype=torch.uint8, device=device)
    dev_b = torch.empty(n, dtype=torch.uint8, device=device)
    s1, s2 = torch.cuda.Stream(device), torch.cuda.Stream(device)

    def timed(fn, reps=4):
        fn(); torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize(device)
        return reps * n / (time.perf_counter() - t0) / 1e9

    def h2d():
        with torch.cuda.stream(s1):
            dev_a.copy_(host_a, non_blocking=True)

    def d2h():
        with torch.cuda.stream(s2):
            host_b.copy_(dev_b, non_blocking=True)

    def both():
        h2d(); d2h()
    out = {"h2d_GBps": timed(h2d), "d2h_GBps": timed(d2h)}
    out["both_directions_GBps_each"] = timed(both)
    out["note"] = "%d MiB page-locked host buffers, torch copies on two streams" % mib
    return out


def drop_in(api, torch, k, B, tapes, device):
    """The reference's two calls on HOST buffers (kosk_verifiable_keygen_batch + kosk_verify_batch and their compact-format
    twins) -- PCIe-inclusive, never `value`.  (i) one caller thread, plain (pageable) buffers; (ii) one caller thread,
    page-locked buffers from kosk_host_alloc; (iii) two caller threads, one proving and one verifying the previous call's proofs,
    so that both directions of the link are busy.  KOSK_STREAMS=3 handles, 6 x B proofs per call in chunks of B."""
    lib = api.lib
    out = {"link": link_rate(torch, "cuda:%d" % device)}
    n = 6 * B
    cb = lib.kosk_compact_proof_bytes(k)
    mk = lambda: api.Kosk(kyber_k=k, max_batch=3 * B, device=device, streams=3)
    prover, verifier = mk(), mk()
    pb, pkb, skb, tb = prover.proof_bytes, prover.pk_bytes, prover.sk_bytes, prover.tape_bytes
    blob = C.create_string_buffer(b"".join(tapes) * 6, tb * n)
    pk = [C.create_string_buffer(pkb * n) for _ in range(2)]
    sk = C.create_string_buffer(skb * n)
    ok = C.create_string_buffer(n)
    # per proof over PCIe: image (or compact) out + in, tape in, 2 x 46.5 KB digest tables out per call
    out["bytes_per_proof"] = {"image": pb, "compact": cb, "tape_h2d": tb, "digest_tables_d2h_per_call": 2 * 1454 * 32, "pk": pkb, "sk": skb}

    def rate(fn, reps):
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return reps * n / (time.perf_counter() - t0)

    def chk(rc, h, what, okbuf=None):
        if rc:
            raise RuntimeError("%s: %s" % (what, lib.kosk_last_error(h.handle).decode()))
        if okbuf is not None and okbuf.raw != b"\x01" * n:
            bad = [i for i, b_ in enumerate(okbuf.raw) if b_ != 1]
            if os.environ.get("KOSK_BENCH_DIAG"):
                masks = h.fail_masks(n)
                sys.stderr.write("DIAG %s: rejected %d: %s\n" % (what, len(bad), [(i, masks[i]) for i in bad][:8]))
                sys.stderr.write("DIAG path counts %s\n" % h.path_counts())
                if diag_buf[0] is not None:
                    # are the proofs themselves good?  prove the same tapes on a fresh single-stream handle, compare, and verify
                    # the suspect buffer there
                    fresh = api.Kosk(kyber_k=k, max_batch=B, device=device)
                    img = C.string_at(diag_buf[0], pb * n) if not isinstance(diag_buf[0], C.Array) else diag_buf[0].raw
                    ref = fresh.verifiable_keygen(tapes)
                    diff = [i for i in range(n) if img[i * pb:(i + 1) * pb] != ref[2][i % B]]
                    sys.stderr.write("DIAG proofs differing from a fresh handle's: %d %s\n" % (len(diff), diff[:12]))
                    okf = fresh.verify([img[i * pb:(i + 1) * pb] for i in range(B)], ref[0])
                    sys.stderr.write("DIAG fresh handle verifies chunk 0 of the buffer: %s\n" % (okf.count(True),))
                    rc2 = lib.kosk_verify_batch(h.handle, n, diag_buf[0], pk[0], okbuf)
                    sys.stderr.write("DIAG same handle, second try: rc %d rejected %d\n" % (rc2, sum(1 for b_ in okbuf.raw if b_ != 1)))
            raise RuntimeError("%s rejected %d of %d honest proofs: indices %s, fail masks %s" % (what, len(bad), n, bad[:12], h.fail_masks(n)[bad[0]:bad[0] + 4]))

    diag_buf = [None]

    def pair(buf, compact):
        def fn():
            diag_buf[0] = None if compact else buf
            if compact:
                chk(lib.kosk_verifiable_keygen_batch_compact(prover.handle, n, blob, tb, pk[0], sk, buf), prover, "keygen_batch_compact")
                chk(lib.kosk_verify_batch_compact(prover.handle, n, buf, pk[0], ok), prover, "verify_batch_compact", ok)
            else:
                chk(lib.kosk_verifiable_keygen_batch(prover.handle, n, blob, tb, pk[0], sk, buf), prover, "keygen_batch")
                chk(lib.kosk_verify_batch(prover.handle, n, buf, pk[0], ok), prover, "verify_batch", ok)
        return fn
    pageable = C.create_string_buffer(pb * n)
    out["one_thread_pageable_image"] = rate(pair(pageable, False), 2)
    del pageable
    pin = [lib.kosk_host_alloc(pb * n) for _ in range(2)]
    if not all(pin):
        raise RuntimeError("kosk_host_alloc failed")
    pinv = [C.c_void_p(p_) for p_ in pin]
    out["one_thread_pinned_image"] = rate(pair(pinv[0], False), 3)
    out["one_thread_pinned_compact"] = rate(pair(pinv[0], True), 3)

    def two_threads(compact, rounds=6):
        """thread A proves call i into buffer i % 2 while thread B verifies call i - 1 from the other buffer"""
        ready = [threading.Semaphore(0), threading.Semaphore(0)]
        free = [threading.Semaphore(1), threading.Semaphore(1)]
        err = []
        okv = C.create_string_buffer(n)

        def prove():
            try:
                for i in range(rounds):
                    free[i % 2].acquire()
                    f = lib.kosk_verifiable_keygen_batch_compact if compact else lib.kosk_verifiable_keygen_batch
                    if f(prover.handle, n, blob, tb, pk[i % 2], sk, pinv[i % 2]):
                        raise RuntimeError(lib.kosk_last_error(prover.handle))
                    ready[i % 2].release()
            except Exception as e:  # noqa: BLE001
                err.append(e)
                for r_ in ready:
                    r_.release()

        def verify():
            try:
                for i in range(rounds):
                    ready[i % 2].acquire()
                    if err:
                        return
                    f = lib.kosk_verify_batch_compact if compact else lib.kosk_verify_batch
                    if f(verifier.handle, n, pinv[i % 2], pk[i % 2], okv) or okv.raw != b"\x01" * n:
                        raise RuntimeError("verify: %s" % lib.kosk_last_error(verifier.handle))
                    free[i % 2].release()
            except Exception as e:  # noqa: BLE001
                err.append(e)
                for f_ in free:
                    f_.release()
        ta, tv = threading.Thread(target=prove), threading.Thread(target=verify)
        t0 = time.perf_counter()
        ta.start(); tv.start(); ta.join(); tv.join()
        dt = time.perf_counter() - t0
        if err:
            raise err[0]
        return rounds * n / dt
    two_threads(False, 2)  # warm both handles (verifier workspace, compact staging)
    two_threads(True, 2)
    out["two_threads_pinned_image"] = two_threads(False)
    out["two_threads_pinned_compact"] = two_threads(True)
    best = max(out["two_threads_pinned_image"], out["two_threads_pinned_compact"])
    lk = out["link"]
    per_dir = {"image": out["two_threads_pinned_image"] * (pb + 2 * 1454 * 32) / 1e9,
               "compact": out["two_threads_pinned_compact"] * (cb + 2 * 1454 * 32) / 1e9}
    out["two_threads_GBps_d2h"] = per_dir
    out["two_threads_fraction_of_link_d2h"] = {k_: v / lk["both_directions_GBps_each"] for k_, v in per_dir.items()}
    out["proofs_per_s"] = best
    out["note"] = ("kyber_verifiable_keygen + kyber_kosk_verify through the drop-in calls on host pointers, %d proofs per call in chunks of %d, "
                   "streams = 3; unit proofs/s; every verify bit asserted" % (n, B))
    for p_ in pinv:
        lib.kosk_host_free(p_)
    prover.close(); verifier.close()
    return out


def usable_host_cores():
    """cores this process can really run on: the scheduler affinity, cut to the container's CPU quota (cgroup v2 cpu.max / v1 cfs quota).
    A GPU box shows 256 hardware threads to a lease whose quota is 16 cores: sizing the host side by the affinity alone made every rank of an
    8-GPU job assume 32 cores, and CFS bandwidth control then stalls ALL threads of the container for the rest of each 100 ms period."""
    try:
        aff = len(os.sched_getaffinity(0))
    except AttributeError:
        aff = os.cpu_count() or 1
    quota = cgroup_cpu()[0]
    if quota is not None and quota > 0:
        return max(1, min(aff, int(quota)))  # whole cores (a 15.5-core quota is 15 usable cores)
    return aff


def host_budget(usable_cores, local_world, slots, threads):
    """(host threads per slot, sleep instead of spin) for a rank that shares `usable_cores` (usable_host_cores(): affinity cut to the
    cgroup quota) with local_world - 1 other ranks and runs `slots` pipeline slots.  With a core for every slot thread nothing changes;
    when cores are scarce (an 8-GPU node with few cores per GPU) the Fiat-Shamir pools shrink and, below twelve cores per rank, the slots'
    waits sleep on events --
    down to three threads per slot while the rank has at least four cores (the pool threads sleep between the four hash rounds of a step,
    so twice as many of them as cores is harmless, while fewer than three would stretch every round: 46 proofs = 6 groups of 8 AVX-512
    lanes), and down to two below that (eight ranks on a 16-core quota: two cores per rank)."""
    cores_per_rank = max(1, usable_cores // max(1, local_world))
    if cores_per_rank >= slots * (threads + 1):
        return threads, False
    floor = 3 if cores_per_rank >= 4 else 2
    # sleeping waits below a dozen cores per rank; from there up the default waits (a nap through most of the expected phase, then a
    # short spin) measured the same rate at one to two busy cores FEWER than sleeping on events (profiles/r05_host18.txt: 10.6-10.9
    # against 11.4-11.9 on a lease with a 16-core quota, never throttled)
    return max(floor, min(threads, 2 * cores_per_rank // slots)), cores_per_rank < 12


def threads_per_caller(budget_threads, blocking, combine, cores_per_rank):
    """Fiat-Shamir workers per handle from host_budget's per-slot figure: a merged run led by a cohort's first member uses `combine` times
    as many.  Floor three per caller, two when the rank has fewer than four cores; at most four when the waits sleep (few cores per rank)."""
    floor = 3 if cores_per_rank >= 4 else 2
    t = max(floor, budget_threads // max(1, combine))
    if blocking and combine > 1:
        t = min(t, 4 if cores_per_rank >= 4 else 2)
    return t


class Slot:
    """One pipeline slot: a library context with its tape bank in HBM (and, for config 4, its own process group)."""

    def __init__(self, api, torch, k, B, device, first_tape, nsets, combine=1, prewake_us=None, fs="host", threads=0, blocking=False):
        # combine = C > 1: the handle joins a cohort of C handles whose resident calls the library merges (kosk_options::combine).  The
        # slots are closed loops that pause between the bench's runs (conditioning, barrier, timed run): members stay "expected" by
        # their cohort for 20 ms instead of the library's default 1 ms, so that a run's first calls merge like all the others
        opts = dict(combine=max(1, combine), fs_mode=api.FS_DEVICE if fs == "device" else api.FS_HOST, host_threads=threads, blocking_sync=int(bool(blocking)))
        if combine > 1:
            opts["combine_idle_us"] = 20000
            # the callers of a merged run sleep to its end, without the library's 400 us of pre-wake spinning, (a) when the waits sleep
            # (few host cores, host_budget) and (b) in cohorts of five and more: fifteen members spinning for the last 400 us of every run
            # cost 2-3 busy cores and no longer buy throughput (profiles/r05_host18.txt; at three per cohort the pre-wake was +4.7 %, round 4)
            # (c) where the configuration says so (config 4's 273-proof runs of three callers: the same rate with 1.3 cores fewer)
            if blocking or combine >= 5 or prewake_us is not None:
                opts["combine_prewake_us"] = int(os.environ.get("KOSK_BENCH_PREWAKE_US", prewake_us or 0))
        self.c = api.Kosk(kyber_k=k, max_batch=B, device=device, **opts)
        self.B, self.nsets = B, nsets
        self.stride = (self.c.tape_bytes + 63) // 64 * 64
        import numpy as np
        host = np.zeros((nsets, B, self.stride), np.uint8)
        self.first_tapes = None
        for s in range(nsets):
            tp = tapes_for(first_tape + s * B, B, self.c.tape_bytes)
            if s == 0:
                self.first_tapes = tp
            for b, t in enumerate(tp):
                host[s, b, :len(t)] = np.frombuffer(t, np.uint8)
        self.bank = torch.from_numpy(host).to("cuda:%d" % device)  # randomness resident in HBM
        self.gather = None  # config 4: (dist, group, [out0, out1], world)
        self.pending = []
        self.steps_done = 0
        self.ph_sum = None
        self.t_prove = self.t_verify = 0.0  # wall seconds inside the two library calls (the rest of a step is the harness)
        self._fast = None

    def step(self, torch, index):
        """one step = the two library calls.  The harness between them is kept to a few bytecodes: the caller threads of a cohort come
        back from a merged run at the same moment and then queue for the interpreter lock, so every microsecond of Python here is paid
        once per caller before the cohort's next merged run can form (the raw ctypes entry points with prebuilt arguments; the
        wrappers of api.py -- argument conversion, a fresh result buffer and a 46-element list per call -- stay for everything else)."""
        c, B = self.c, self.B
        if self._fast is None:  # first call: through the wrapper (allocates the handle's pk / sk buffers), then bind the raw calls
            c.verifiable_keygen_resident(self.bank[index % self.nsets].data_ptr(), n=B, tape_stride=self.stride)
            import ctypes as C_
            from mpcith_kyber_kosk_amd import api as api_
            self._ok = C_.create_string_buffer(B)
            self._fast = (api_.lib.kosk_verifiable_keygen_resident, api_.lib.kosk_verify_resident_pk, c.handle, c._pk, c._sk,
                          [C_.c_void_p(self.bank[s_].data_ptr()) for s_ in range(self.nsets)], b"\x01" * B)
            t_a = t_b = time.perf_counter()
        else:
            kg, vf, h, pk, sk, ptrs, ones = self._fast
            t_a = time.perf_counter()
            if kg(h, B, ptrs[index % self.nsets], self.stride, pk, sk):
                c._chk(1, "verifiable_keygen_resident")
            t_b = time.perf_counter()
        self.t_prove += t_b - t_a
        if self.pending:
            # the verifier reuses the digest tables: the all-gathers that read them must have completed
            for w in self.pending:
                w.wait()
            torch.cuda.current_stream().synchronize()
            self.pending = []
        kg, vf, h, pk, sk, ptrs, ones = self._fast
        t_c = time.perf_counter()
        if vf(h, B, None, self._ok):
            c._chk(1, "verify_resident_pk")
        self.t_verify += time.perf_counter() - t_c
        if self.ph_sum is not None:  # --phase-stats: the library's phase clocks of THIS step (prove phases are still the last prove's)
            for i_, v_ in enumerate(c.phase_seconds()):
                self.ph_sum[i_] += v_
        if self._ok.raw != ones:
            ok = [b_ == 1 for b_ in self._ok.raw]
            raise RuntimeError("verifier rejected %d of %d honest proofs (masks %s)" % (ok.count(False), B, c.fail_masks(B)[:8]))
        self.steps_done += 1

    def enable_gather(self, api, torch, dist, group, world, host_staged=False):
        """host_staged (KOSK_BENCH_REHEARSE=1: several ranks on ONE GPU over gloo, which RCCL cannot do): the table is copied to
        the host and all-gathered there -- same hook order, same static step dealing, same pending.wait() ordering as the RCCL
        path, so that config 4's multi-rank control flow can run on a one-GPU box."""
        B = self.B
        # concatenation form [world * B][1454][32] (rank-major = global proof order of the contiguous partition)
        outs = [torch.empty((world * B, 1454, 32), dtype=torch.uint8, device="cpu" if host_staged else self.bank.device) for _ in range(2)]
        self.gather = (dist, group, outs, world)
        self.gathers_issued = 0

        def hook(role, rnd, ptr, nbytes):
            if role != 0:
                return  # the verifier's tables are the prover's again (opened digests recomputed, the rest from the proof)
            assert nbytes == B * 1454 * 32
            src = torch.as_tensor(api.DeviceView(ptr, (B, 1454, 32)), device=self.bank.device)
            if host_staged:
                src = src.cpu()  # the table is complete in HBM when the hook fires; a blocking D2H on torch's stream
            self.pending.append(dist.all_gather_into_tensor(outs[rnd], src, group=group, async_op=True))
            self.gathers_issued += 1
        self.c.set_round_hook(hook)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3600)  # ~1 s of timed window at the default configuration (18 callers x 200 calls)
    ap.add_argument("--warmup", type=int, default=180)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS), help="BASELINE.json workload, 1-based (see the module docstring)")
    ap.add_argument("--kyber-k", type=int, default=0, help="override the configuration's KYBER_K")
    ap.add_argument("--batch", type=int, default=0, help="override the configuration's proofs per GPU per step")
    ap.add_argument("--slots", type=int, default=int(os.environ.get("KOSK_BENCH_SLOTS", "0")),
                    help="independent batches kept in flight per GPU (own HIP stream + host threads each); 0 = the configuration's "
                         "default.  Never depends on --steps.")
    ap.add_argument("--combine", type=int, default=int(os.environ.get("KOSK_BENCH_COMBINE", "0")),
                    help="handles per cohort (kosk_options::combine): the resident calls of the slots of a cohort are served by one merged pipeline "
                         "run; 0 = the configuration's default, 1 = every slot on its own (the round-3 arrangement)")
    ap.add_argument("--fs", default=os.environ.get("KOSK_BENCH_FS", ""), choices=["", "host", "device"],
                    help="where the Fiat-Shamir aggregation hashes run: host (the cores hash the digest tables, which cross PCIe) or device "
                         "(one wave per proof hashes them in HBM, no host round trip inside a call); default: the configuration's")
    ap.add_argument("--threads", type=int, default=int(os.environ.get("KOSK_BENCH_THREADS", "0")), help="Fiat-Shamir workers per caller (0: the host budget's)")
    ap.add_argument("--phase-stats", action="store_true", help="mean of the library's phase clocks over every step (diagnostic; one extra ABI call per step)")
    ap.add_argument("--tape-sets", type=int, default=4, help="distinct resident tape sets per slot, rotated step by step")
    ap.add_argument("--cpu-proofs", type=int, default=12, help="bounded CPU baseline sample (about 13 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernels", action="store_true", help="skip the 65 536-lane kernel leg and the PCIe-inclusive drop-in leg")
    ap.add_argument("--leg", default="", choices=["", "drop_in"], help="(internal) run ONE auxiliary leg in this process and print its JSON object")
    args = ap.parse_args()
    cfg = dict(CONFIGS[args.config])
    if args.kyber_k:
        cfg["k"] = args.kyber_k
    if args.batch:
        cfg["batch"] = args.batch
    custom = bool(args.kyber_k or args.batch)
    k, B = cfg["k"], cfg["batch"]
    if args.leg == "drop_in":  # a child of the measuring process (below): handles, threads and page-locked buffers of its own
        import torch
        from mpcith_kyber_kosk_amd import api
        print(json.dumps(drop_in(api, torch, k, B, tapes_for(0, B, api.tape_bytes(k)), 0)))
        return
    S = args.slots if args.slots > 0 else cfg["slots"]
    CMB = args.combine if args.combine > 0 else (cfg.get("combine", 1) if not args.batch else 1)
    # host cores this rank can count on: when they are scarce (an 8-GPU node with few cores per GPU), the slots' waits sleep on
    # events instead of spinning and the Fiat-Shamir pools shrink; with 32+ cores per rank nothing changes
    usable_cores = usable_host_cores()  # scheduler affinity cut to the cgroup CPU quota (a lease shows 256 hardware threads and has 16 cores)
    local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))
    cores_per_rank = max(1, usable_cores // local_world)
    threads, blocking = host_budget(usable_cores, local_world, -(-S // CMB), cfg["threads"] * CMB)
    threads = threads_per_caller(threads, blocking, CMB, cores_per_rank)  # per handle; a merged run led by a cohort's first member uses CMB times as many
    # Where the Fiat-Shamir hashes run.  On the host they cost four to five busy cores per GPU and four PCIe copies of 46.5 KB per proof,
    # and give the higher rate when the cores exist (189-191 k proofs/s at 10-11 busy cores on one GPU); on the device a rank needs two to
    # three cores with this Python harness and makes 166-176 k with sixteen callers per cohort (below).  Host mode costs ~0.06 busy cores per
    # k proofs/s, so below a dozen cores per rank it would be CPU-bound under what device mode delivers.  Not given: the device below twelve
    # usable cores per rank (eight ranks on a 16-core quota could not even start the host's hashing), else the host.
    FS = args.fs or cfg.get("fs") or ("device" if cores_per_rank < 12 else "host")
    if FS == "device":
        # Nothing to hash on the host: ONE worker per caller (it only finishes the key records).  Three of them -- host mode's figure --
        # cost 1.9 busy cores per GPU in wake-ups for nothing (profiles/r06b_device_arrangements.txt: 48 callers 4.76 -> 2.88 cores).
        threads = 1
        # A cohort's stream waits 4 x 0.79 ms per step for its chains whatever the launch size, so device mode wants LARGE cohorts: sixteen
        # callers per merged run (736 proofs per launch) make 179 k proofs/s at 2.9 busy cores where six make 124 k at 1.8 (native callers;
        # this harness: 166-176 k at 2.8-3.2 cores, profiles/r06b_device_arrangements.txt).  Taken when neither --slots nor --combine is
        # given, at 46 proofs per call, where the rank has four cores for the 48 caller threads (with three, 36 callers made 86 k: the
        # interpreter's threads starve each other; two or three cores keep eighteen callers: 94 k at 1.9 cores).
        if args.slots <= 0 and args.combine <= 0 and not args.batch and cfg.get("combine", 1) == 6 and cores_per_rank >= 4:
            S, CMB = 48, 16
    if args.threads > 0:
        threads = args.threads
    if os.environ.get("KOSK_BENCH_BLOCKING") in ("0", "1"):
        blocking = os.environ["KOSK_BENCH_BLOCKING"] == "1"
    # stdout carries exactly ONE line (the JSON): whatever libraries print there (RCCL's version banner on the first
    # communicator, for one) goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the KOSK path has no CPU fallback")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node N" % (args.gpus, world))
    # rehearsal hook (1-GPU box): KOSK_BENCH_REHEARSE=1 runs all ranks on cuda:0 over gloo to exercise the N>1 code path
    rehearse = os.environ.get("KOSK_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    want_gather = args.config == 4 and not custom
    if world > 1 or want_gather:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:  # plain `python bench.py --config 4`: a one-rank RCCL job
            with socket.socket() as s_:
                s_.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from mpcith_kyber_kosk_amd import api, sharding
    slots = [Slot(api, torch, k, B, local_rank, ((rank * S + si) * args.tape_sets) * B, args.tape_sets, combine=CMB, prewake_us=cfg.get("prewake_us"), fs=FS,
                  threads=threads, blocking=blocking) for si in range(S)]
    ctx = slots[0].c
    tapes = slots[0].first_tapes
    if want_gather:
        for sl in slots:  # one communicator per slot: each slot issues its collectives in its own fixed order
            sl.enable_gather(api, torch, dist, dist.new_group(list(range(world))), world, host_staged=rehearse)
    for sl in slots:  # setup, not a benchmark step: first use allocates the verifier workspace and builds its tables
        sl.step(torch, 0)

    W, K = max(0, args.warmup), max(1, args.steps)
    D = S  # untimed drain: the window's last steps finish with the pipeline still full
    total = W + K + D
    if CMB > 1:
        # every caller thread makes the same number of calls (steps dealt statically, total a multiple of the slot count): the
        # cohorts stay complete to the last step instead of breaking up when a shared step counter runs out
        total = -(-total // S) * S
        D = total - W - K
    state = {"next": 0, "limit": 0, "err": None}
    lock = threading.Lock()
    done_t = []
    lat = []
    static = want_gather or CMB > 1  # collectives: every rank must run the same steps on the same slot; cohorts: see `total`

    def take(si, nth):
        """index of slot si's nth step of this run, or None when the run is over"""
        if static:
            i = si + nth * S
            return i if i < state["limit"] and state["err"] is None else None
        with lock:
            i = state["next"]
            if i >= state["limit"] or state["err"] is not None:
                return None
            state["next"] = i + 1
            return i

    def worker(si, go, fin):
        sl = slots[si]
        while True:
            go.wait(); go.clear()
            if state["limit"] < 0:
                return
            try:
                nth = 0
                while True:
                    i = take(si, nth)
                    if i is None:
                        break
                    nth += 1
                    t_a = time.perf_counter()
                    sl.step(torch, i)
                    t_b = time.perf_counter()
                    done_t.append(t_b)  # (list.append is atomic under the interpreter lock: no second lock for the callers to queue on)
                    lat.append(t_b - t_a)
            except Exception as e:  # noqa: BLE001
                with lock:
                    state["err"] = e
            fin.set()
    gos = [threading.Event() for _ in range(S)]
    fins = [threading.Event() for _ in range(S)]
    workers = [threading.Thread(target=worker, args=(si, gos[si], fins[si]), daemon=True) for si in range(S)]
    for wt in workers:
        wt.start()

    def run(nsteps):
        """nsteps whole steps through the S slots, back to back; returns (start time, completion times, step latencies)"""
        with lock:
            state["next"], state["limit"] = 0, nsteps
            del done_t[:]
            del lat[:]
        t_s = time.perf_counter()
        for g_ in gos:
            g_.set()
        for f_ in fins:
            f_.wait(); f_.clear()
        if state["err"] is not None:
            raise state["err"]
        return t_s, sorted(done_t), list(lat)

    def barrier():
        torch.cuda.synchronize()
        for sl in slots:
            sl.c.synchronize()
        if dist is not None:
            dist.barrier()

    # conditioning (setup, untimed, independent of --warmup/--steps): clocks, runtime and worker threads of a fresh
    # process ramp for a few hundred milliseconds.  Two things found with tools/tail_probe.py (round 5) end here, not
    # in the timed run: (1) the HIP runtime grows a system-memory pool by 8 MB chunks while the number of commands in flight is
    # still rising (a KFD allocation + a 4 ms SVM ioctl under a runtime-wide lock inside a launch or copy call: every caller
    # stalls 5-9 ms), three times within the first ~100 steps per caller; (2) the interpreter's cyclic collector: a full
    # collection over the objects of torch / numpy holds the interpreter lock for 35-75 ms, and every caller thread needs that
    # lock when its library call returns.  The harness's objects are collected and frozen here and the collector is off during
    # the timed run (the library never calls into Python; the harness allocates no cycles in its loop).
    import gc
    gc.collect()
    gc.freeze()
    t_cond = time.perf_counter()
    if want_gather:
        for _ in range(3):  # a fixed count: every rank must issue the same collectives
            run(2 * S)
    else:
        while time.perf_counter() - t_cond < float(os.environ.get("KOSK_BENCH_CONDITION_S", "1.0")):
            run(2 * S)
    gc_was_enabled = gc.isenabled()
    if os.environ.get("KOSK_BENCH_GC", "0") != "1":
        gc.disable()
    for sl in slots:
        sl.c.profile_enable(True)
        if args.phase_stats:
            sl.ph_sum = [0.0] * 16
            sl.steps_at_stats = sl.steps_done
    import resource
    barrier()
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    cg0 = cgroup_cpu()
    t_s, comp, lats = run(total)
    barrier()
    t_e = time.perf_counter()
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    cg1 = cgroup_cpu()
    if gc_was_enabled:
        gc.enable()
    # The timed window is exactly K steps, from the completion of step W to the completion of step W + K.  The members of a
    # cohort complete together and the cohorts of a run tend to stay in phase, so completions come in bursts of up to S steps:
    # one window of a few tens of steps lands anywhere between "just before a burst" and "just after one" (+-10 % at the driver's
    # --steps 20).  The mean over the S adjacent windows W .. W + S - 1 (all of exactly K steps, all inside the run: D >= S steps
    # follow the first window) takes that alignment out; with hundreds of steps it changes nothing.
    nwin = max(1, min(S, D))
    dts = [comp[j + K - 1] - (comp[j - 1] if j > 0 else t_s) for j in range(W, W + nwin)]
    dt = sum(dts) / len(dts)
    dt_drained = t_e - t_s
    host_cpu_s = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
    prof = {}
    for sl in slots:
        for name, (ms, cnt_, units_) in sl.c.profile_read_units().items():
            a, b, u = prof.get(name, (0.0, 0, 0))
            prof[name] = (a + ms, b + cnt_, u + units_)
        sl.c.profile_enable(False)
    cstats = [sl.c.combine_stats() for sl in slots]
    phases = ctx.phase_seconds()
    gather_info = None
    if want_gather:
        # the gathered tables: this rank's block must be its own digest table (the verifier rebuilt the same table)
        sl = slots[0]
        _, _, outs, _ = sl.gather
        for w_ in sl.pending:
            w_.wait()
        torch.cuda.synchronize()
        mine = [torch.as_tensor(sl.c.resident_digests(r, B), device=sl.bank.device).to(outs[r].device) for r in (0, 1)]
        same = all(bool(torch.equal(outs[r].view(world, B, 1454, 32)[rank], mine[r])) for r in (0, 1))
        if not same:
            raise RuntimeError("all-gathered digest table differs from the resident one")
        # every rank's block of MY gathered tables must be what THAT rank holds: one sha3_256 per (rank, round), all-gathered
        peers_ok = True
        for r in (0, 1):
            own = torch.frombuffer(bytearray(hashlib.sha3_256(mine[r].cpu().numpy().tobytes()).digest()), dtype=torch.uint8).reshape(1, 32).to(outs[r].device)
            table = sharding.allgather_digest_table(own, world, dist)
            for q_ in range(world):
                blk = outs[r].view(world, B, 1454, 32)[q_].cpu().numpy().tobytes()
                peers_ok &= bytes(table[q_].tolist()) == hashlib.sha3_256(blk).digest()
        if not peers_ok:
            raise RuntimeError("a peer's block of the all-gathered digest table differs from that peer's resident table")
        gather_info = {"collective": "all_gather_into_tensor (%s)" % ("gloo, host-staged rehearsal" if rehearse else "RCCL"),
                       "tables_per_step": 2, "bytes_per_rank_per_table": B * 1454 * 32,
                       "gathered_shape": list(outs[0].shape), "own_block_matches_resident_table": True,
                       "peer_blocks_match_their_resident_tables": True,
                       "gathers_issued_per_slot": [s_.gathers_issued for s_ in slots], "steps_per_slot": [s_.steps_done for s_ in slots]}
    if dist is not None:
        cdev = "cpu" if rehearse else "cuda"
        t = torch.tensor([dt, dt_drained], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dt_drained = float(t[0].item()), float(t[1].item())
        # who took part: every rank's (rank, local rank, device ordinal, PCI bus / device id) all-gathered over the job's own backend
        # (RCCL on device tensors unless this is the one-GPU gloo rehearsal), so that the line proves N ranks on N devices
        try:
            pr_ = torch.cuda.get_device_properties(local_rank)
            me_ = torch.tensor([[rank, local_rank, torch.cuda.current_device(), int(getattr(pr_, "pci_bus_id", -1)), int(getattr(pr_, "pci_device_id", -1))]],
                               dtype=torch.int64, device=cdev)
            all_ = torch.empty((world, 5), dtype=torch.int64, device=cdev)
            dist.all_gather_into_tensor(all_, me_)
            rccl_info = {"world": world, "backend": dist.get_backend(), "devices": [{"rank": int(r_[0]), "local_rank": int(r_[1]), "device": int(r_[2]),
                                                                                     "pci_bus_id": int(r_[3]), "pci_device_id": int(r_[4])} for r_ in all_.cpu().tolist()]}
        except Exception as e:  # noqa: BLE001
            rccl_info = {"world": world, "error": repr(e)[:300]}
        # every rank's host side: busy cores, usable cores (affinity cut to the cgroup quota), CFS throttling inside the run -- the
        # quantities that decide an N-GPU curve on a container with a CPU quota (rank 0's own figures are top-level fields of the line)
        try:
            hs_ = torch.tensor([[rank, host_cpu_s / max(dt_drained, 1e-9), usable_cores, cg1[0] if cg1[0] is not None else -1.0,
                                 (cg1[1] - cg0[1]) if cg1[1] is not None and cg0[1] is not None else -1.0,
                                 (cg1[2] - cg0[2]) * 1e3 if cg1[2] is not None and cg0[2] is not None else -1.0]], dtype=torch.float64, device=cdev)
            hall_ = torch.empty((world, 6), dtype=torch.float64, device=cdev)
            dist.all_gather_into_tensor(hall_, hs_)
            host_per_rank = [{"rank": int(r_[0]), "host_cpu_cores_busy": round(r_[1], 2), "host_cores_usable": int(r_[2]),
                              "cgroup_quota_cores": None if r_[3] < 0 else r_[3], "throttled_periods_in_run": None if r_[4] < 0 else int(r_[4]),
                              "throttled_ms_in_run": None if r_[5] < 0 else round(r_[5], 3)} for r_ in hall_.cpu().tolist()]
        except Exception as e:  # noqa: BLE001
            host_per_rank = {"error": repr(e)[:300]}
        if not want_gather:
            # result gather (not on the data path): one digest per rank over its last batch of proofs
            pr = slots[0].c.fetch_proofs(B)
            mine = torch.frombuffer(bytearray(hashlib.sha3_256(b"".join(pr)).digest()), dtype=torch.uint8).reshape(1, 32).to(cdev)
            table = sharding.allgather_digest_table(mine, world, dist)
            assert table.shape == (world, 32) and bytes(table[rank].tolist()) == bytes(mine[0].tolist())

    if rank == 0:
        kern = {}
        for name, (ms, cnt_, units_) in prof.items():
            if cnt_:
                kern[name] = {"avg_us": ms / cnt_ * 1e3, "launches": cnt_, "total_ms": ms, "proofs_per_launch": units_ / cnt_}
        # graded kernel: the SHA3-256 view commitment (K4); algorithmic bytes = message + digest per party lane
        hv = kern.get("hash_view")
        roof = None
        if hv:
            # every launch of the view-commitment kernel inside the timed run, whatever it served: a merged run of a cohort hashes
            # the batches of 1..C callers with one launch (`proofs_per_launch` is the mean).  achieved = bytes those launches served / time they took.
            ppl = hv["proofs_per_launch"]
            lanes_per_launch = ppl * 1454
            nbytes = lanes_per_launch * (VIEW_MSG[k] + 32)
            ach = nbytes / (hv["avg_us"] * 1e-6) / 1e9
            traffic, tsrc, extrap = None, None, False
            tfile = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tfile) and k == 3 and B == 46:
                tj = json.load(open(tfile))
                per_lane = tj.get("hash_view_hbm_bytes_per_lane")
                if per_lane:  # PMC bytes of the same kernel at a 138-proof launch (tools/pmc_workload.py), per party lane
                    traffic = int(round(per_lane * lanes_per_launch))
                    pmc_lanes = tj.get("hash_view_lanes_per_launch", 0)
                    extrap = abs(pmc_lanes - lanes_per_launch) > 0.02 * lanes_per_launch
                    tsrc = "%s; measured on a %d-lane launch%s" % (tj.get("source"), pmc_lanes, ", scaled by lanes to this run's mean launch" if extrap else " (this run's launch size)")
            roof = {"kernel": "k_commit_hash_dma (SHA3-256 view commitment, prover; %.1f callers' batches = %.0f party lanes per launch on average)"
                              % (ppl / B, lanes_per_launch),
                    "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "traffic": traffic, "traffic_source": tsrc, "traffic_extrapolated": bool(traffic is not None and extrap), "algorithmic_bytes_per_launch": nbytes, "avg_launch_us": hv["avg_us"],
                    "lanes_per_launch": lanes_per_launch, "launches": hv["launches"],
                    "note": "HIP events on the stream of the handle that led the (merged) run, inside the timed run; %d handles in %d cohorts "
                            "share the GPU, so a launch's duration includes co-running kernels of other runs" % (S, -(-S // CMB))}
            ht = kern.get("hash_tcomm")
            if ht:
                ht["GBps"] = ht["proofs_per_launch"] * 1454 * (TCOMM_MSG[k] + 32) / (ht["avg_us"] * 1e-6) / 1e9
            hv["GBps"] = ach
        g1 = kern.get("gemm_expand1")
        if g1:
            rows = {2: 214 - 6, 3: 226 - 9, 4: 254 - 12}[k]
            g1["useful_GMACps"] = g1["proofs_per_launch"] * rows * 1303 * 407 / (g1["avg_us"] * 1e-6) / 1e9
        lats.sort()
        line = {
            "metric": "kyber768_kosk_proofs_per_sec_prove_plus_verify" if k == 3 else "kyber%d_kosk_proofs_per_sec_prove_plus_verify" % (256 * k),
            "value": world * K * B / dt, "unit": "proofs/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u16", "data": "synthetic",
            "config": {"workload": (cfg["what"] if not custom else "KYBER_K=%d, %d proofs per GPU per step" % (k, B)) +
                                   ": kyber_verifiable_keygen (GPU key generation + offline + online prover) + kyber_kosk_verify (pk decoding + "
                                   "verifier), randomness tapes resident in HBM, a different tape set every step",
                       "baseline_config": "configs[%d]" % (args.config - 1), "kyber_k": k, "proofs_per_gpu": B, "party_lanes_per_gpu": B * 1454,
                       "sharding": "by proof", "pipeline_slots_per_gpu": S, "handles_per_cohort": CMB,
                       "call_combining": ("kosk_options::combine = %d: %d caller threads, one handle and one %d-proof call each; the library serves the "
                                          "calls of a cohort of %d handles with one pipeline run" % (CMB, S, B, CMB)) if CMB > 1 else "off",
                       "fiat_shamir": FS + (": one wave per proof hashes each digest table in HBM (k_fs_chain), alpha / I / the verifier's I' == I stay on "
                                            "the device, no digest table crosses PCIe and a resident call has no host round trip" if FS == "device" else
                                            ": the host's cores hash the four 46.5 KB digest tables per proof, which cross PCIe (four host round trips per step)"),
                       "tape_sets_per_slot": args.tape_sets,
                       "host_threads_per_slot": threads, "host_waits": "sleep" if blocking else "nap + spin",
                       "timing": "steady-state window of exactly K steps (completion of step W to completion of step W+K, slots running continuously), "
                                 "mean over the %d adjacent window positions W..W+%d: completions come in bursts of a cohort's calls" % (nwin, nwin - 1)},
            "drained_run": {"steps": total, "ms_per_step": dt_drained / total * 1e3, "value": world * total * B / dt_drained,
                            "note": "the same run from first issue to last completion, barrier + synchronize on both sides (fill and drain included)"},
            "step_latency_ms": {"median": lats[len(lats) // 2] * 1e3, "p90": lats[int(len(lats) * 0.9)] * 1e3,
                                "p99": lats[int(len(lats) * 0.99)] * 1e3, "max": lats[-1] * 1e3, "mean": sum(lats) / len(lats) * 1e3,
                                "mean_in_keygen_call": sum(s_.t_prove for s_ in slots) / max(1, sum(s_.steps_done for s_ in slots)) * 1e3,
                                "mean_in_verify_call": sum(s_.t_verify for s_ in slots) / max(1, sum(s_.steps_done for s_ in slots)) * 1e3,
                                "per_cohort_mean": [round(sum(s_.t_prove + s_.t_verify for s_ in slots[c0:c0 + max(1, CMB)]) /
                                                          max(1, sum(s_.steps_done for s_ in slots[c0:c0 + max(1, CMB)])) * 1e3, 3) for c0 in range(0, S, max(1, CMB))],
                                "note": "one slot's keygen + prove + verify of its %d proofs while the other slots run; per_cohort_mean: the "
                                        "same per cohort (every caller makes the same number of calls, so the slowest cohort sets the run's length)" % B},
            "roofline": roof,
            "combining": {"calls": sum(c_[0] for c_ in cstats), "mean_callers_per_run": (sum(c_[1] for c_ in cstats) / max(1, sum(c_[0] for c_ in cstats))),
                          "note": "resident calls that went through the combiner since the handles were created, and the mean number of "
                                  "callers served by the run a call ended up in"} if CMB > 1 else None,
            # what the rate was bought with: the proofs in flight per GPU, the latency of one caller's call pair, and what the harness
            # does with its interpreter during the timed run -- next to `value`, not only inside `config` / `harness`
            "proofs_in_flight_per_gpu": S * B, "call_pair_latency_ms_median": lats[len(lats) // 2] * 1e3,
            "python_gc_in_timed_run": os.environ.get("KOSK_BENCH_GC", "0") == "1",
            "host_cpu_cores_busy": round(host_cpu_s / max(dt_drained, 1e-9), 2),
            "host_cpus_usable": usable_cores, "host_cpus_affinity": len(os.sched_getaffinity(0)), "host_cores_per_rank": cores_per_rank,
            "harness": {"python_gc_in_timed_run": os.environ.get("KOSK_BENCH_GC", "0") == "1",
                        "note": "the interpreter's cyclic collector is off during the timed run (objects collected and frozen before the conditioning "
                                "runs): a full collection holds the interpreter lock 35-75 ms and every caller thread needs it when its library "
                                "call returns (tools/tail_probe.py, DESIGN 15.11); KOSK_BENCH_GC=1 leaves it on"},
            "cgroup_cpu": {"quota_cores": cg1[0], "throttled_periods_in_run": (cg1[1] - cg0[1]) if cg1[1] is not None and cg0[1] is not None else None,
                           "throttled_ms_in_run": round((cg1[2] - cg0[2]) * 1e3, 3) if cg1[2] is not None and cg0[2] is not None else None,
                           "note": "CPU quota of the container and how often / how long the kernel stalled its threads during the timed run (cpu.stat)"},
            "kernels_in_pipeline": kern,
            "profiled_kernel_ms_per_step": round(sum(v["total_ms"] for v in kern.values()) / max(1, (hv or {}).get("launches", 0)), 4),
            "prove_phase_ms": dict(zip(["host_pre", "gpu_commit", "fs_alpha_host", "gpu_relation", "fs_open_host", "gpu_assemble", "d2h"],
                                       [round(x * 1e3, 3) for x in phases])),
            "prove_issue_ms": dict(zip(["p1", "p2", "p3"], [round(x * 1e3, 3) for x in phases[7:10]])),
            "verify_phase_ms": dict(zip(["issue1", "wait1", "fs_alpha_host", "issue2", "wait2", "fs_open_host_and_masks"],
                                        [round(x * 1e3, 3) for x in phases[10:16]])),
        }
        if args.phase_stats:
            names = ["host_pre", "gpu_commit", "fs_alpha_host", "gpu_relation", "fs_open_host", "gpu_assemble", "d2h", "issue_p1", "issue_p2", "issue_p3",
                     "v_issue1", "v_wait1", "v_fs_alpha_host", "v_issue2", "v_wait2", "v_fs_open_host_and_masks"]
            nst = max(1, sum(s_.steps_done - s_.steps_at_stats for s_ in slots))
            line["phase_means_ms"] = {nm: round(sum(s_.ph_sum[i_] for s_ in slots) / nst * 1e3, 4) for i_, nm in enumerate(names)}
        if gather_info:
            line["digest_allgather"] = gather_info
        if dist is not None:
            line["rccl"] = rccl_info
            line["host_per_rank"] = host_per_rank
        if args.config == 5 and not custom:
            # batch-of-1 latency: one verifiable keygen + verify alone on an idle GPU
            c1 = api.Kosk(kyber_k=k, max_batch=1, device=local_rank)
            t1s = []
            tp1 = tapes_for(10 ** 6, 12, c1.tape_bytes)
            for i in range(12):
                ta = time.perf_counter()
                c1.verifiable_keygen_resident([tp1[i]])
                assert c1.verify_resident_pk(1) == [True]
                t1s.append(time.perf_counter() - ta)
            t1s = sorted(t1s[2:])
            c1.close()
            line["latency"] = {"batch_of_512_ms_median": lats[len(lats) // 2] * 1e3, "batch_of_1_ms_median": t1s[len(t1s) // 2] * 1e3,
                               "per_proof_us_at_throughput": dt / K / B * 1e6}
        aux = world == 1 and not args.no_kernels and args.config in (2, 3) and not custom
        if aux:
            try:
                line["kernels_65536_lanes"] = kernel_microbench(ctx, torch, k)
                if line.get("roofline") is not None:
                    # the graded kernels at exactly 65 536 lanes (BASELINE.json configs[1..2]) where the driver's record keeps them:
                    # SHA3 view / Tcomm commitment and NTT-256, 20 launches between two HIP events each, outputs checked
                    line["roofline"]["graded_65536"] = {n_: {f_: v_ for f_, v_ in line["kernels_65536_lanes"][n_].items()
                                                             if f_ in ("lanes", "polys", "msg_bytes", "us", "GBps", "frac_hbm_peak", "keccak_f_per_s", "checked")}
                                                        for n_ in ("sha3_view", "sha3_tcomm", "ntt256")}
            except Exception as e:  # noqa: BLE001  -- an auxiliary leg never costs the line of record
                line["kernels_65536_lanes"] = {"error": repr(e)[:400]}
    # the measured run is over: stop the slot threads and free the slots' HBM, host threads and streams BEFORE the auxiliary legs
    state["limit"] = -1
    for g_ in gos:
        g_.set()
    if dist is None:
        for sl in slots:
            sl.c.close()
    if rank == 0:
        if aux:
            # the host-buffer leg runs in a child process as well: it drives KOSK_STREAMS=3 handles from two threads over caller
            # memory, and nothing that could go wrong there may take the measured line with it
            try:
                import subprocess
                env_ = {k_: v_ for k_, v_ in os.environ.items() if k_ not in ("KOSK_BENCH_THREADS", "KOSK_BENCH_BLOCKING", "KOSK_BENCH_COMBINE", "KOSK_BENCH_SLOTS", "KOSK_BENCH_FS")}
                r_ = subprocess.run([sys.executable, os.path.abspath(__file__), "--leg", "drop_in", "--config", str(args.config)], stdout=subprocess.PIPE,
                                    stderr=subprocess.PIPE, text=True, timeout=600, env=env_)
                sub_ = [ln for ln in r_.stdout.splitlines() if ln.startswith("{")]
                if r_.returncode != 0 or not sub_:
                    raise RuntimeError("rc %d: %s" % (r_.returncode, (r_.stderr or "")[-300:]))
                line["drop_in"] = json.loads(sub_[-1])
            except Exception as e:  # noqa: BLE001
                line["drop_in"] = {"error": repr(e)[:400]}
            # side runs as fresh child processes with their own host-thread budget, reported NEXT TO the line of record, never as
            # `value`: the same workload WITHOUT call combining (every handle on its own: the round-3 arrangement, 6 slots), and the
            # lower-latency arrangements of rounds 4 and 5a (cohorts of three / four)
            import subprocess
            env = {k_: v_ for k_, v_ in os.environ.items() if k_ not in ("KOSK_BENCH_THREADS", "KOSK_BENCH_BLOCKING", "KOSK_BENCH_COMBINE", "KOSK_BENCH_SLOTS", "KOSK_BENCH_FS")}

            def side_run(slots_, combine_, note, threads_=6, fs_=None):  # threads_: Fiat-Shamir workers per caller (the smaller arrangements keep the six they were tuned with)
                cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--config", str(args.config), "--fs", fs_ or FS, "--combine", str(combine_), "--slots", str(slots_),
                       "--steps", str(max(20, K // 2) // slots_ * slots_ + slots_), "--warmup", str(max(4, W // 2)), "--no-kernels", "--no-cpu-baseline"]
                try:
                    r = subprocess.run(cmd + (["--threads", str(threads_)] if threads_ else []), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env)
                    sub = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                    if r.returncode != 0 or not sub:
                        raise RuntimeError("rc %d: %s" % (r.returncode, (r.stderr or "")[-300:]))
                    j = json.loads(sub[-1])
                    return {"proofs_per_s": j["value"], "ms_per_step": j["ms_per_step"], "slots": slots_, "handles_per_cohort": combine_, "steps": j["steps"],
                            "roofline_frac": (j.get("roofline") or {}).get("frac"),
                            "hash_view_avg_us": (j.get("kernels_in_pipeline", {}).get("hash_view") or {}).get("avg_us"),
                            "hash_view_proofs_per_launch": (j.get("kernels_in_pipeline", {}).get("hash_view") or {}).get("proofs_per_launch"),
                            "step_latency_ms_median": (j.get("step_latency_ms") or {}).get("median"),
                            "host_cpu_cores_busy": j.get("host_cpu_cores_busy"), "fiat_shamir": fs_ or FS, "note": note}
                except Exception as e:  # noqa: BLE001  (TimeoutExpired, a missing key, bad JSON ...)
                    return {"error": repr(e)[:400]}
            # one cohort ALONE on the GPU (three callers, one merged run in flight): the graded kernel's launch time without co-running kernels
            oc_ = side_run(CMB, CMB, threads_=0, note="python bench.py --combine %d --slots %d: ONE cohort alone on the GPU; its view-commitment launches are not stretched by "
                                     "other cohorts' kernels; not the line of record" % (CMB, CMB))
            line["one_cohort_alone"] = oc_
            if line.get("roofline") is not None and oc_.get("hash_view_avg_us"):
                ppl_ = oc_.get("hash_view_proofs_per_launch") or 0
                line["roofline"]["alone_us"] = oc_["hash_view_avg_us"]
                line["roofline"]["alone_proofs_per_launch"] = ppl_
                line["roofline"]["alone_frac"] = ppl_ * 1454 * (VIEW_MSG[k] + 32) / (oc_["hash_view_avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS
            # the same arrangement driven by a NATIVE caller loop (examples/throughput.cpp: std::thread callers on the C ABI, the same
            # tape banks, every verify bit asserted) as a child process: what a C++ host -- the reference is one -- would run; proofs/s
            # and busy host cores without the interpreter lock.  Reported next to `value`, never instead of it.
            def native_run(fs_, callers_=None, combine_=None):
                exe = os.path.join(ROOT, "examples", "throughput")
                if not os.path.exists(exe):
                    return {"error": "examples/throughput is not built (__graft_entry__.build() builds it)"}
                cmd = [exe, "--k", str(k), "--batch", str(B), "--callers", str(callers_ or S), "--combine", str(combine_ or CMB), "--fs", fs_, "--threads", str(1 if fs_ == "device" else threads),
                       "--steps", str(max(K, 20)), "--warmup", str(W), "--tape-sets", str(args.tape_sets), "--device", str(local_rank)]
                try:
                    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env)
                    sub = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                    if r.returncode != 0 or not sub:
                        raise RuntimeError("rc %d: %s" % (r.returncode, (r.stderr or "")[-300:]))
                    return json.loads(sub[-1])
                except Exception as e:  # noqa: BLE001
                    return {"error": repr(e)[:400]}
            line["native_callers"] = native_run(FS)
            other_ = "host" if FS == "device" else "device"
            line["native_callers_fs_" + other_] = native_run(other_)
            # device mode's fixed cost per round (a 0.78 ms chain) does not grow with the launch: sixteen callers per cohort (736 proofs per
            # launch; the combiner's limit) is where it reaches the host mode's rate, at under five busy cores and 12 ms per call pair
            line["native_callers_fs_device_cohorts_of_16"] = native_run("device", callers_=48, combine_=16)
            line["fiat_shamir_" + other_] = side_run(S, CMB, threads_=0, fs_=other_, note="python bench.py --fs %s: the line of record's arrangement with the Fiat-Shamir "
                                                     "hashes on the %s; not the line of record" % (other_, other_))
            line["uncombined"] = side_run(6, 1, "python bench.py --combine 1 --slots 6: six independent handles, every launch serves one 46-proof call "
                                                "(round 3's line of record); not the line of record")
            line["cohorts_of_three"] = side_run(9, 3, "python bench.py --combine 3 --slots 9: nine callers, three per merged run (138 proofs per launch): round 4's line of "
                                                      "record (147.6 k proofs/s at 2.8 ms there), on this box for a like-for-like comparison; not the line of record")
            line["cohorts_of_four"] = side_run(12, 4, "python bench.py --combine 4 --slots 12: twelve callers, four per merged run (184 proofs per launch; less latency "
                                                      "per call, fewer proofs per launch than the default's six); not the line of record")
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(k, tapes, min(args.cpu_proofs, B))
            except Exception as e:  # noqa: BLE001
                line["cpu_baseline"] = {"error": repr(e)[:400]}
        os.write(json_fd, (json.dumps(line) + "\n").encode())  # written here: the RCCL teardown below can end the process without Python's exit flush
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
        for sl in slots:
            sl.c.close()


if __name__ == "__main__":
    main()
