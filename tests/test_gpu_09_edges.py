"""GPU edge cases of the batch ABI: empty and ragged batches, the chunk-parallel host-buffer path (kosk_options::streams > 1), and
batches larger than the context (chunking), all against the single-context path and the CPU oracle, bit for bit."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: torch.cuda.is_available() is False")
    return torch


def test_empty_batches_are_no_ops(oracle, torch_cuda):
    from mpcith_kyber_kosk_amd import api
    ctx = api.Kosk(kyber_k=3, max_batch=2)
    lib, h = api.lib, ctx.handle
    guard = C.create_string_buffer(b"\xa5" * 64)
    assert lib.kosk_verifiable_keygen_batch(h, 0, None, 0, guard, guard, guard) == 0
    assert lib.kosk_verify_batch(h, 0, guard, guard, guard) == 0
    assert lib.kosk_verifiable_keygen_resident(h, 0, None, 0, guard, guard) != 0  # the resident calls want 1..max_batch
    assert b"kosk_verifiable_keygen_resident" in lib.kosk_last_error(h)
    assert guard.raw[:64] == b"\xa5" * 64  # nothing written
    # a negative count is an error, not a crash; the context stays usable afterwards
    assert lib.kosk_verify_batch(h, -1, guard, guard, guard) != 0
    assert b"kosk_verify_batch" in lib.kosk_last_error(h)
    tape = [oracle.tape_bytes_for(3, 0)]
    pks, sks, pis = ctx.verifiable_keygen(tape)
    assert ctx.verify(pis, pks) == [True]
    ctx.close()


@pytest.mark.parametrize("k", [2, 3])
def test_streamed_chunks_equal_single_context_and_oracle(k, torch_cuda, gpu_child):
    """KOSK_STREAMS=3 host-buffer path (tests/gpu_child_cases.py: streamed_chunks) in a fresh child process: lane threads, page-locked
    caller memory and three live handles must not be able to take the rest of the suite down."""
    out = gpu_child("from tests.gpu_child_cases import streamed_chunks; streamed_chunks(%d)" % k)
    assert "streamed_chunks ok %d" % k in out


def test_streamed_calls_in_a_loop_over_live_handles(torch_cuda, gpu_child):
    """The round-2 SIGABRT's territory 120 times over (tests/gpu_child_cases.py: streamed_loop)."""
    out = gpu_child("from tests.gpu_child_cases import streamed_loop; streamed_loop(3, 120)")
    assert "streamed_loop ok 3 120" in out


def test_pinned_caller_buffers_are_copied_directly(torch_cuda, gpu_child):
    out = gpu_child("from tests.gpu_child_cases import pinned_buffers; pinned_buffers(3)")
    assert "pinned_buffers ok 3" in out


def test_host_paths_under_glibc_heap_checking(torch_cuda, gpu_child):
    """The multi-handle cases and the big batch shapes once more in a child whose allocator checks every free()
    (MALLOC_CHECK_=3: abort with a message on a corrupted chunk header; MALLOC_PERTURB_: freed and fresh memory is filled, so a
    use-after-free or an uninitialised read changes bytes that the cases compare with the oracle)."""
    code = ("from tests.gpu_child_cases import *; streamed_chunks(3); pinned_buffers(2); streamed_loop(2, 24); big_batches(); "
            "print('heap-checked ok')")
    out = gpu_child(code, env={"MALLOC_CHECK_": "3", "MALLOC_PERTURB_": "165", "LIBC_FATAL_STDERR_": "1"})
    assert "heap-checked ok" in out


def test_abi_errors_are_return_codes(torch_cuda, gpu_child):
    out = gpu_child("from tests.gpu_child_cases import errors_do_not_kill; errors_do_not_kill(2)")
    assert "errors_do_not_kill ok" in out


def test_batch_of_one_on_a_large_context_and_full_context(oracle, torch_cuda):
    """n = 1 on a context sized for 46 (the bench's slot) and n = max_batch exactly."""
    from mpcith_kyber_kosk_amd import api
    k = 3
    ctx = api.Kosk(kyber_k=k, max_batch=46)
    t = [oracle.tape_bytes_for(k, 7)]
    pks, sks, pis = ctx.verifiable_keygen(t)
    opk, osk, opi, _, _ = oracle.verifiable_keygen(k, t[0])
    assert (pks[0], sks[0], pis[0]) == (opk, osk, opi)
    assert ctx.verify(pis, pks) == [True]
    ctx.close()
    ctx = api.Kosk(kyber_k=k, max_batch=5)
    tapes = [oracle.tape_bytes_for(k, 60 + b) for b in range(5)]
    pks, sks, pis = ctx.verifiable_keygen(tapes)
    opk, osk, opi, _, _ = oracle.verifiable_keygen(k, tapes[4])
    assert (pks[4], sks[4], pis[4]) == (opk, osk, opi)
    assert ctx.verify(pis, pks) == [True] * 5
    ctx.close()


@pytest.mark.parametrize("k", [2, 3])
def test_degenerate_randomness_tapes(k, oracle, torch_cuda):
    """Tapes of constant bytes: every BE16 % q draw, every seed and every sampler sees its extreme input (0x0000 and 0xFFFF words,
    identical seeds for d, z and all PRF keys, all-equal share randoms).  Byte for byte against the oracle, and verified."""
    from mpcith_kyber_kosk_amd import api
    ctx = api.Kosk(kyber_k=k, max_batch=4)
    n = ctx.tape_bytes
    tapes = [b"\x00" * n, b"\xff" * n, (b"\x0d\x00" * n)[:n], bytes((i * 251 + 7) & 0xFF for i in range(n))]
    pks, sks, pis = ctx.verifiable_keygen(tapes)
    for b in range(len(tapes)):
        opk, osk, opi, _, _ = oracle.verifiable_keygen(k, tapes[b])
        assert pks[b] == opk and sks[b] == osk, b
        assert pis[b] == opi, b
    assert ctx.verify(pis, pks) == [True] * len(tapes)
    ctx.close()


KNOBS = {  # handle setting -> (path counter that must be > 0 on that handle, counter that must stay 0 there)
    "KOSK_WAIT_NAP=0": (None, None),        # debug knob (environment): waits spin throughout
    "KOSK_GRAPHS=1": ("graph_replay", None),  # debug knob: the segments between host rounds as replayed hipGraphs
    "blocking_sync=1": (None, None),          # option: the handle's host waits sleep on events
    "host_threads=2": (None, None),           # option: two Fiat-Shamir workers
    "fs_mode=1": ("fs_device", "fs_host"),    # option: the Fiat-Shamir hashes on the device
}


@pytest.mark.parametrize("knob", list(KNOBS))
def test_handle_settings_do_not_change_results(knob, oracle, torch_cuda, monkeypatch):
    """The handle options (kosk_options, lower case) and the two debug knobs that remain in the environment (upper case)
    select another way of waiting, of launching or another place for the Fiat-Shamir hashes, never other bytes: proofs, keys and verify
    bits equal the default handle's (which the other tests pin to the oracle), for K = 3 and a K = 4 spot check.  kosk_path_count proves
    which path ran.  (The losing kernel variants that rounds 2-5 kept behind knobs are gone: DESIGN.md 16.5.)"""
    from mpcith_kyber_kosk_amd import api
    name, val = knob.split("=")
    must, must_not = KNOBS[knob]
    for k, n in ((3, 3), (4, 1)):
        tapes = [oracle.tape_bytes_for(k, 500 + b) for b in range(n)]
        base = api.Kosk(kyber_k=k, max_batch=n)
        ref = base.verifiable_keygen(tapes)
        pc0 = base.path_counts()
        assert pc0["table_gemm"] > 0 and pc0["hash_dma"] > 0 and pc0["graph_replay"] == 0 and pc0["hash_plain"] == 0
        assert pc0["fs_host"] > 0 and pc0["fs_device"] == 0 and pc0["digest_copy"] > 0  # the default: the host hashes the digest tables
        assert pc0["small_copy_kernel"] > 0  # challenge vectors, opened lists, key records and fail masks move by kernel
        if name.islower():
            ctx = api.Kosk(kyber_k=k, max_batch=n, **{name: int(val)})
        else:
            monkeypatch.setenv(name, val)
            ctx = api.Kosk(kyber_k=k, max_batch=n)
            monkeypatch.delenv(name)
        # GRAPHS captures at the resident split (the keygen-in-front call never captures its first segment, but P1B.. do)
        got = ctx.verifiable_keygen(tapes)
        assert got == ref
        assert ctx.verify(got[2], got[0]) == [True] * n
        if knob == "KOSK_GRAPHS=1":
            got2 = ctx.verifiable_keygen(tapes)   # second call replays the captured segments
            assert got2 == ref
        if knob == "host_threads=2":
            assert ctx.host_threads == 2
        pc = ctx.path_counts()
        if must:
            assert pc[must] > 0, (knob, pc)
        if must_not:
            assert pc[must_not] == 0, (knob, pc)
        # the default handle created earlier is unaffected (per handle, not per process)
        assert base.verifiable_keygen(tapes) == ref and base.path_counts()["hash_plain"] == 0
        base.close()
        bad = bytearray(got[2][0]); bad[oracle.params(k).off[0] + 7] ^= 4  # an f share of an opened party: always read
        assert ctx.verify([bytes(bad)], [got[0][0]]) == [False]
        ctx.close()


def test_retired_environment_knobs_are_ignored(oracle, torch_cuda, monkeypatch):
    """what a host decides per handle is in kosk_options since round 6: the variables rounds 2-5 read at kosk_create no longer change anything"""
    from mpcith_kyber_kosk_amd import api
    for name, val in (("KOSK_STREAMS", "3"), ("KOSK_COMBINE", "3"), ("KOSK_STRICT_ENCODING", "0"), ("KOSK_FS_DEVICE", "1"), ("KOSK_HOST_THREADS", "1"),
                      ("KOSK_LINCOMB_FUSED", "2"), ("KOSK_ASSEMBLE_GROUPS", "0"), ("KOSK_DIGEST_DIRECT", "1"), ("KOSK_NTT_FP32", "1")):
        monkeypatch.setenv(name, val)
    ctx = api.Kosk(kyber_k=2, max_batch=2)
    tapes = [oracle.tape_bytes_for(2, 0)]
    pks, sks, pis = ctx.verifiable_keygen(tapes)
    assert oracle.verifiable_keygen(2, tapes[0])[:3] == (pks[0], sks[0], pis[0])
    pc = ctx.path_counts()
    assert ctx.streams == 1 and ctx.combine_stats() == (0, 0) and pc["fs_host"] > 0 and pc["fs_device"] == 0, pc
    ctx.close()


def test_graphs_with_alternating_device_tape_buffers(oracle, torch_cuda, monkeypatch):
    """KOSK_GRAPHS=1 + kosk_stage_prover_inputs / kosk_prove_resident with tapes read IN PLACE from two different device
    buffers, then from host memory: the captured first segment must never replay against the previous call's tape pointer
    (the tape buffer is part of the segment graph's key)."""
    torch = torch_cuda
    import numpy as np
    from mpcith_kyber_kosk_amd import api
    import ctypes as C
    k, n = 3, 2
    monkeypatch.setenv("KOSK_GRAPHS", "1")
    ctx = api.Kosk(kyber_k=k, max_batch=n)
    monkeypatch.delenv("KOSK_GRAPHS")
    stride = (ctx.tape_bytes + 63) // 64 * 64
    sets = [[oracle.tape_bytes_for(k, 700 + 10 * s + b) for b in range(n)] for s in range(3)]
    devs = []
    for tp in sets[:2]:
        host = np.zeros((n, stride), np.uint8)
        for b, t in enumerate(tp):
            host[b, :len(t)] = np.frombuffer(t, np.uint8)
        devs.append(torch.from_numpy(host).cuda())
    pk = C.create_string_buffer(ctx.pk_bytes * n); sk = C.create_string_buffer(ctx.sk_bytes * n)
    order = [0, 1, 0, 0, 1, 2, 0]   # 2 = host tapes (uploaded into the library's own buffer)
    for s in order:
        if s < 2:
            r = api.lib.kosk_stage_prover_inputs(ctx.handle, n, C.c_void_p(devs[s].data_ptr()), stride, pk, sk)
        else:
            r = api.lib.kosk_stage_prover_inputs(ctx.handle, n, C.c_char_p(b"".join(sets[2])), ctx.tape_bytes, pk, sk)
        assert r == 0, api.lib.kosk_last_error(ctx.handle)
        ctx.prove_resident(n)
        pis = ctx.fetch_proofs(n)
        for b in range(n):
            assert pis[b] == oracle.verifiable_keygen(k, sets[s][b])[2], (s, b)
    assert ctx.path_counts()["graph_replay"] > 0
    ctx.close()


def test_xof_block_limit_is_an_error_code_not_an_abort_or_a_wrong_matrix(oracle, torch_cuda, monkeypatch):
    """gen_matrix on the GPU squeezes at most 32 SHAKE128 blocks per matrix entry (indcpa.c:124-145 loops without a bound; the GPU
    loop needs an exit).  Reaching the limit must be rc -1 with text -- never a partly written matrix, never an abort -- for the
    prover's key generation and for the verifier's pk decoding.  KOSK_DEBUG_XOF_BLOCKS=1 forces it (one block holds at most 112
    candidates for 256 coefficients)."""
    from mpcith_kyber_kosk_amd import api
    k = 3
    tapes = [oracle.tape_bytes_for(k, 700 + b) for b in range(2)]
    good = api.Kosk(kyber_k=k, max_batch=2)
    pks, sks, pis = good.verifiable_keygen(tapes)
    monkeypatch.setenv("KOSK_DEBUG_XOF_BLOCKS", "1")
    ctx = api.Kosk(kyber_k=k, max_batch=2)
    monkeypatch.delenv("KOSK_DEBUG_XOF_BLOCKS")
    with pytest.raises(api.KoskError, match="block limit"):
        ctx.verifiable_keygen(tapes)
    with pytest.raises(api.KoskError, match="block limit"):
        ctx.verifiable_keygen_resident(tapes)
    with pytest.raises(api.KoskError, match="block limit"):
        ctx.verify(pis, pks)
    # the knob is per handle and the error does not stick to the process: the other handle is unaffected
    assert good.verifiable_keygen(tapes) == (pks, sks, pis)
    assert good.verify(pis, pks) == [True, True]
    ctx.close(); good.close()
