"""GPU parity: HIP prover (through the C-ABI) vs the CPU oracle, bit for bit."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: torch.cuda.is_available() is False")
    return torch


@pytest.mark.parametrize("k", [2, 3, 4])
def test_verifiable_keygen_matches_oracle(k, oracle, torch_cuda):
    from mpcith_kyber_kosk_amd import api
    n = 3
    ctx = api.Kosk(kyber_k=k, max_batch=2)  # max_batch < n: exercises the chunking of the batch ABI
    tapes = [oracle.tape_bytes_for(k, b) for b in range(n)]
    pks, sks, pis = ctx.verifiable_keygen(tapes)
    for b in range(n):
        opk, osk, opi, calls, pos = oracle.verifiable_keygen(k, tapes[b])
        assert pks[b] == opk and sks[b] == osk
        if pis[b] != opi:
            p = oracle.params(k)
            bad = [i for i in range(24) if pis[b][p.off[i]:p.off[i] + p.size[i]] != opi[p.off[i]:p.off[i] + p.size[i]]]
            pytest.fail(f"proof {b} differs from the oracle in fields {bad}")
    ctx.close()
