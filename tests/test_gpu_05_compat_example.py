"""The reference-API shim (include/kosk_compat.hpp) driven by a main.cpp-style C++ caller: its printed digests
must equal the digests recorded from the compiled reference (tests/golden)."""
import hashlib
import json
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "kosk_tape_v1.json")))["reference"]


@pytest.mark.parametrize("k", [2, 3, 4])
def test_main_like_example_reproduces_reference_digests(k, oracle):
    exe = os.path.join(ROOT, "examples", "main_like_k%d" % k)
    if not os.path.exists(exe):
        pytest.fail("examples/main_like_k%d missing: run __graft_entry__.build()" % k)
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    out = r.stdout
    assert r.returncode == 0, out
    assert "[result] kyber kosk verify success" in out
    assert "[tampered] verify = 0" in out
    got = dict(re.findall(r"^(pk|sk|pi) sha3_256 = ([0-9a-f]{64})$", out, flags=re.M))
    ref = GOLD[str(k)]
    assert got == {"pk": ref["sha3_pk"], "sk": ref["sha3_sk"], "pi": ref["sha3_pi"]}
    assert "[proof size] %d kilobytes" % (ref["proof_bytes"] // 1024) in out
    # the "mlwe prover test" half (main.cpp:18-62) on the second-level API, against the oracle in the same call order
    assert "[result] mlwe verify success" in out
    tape = hashlib.shake_256(b"kosk-tape-v1:main-order").digest(oracle.params(k).tape_bytes)
    ref2 = oracle.main_order(k, tape)
    got2 = dict(re.findall(r"^(pre|pi\(main-order\)) sha3_256 = ([0-9a-f]{64})$", out, flags=re.M))
    assert got2["pre"] == hashlib.sha3_256(ref2["rand"] + ref2["range"]).hexdigest()
    assert got2["pi(main-order)"] == hashlib.sha3_256(ref2["pi"]).hexdigest()
    assert "[tape] consumed %d of" % ref2["used"][-1] in out
