#!/usr/bin/env python3
"""Regenerates tests/golden/kosk_tape_v1.json.

Two kinds of entries:
  * "reference": digests recorded from the COMPILED REFERENCE (survey session, SURVEY.md 8(c) /
    BASELINE.md section 2: reference sources + deterministic randombytes tape).  These are the pin;
    this script only copies them in -- it does not (and cannot) recompute them.
  * "oracle": per-field digests, tape accounting and stage digests produced by oracle/libkosk_oracle.so
    AFTER it has been checked against the "reference" entries; they localise a regression to a field.
Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import oracle_lib as o  # noqa: E402

REFERENCE = {
    "tape_seed_format": "kosk-tape-v1:<index>  (tape = SHAKE256(seed) byte stream feeding randombytes)",
    "tape0_first16": "5cef891e3373f83be19b8dd61db0c8bd",
    "2": {"tape_calls": 284, "tape_bytes": 65280, "proof_bytes": 664340,
          "sha3_pk": "5303acc35b8f721f343bdfe43cafec16c69ca1ac5c5e79bf0379fc9d902508f2",
          "sha3_sk": "ae6d5d9158c3f86f990c6c7d397e9e91e00b73ee4960f2d4a402aaa97e4e6404",
          "sha3_pi": "e8252bad44ae1e49bdb9e5f5d75bbabb98e2453a32909013c2b2aed8d425d330",
          "I_first8": [1016, 840, 168, 386, 1056, 704, 1217, 1408],
          "f_shares_0_first4": [518, 1840, 2941, 965], "tcomm0_first8": "3f5c46724db0714c", "comm0_first8": "0d73aa7fcc8eb99b"},
    "3": {"tape_calls": 295, "tape_bytes": 68062, "proof_bytes": 680980,
          "sha3_pk": "ee60e7915d57c403a871e47fbd627a1cbb347e8ccb4553e37be7ddce376faa2b",
          "sha3_sk": "63e7b9e6249fcce87dca983dc6b02d6269e74bbbd18d36894ba715ad9dcd91a0",
          "sha3_pi": "3c8192372ced98f0eb195db9dd2e68afcddddd9762a565baa6fea85504decec1",
          "I_first8": [425, 1311, 778, 1368, 1315, 431, 1418, 443]},
    "4": {"tape_calls": 322, "tape_bytes": 75676, "proof_bytes": 744148,
          "sha3_pk": "52a4e32493e9f1ebb3af14684191ef66d49122751543c682adf1e73271b6f96a",
          "sha3_sk": "3ea4d2f1c4358189a75a035a21ff43cb70f4487ba66d02abb29e6c861ad38ed6",
          "sha3_pi": "ad01a6be2940dcc464e5083c0ca8226777ec6ce12303509b8993763812a93e39",
          "I_first8": [1433, 345, 291, 1127, 1404, 1414, 763, 1144]},
    "lagrange_tables_sha256_prefix_suffix": {"share_ddeg": ["7d083b33", "4217cc61"], "recon_ddeg": ["84e06c25", "712c44dc"],
                                             "recon_2ddeg": ["f9237bd6", "40282717"]},
    "share_ddeg_row0_first4": [1, 2922, 2725, 1644],
}


def main():
    out = {"reference": REFERENCE, "oracle": {}}
    for k in (2, 3, 4):
        p = o.params(k)
        for idx in (0, 1):
            tape = o.tape_bytes_for(k, idx)
            pk, sk, pi, calls, pos, tr = o.verifiable_keygen(k, tape, trace=True)
            if idx == 0:
                ref = REFERENCE[str(k)]
                assert hashlib.sha3_256(pi).hexdigest() == ref["sha3_pi"], "oracle no longer matches the reference pin"
            out["oracle"]["k%d_tape%d" % (k, idx)] = {
                "sha3_pk": hashlib.sha3_256(pk).hexdigest(), "sha3_sk": hashlib.sha3_256(sk).hexdigest(),
                "sha3_pi": hashlib.sha3_256(pi).hexdigest(), "tape_calls": calls, "tape_pos": pos,
                "fields_sha3": [hashlib.sha3_256(pi[p.off[i]:p.off[i] + p.size[i]]).hexdigest()[:16] for i in range(24)],
                "field_offsets": [p.off[i] for i in range(24)],
                "h1": bytes(tr.h1).hex(), "ch": bytes(tr.ch).hex(), "alpha_first4": list(tr.alpha)[:4],
            }
    with open(os.path.join(ROOT, "tests", "golden", "kosk_tape_v1.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote tests/golden/kosk_tape_v1.json")


if __name__ == "__main__":
    main()
