"""Build the gfx950 shared library (HIP kernels + C-ABI) in-tree with hipcc.

    python -m mpcith_kyber_kosk_amd.build            # rebuild if sources changed
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libkosk_mi355x.so")
SOURCES = ["kosk_kernels.hip", "kosk_verify_kernels.hip", "kosk_host.cpp", "kosk_ctx.cpp", "kosk_verify.cpp", "kosk_capi.cpp"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "kosk_mi355x.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", LIB] + srcs + ["-lpthread"]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError("hipcc failed building libkosk_mi355x.so")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
