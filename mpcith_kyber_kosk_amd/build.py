"""Build the gfx950 shared library (HIP kernels + C-ABI) in-tree with hipcc.

    python -m mpcith_kyber_kosk_amd.build [--force]

.hip sources and the HIP-runtime host code are compiled as HIP for gfx950; the pure host
file (kosk_host.cpp: Fiat-Shamir hashing, keygen, tables, thread pool) as plain C++.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libkosk_mi355x.so")
HIP_SOURCES = ["kosk_kernels.hip", "kosk_verify_kernels.hip", "kosk_keygen_kernels.hip", "kosk_compact.hip", "kosk_fs_kernels.hip", "kosk_ctx.cpp", "kosk_verify.cpp", "kosk_split.cpp", "kosk_capi.cpp"]
CXX_SOURCES = ["kosk_host.cpp"]
COMMON = ["-O3", "-std=c++20", "-fPIC"]


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".inc"))] + [os.path.join(HERE, "..", "include", "kosk_mi355x.h")]


def _compile(src, hip, force):
    obj = os.path.join(OBJ, src + ".o")
    path = os.path.join(CSRC, src)
    deps = [path] + _headers()
    if not force and os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in deps):
        return obj
    cmd = ["hipcc"] + COMMON + (["--offload-arch=gfx950", "-x", "hip"] if hip else ["-x", "c++"]) + ["-c", path, "-o", obj]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout)
        raise RuntimeError("hipcc failed on " + src)
    return obj


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    jobs = [(s, True) for s in HIP_SOURCES] + [(s, False) for s in CXX_SOURCES]
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(lambda j: _compile(j[0], j[1], force), jobs))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lpthread"]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout)
            raise RuntimeError("link failed for libkosk_mi355x.so")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
