#!/bin/bash
# Host code of the product library under AddressSanitizer, ON the GPU box: the five host C++ files (C ABI, lanes, contexts,
# verifier, split API, host hashing / pool) are compiled with g++ -fsanitize=address and linked with the normal gfx950 kernel
# objects; the multi-handle / lane-thread / page-locking cases of tests/gpu_child_cases.py then run against that library in a
# python started with libasan preloaded.  Device code is NOT instrumented (GPU ASan / xnack+ objects are not available on this
# pool); this catches heap overflows, use-after-free and double frees in the host paths that round 2's SIGABRT went through.
#   tools/host_asan.sh build          (here or on the box; needs mpcith_kyber_kosk_amd/_build/*.hip.o from the normal build)
#   tools/host_asan.sh run [logfile]  (on the box)
set -e
cd "$(dirname "$0")/.."
PKG=mpcith_kyber_kosk_amd
if [ "$1" = build ]; then
    python -m $PKG.build > /dev/null
    mkdir -p $PKG/_build_asan
    for f in kosk_capi kosk_ctx kosk_verify kosk_split kosk_host; do
        g++ -std=c++20 -O1 -g -fPIC -fsanitize=address -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -w \
            -c $PKG/csrc/$f.cpp -o $PKG/_build_asan/$f.o
    done
    g++ -shared -fPIC -fsanitize=address -o $PKG/libkosk_mi355x_asan.so $PKG/_build/kosk_kernels.hip.o $PKG/_build/kosk_verify_kernels.hip.o \
        $PKG/_build/kosk_keygen_kernels.hip.o $PKG/_build/kosk_compact.hip.o $PKG/_build_asan/*.o -L/opt/rocm/lib -lamdhip64 -lpthread -Wl,-rpath,/opt/rocm/lib
    echo built $PKG/libkosk_mi355x_asan.so
    exit 0
fi
LOG=${2:-/dev/stdout}
export KOSK_LIB_PATH=$PWD/$PKG/libkosk_mi355x_asan.so
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1
LD_PRELOAD=$(gcc -print-file-name=libasan.so) python3 -c "
from tests.gpu_child_cases import *
streamed_chunks(2); streamed_chunks(3); pinned_buffers(3); errors_do_not_kill(2); streamed_loop(3, 40); big_batches()
print('host_asan: ok')
" > "$LOG" 2>&1
