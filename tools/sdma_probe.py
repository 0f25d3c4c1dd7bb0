"""Which engine does the runtime use for device-to-host copies into page-locked memory?  Copies of 64 KB .. 8 MB (a) on a stream that
has just run a kernel, (b) on a stream that never runs kernels.  Run under
    rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d <dir> -- python3 tools/sdma_probe.py
blit copies show up as __amd_rocclr_copyBuffer kernels, SDMA copies only in the memory-copy trace."""
import sys, time, torch
sizes = [64 << 10, 512 << 10, 2 << 20, 6 << 20, 8 << 20]
dev = [torch.empty(s, dtype=torch.uint8, device="cuda") for s in sizes]
host = [torch.empty(s, dtype=torch.uint8).pin_memory() for s in sizes]
x = torch.zeros(1 << 20, device="cuda")
s_k, s_c = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
for rep in range(3):
    for i, s in enumerate(sizes):
        with torch.cuda.stream(s_k):
            x.add_(1.0)                       # a kernel right in front of the copy
            t0 = time.perf_counter()
            host[i].copy_(dev[i], non_blocking=True)
        s_k.synchronize()
        ta = time.perf_counter() - t0
        with torch.cuda.stream(s_c):
            t0 = time.perf_counter()
            host[i].copy_(dev[i], non_blocking=True)
        s_c.synchronize()
        tb = time.perf_counter() - t0
        if rep == 2:
            print("size %8d: behind a kernel %.1f us (%.1f GB/s), copy-only stream %.1f us (%.1f GB/s)" % (s, ta * 1e6, s / ta / 1e9, tb * 1e6, s / tb / 1e9))
