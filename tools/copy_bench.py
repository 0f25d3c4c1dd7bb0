import torch
for mb in (32, 128, 512):
    n = mb * 1024 * 1024 // 2
    a = torch.randint(0, 3329, (n,), dtype=torch.int16, device="cuda"); b = torch.zeros_like(a)
    for _ in range(3): b.copy_(a)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("torch copy %d MB -> %d MB: %.1f us, %.0f GB/s (read+write)" % (mb, mb, ms * 1e3, 2 * mb * 1.048576 / ms))
