import torch, time
for lanes in (65536, 262144):
    a = torch.randint(0, 3329, (lanes, 256), dtype=torch.int16, device="cuda")
    b = torch.zeros_like(a)
    for _ in range(5): b.copy_(a)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): b.copy_(a)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    print("torch copy %d x 512 B: %.2f us, %.2f TB/s (read + write)" % (lanes, best * 1e3, lanes * 1024 / best / 1e9))
    c = a.view(torch.int32)
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): s_ = c.sum()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    print("torch sum (read only) %d x 512 B: %.2f us, %.2f TB/s" % (lanes, best * 1e3, lanes * 512 / best / 1e9))
