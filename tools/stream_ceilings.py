import torch, time
x = torch.empty(1 << 28, dtype=torch.float32, device="cuda")   # 1 GiB
y = torch.empty_like(x)
for name, fn, nbytes in (("fill (write only)", lambda: x.fill_(1.5), x.numel() * 4), ("copy (read + write)", lambda: y.copy_(x), x.numel() * 8),
                         ("sum (read only)", lambda: x.sum(), x.numel() * 4)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    print(f"{name}: {ms:.4f} ms for {nbytes / 2**30:.0f} GiB moved -> {nbytes / ms / 1e6:.0f} GB/s")
