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

# in place (the update kernel's pattern: every line read, then written) against out of place, at the firework's 402 MB and at 1 GiB
for n in (100663296, 1 << 28):
    a = torch.empty(n, dtype=torch.float32, device="cuda").fill_(1.0)
    b = torch.empty_like(a)
    for name, fn in (("in place  a *= k", lambda: a.mul_(1.0001)), ("out of place b = a * k", lambda: torch.mul(a, 1.0001, out=b))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): fn()
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 20
        print(f"{n * 4 / 1e6:.0f} MB, {name}: {ms:.4f} ms -> {n * 8 / ms / 1e6:.0f} GB/s (read + write)")
