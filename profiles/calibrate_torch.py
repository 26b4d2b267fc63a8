"""What PyTorch's own streaming kernels reach on the bench box, for scale next to the 8 TB/s
nominal peak: full reductions (pure reads) and fills (pure writes) of bench-sized buffers."""
import torch, time
x = torch.randn(78_336_000 // 4, device="cuda")
big = torch.randn(1 << 28, device="cuda")   # 1 GiB
for name, t, f in (("sum 78MB", x, lambda t: t.sum()), ("sum 1GiB", big, lambda t: t.sum()),
                   ("max 78MB", x, lambda t: t.max()), ("fill 65MB", torch.empty(65_288_200 // 4, device="cuda"), lambda t: t.fill_(1.0)),
                   ("fill 1GiB", big, lambda t: t.fill_(1.0))):
    for _ in range(5): f(t)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f(t)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print(name, round(ms * 1e3, 2), "us", round(t.numel() * 4 / ms / 1e6, 1), "GB/s")
