#!/usr/bin/env python3
"""Calibration of plain streaming kernels on this GPU (torch ops), for comparison with the
box-pipeline kernels: fill and copy of buffers of the pipeline's sizes."""
import torch
dev = "cuda"
def timeit(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
a = torch.empty(32 * 102000 * 5, device=dev)
b = torch.empty(32 * 102000 * 6, device=dev)
c = torch.empty_like(b)
big = torch.empty(256 * 1024 * 1024, device=dev)
big2 = torch.empty_like(big)
for name, fn, nbytes in [
    ("fill 65 MB", lambda: a.fill_(1.0), a.numel() * 4),
    ("fill 78 MB", lambda: b.fill_(1.0), b.numel() * 4),
    ("copy 78 MB", lambda: c.copy_(b), b.numel() * 8),
    ("fill 1 GiB", lambda: big.fill_(1.0), big.numel() * 4),
    ("copy 1 GiB", lambda: big2.copy_(big), big.numel() * 8),
    ("exp 65 MB in place", lambda: a.exp_(), a.numel() * 8),
]:
    us = timeit(fn)
    print("%-22s %9.2f us  %8.1f GB/s" % (name, us, nbytes / us / 1e3))
