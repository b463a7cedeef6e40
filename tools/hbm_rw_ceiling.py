#!/usr/bin/env python3
"""GPU: what library kernels reach on this device: fill (pure write), copy, sum (pure read) of 4 GiB -- the ceilings the
HBM-bound kernels are compared with (round 6: fill 6.9 TB/s, copy 4.7 TB/s, sum 4.0 TB/s)."""
import torch, time
x = torch.empty(1 << 30, dtype=torch.float32, device="cuda")   # 4 GiB
y = torch.empty_like(x)
for name, fn, nbytes in (("fill (write 4 GiB)", lambda: x.fill_(1.0), x.numel() * 4),
                         ("copy (read 4 + write 4 GiB)", lambda: y.copy_(x), 2 * x.numel() * 4),
                         ("sum (read 4 GiB)", lambda: x.sum(), x.numel() * 4)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name}: {ms:.3f} ms = {nbytes / ms / 1e9:.2f} TB/s")
