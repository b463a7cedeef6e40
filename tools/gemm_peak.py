#!/usr/bin/env python3
"""GPU: what the vendor GEMM library (hipBLASLt through torch.matmul) sustains on the matrix pipe of THIS box, bf16 with
fp32 accumulation, on large square-ish problems run long enough for the clock to settle -- the practical ceiling the
residual-block kernel's issued bf16 rate (bench.py: roofline.issued_bf16_tflops) can be read against."""
import json
import time
import torch

out = {"device": torch.cuda.get_device_name(0), "runs": []}
for (m, n, k) in ((8192, 8192, 8192), (16384, 8192, 8192), (32768, 4096, 4096), (2949120 // 16, 128, 1152)):
    a = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(k, n, device="cuda", dtype=torch.bfloat16)
    c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    for _ in range(10):
        torch.matmul(a, b, out=c)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    iters = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while True:
        for _ in range(20):
            torch.matmul(a, b, out=c)
        iters += 20
        torch.cuda.synchronize()
        if time.perf_counter() - t0 > 3.0:
            break
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    out["runs"].append({"m": m, "n": n, "k": k, "ms": ms, "tflops": 2.0 * m * n * k / (ms * 1e-3) / 1e12, "iters": iters})
    del a, b, c
print(json.dumps(out))
