#!/usr/bin/env python3
"""GPU: times cz_input_conv (5x5 input layer on the u8 planes, split operands) on the benchmark batch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "chinesechess-alphazero_amd"), ROOT]
import torch  # noqa: E402
from cchess_alphazero import _native  # noqa: E402

n, c = 32768, 128
torch.manual_seed(0)
planes = (torch.rand((n, 14, 10, 9), device="cuda") < 0.07).to(torch.uint8)
w = _native.pack_input_conv_weights(torch.randn(c, 14, 5, 5) * 0.1, torch.bfloat16, 2).cuda()
b = torch.randn(c).cuda()
out = tuple(torch.empty((n, 90, c), dtype=torch.bfloat16, device="cuda") for _ in range(2))
for _ in range(3):
    _native.input_conv(planes, w, b, out)
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    _native.input_conv(planes, w, b, out)
e.record()
torch.cuda.synchronize()
print("cz_input_conv", n, "boards:", round(a.elapsed_time(e) / 20, 4), "ms; checksum", float(out[0].float().sum()))
