#!/usr/bin/env python3
"""GPU probe: the split arithmetic on (hi, lo) pairs of fp16 instead of bf16 -- same kernels (E = _Float16, parts = 2), same
three MFMAs per product, 22 bits per operand instead of 16 where fp16's range holds the lo parts.  The lo part of a
typical filter tap (|w| ~ 0.03 -> |lo| ~ 1e-5) is an fp16 SUBNORMAL: the probe answers whether v_mfma_f32_32x32x16_f16
honours subnormal inputs (then the error class is ~2^-21 per product) or flushes them (then it is fp16-class).

    python tools/f16x3_probe.py          (on the MI355X)
"""
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chinesechess-alphazero_amd"))
from cchess_alphazero import _native  # noqa: E402


def split(t, dtype):
    hi = t.to(dtype)
    return hi, (t - hi.float()).to(dtype)


def main():
    c, n = 128, 64
    d = torch.float64
    out = {}
    for name, xs, ws in (("typical", 1.5, 1.0), ("tiny_w", 1.5, 2.0 ** -8), ("tiny_x", 2.0 ** -10, 1.0)):
        g = torch.Generator(device="cuda").manual_seed(7)
        x = (torch.randn((n, 90, c), device="cuda", generator=g) * xs).relu()
        w = torch.randn((c, c, 3, 3), device="cuda", generator=g) / (3.0 * c ** 0.5) * ws
        bias = torch.zeros((c,), device="cuda")
        img = lambda t: t.to(d).view(n, 10, 9, c).permute(0, 3, 1, 2)
        conv = lambda a, ww: F.conv2d(a, ww, None, padding=1).permute(0, 2, 3, 1).reshape(n, 90, c)
        exact = conv(img(x), w.to(d))
        mag = conv(img(x).abs(), w.to(d).abs()) + 1e-300
        res = {}
        for dt in (torch.bfloat16, torch.float16):
            wp = _native.pack_conv3x3_weights(w, dt, 2).cuda()
            y = torch.empty((n, 90, c), device="cuda")
            _native.conv3x3(split(x, dt), wp, bias, out_f32=y, relu=False)
            res[str(dt)] = float(((y.to(d) - exact).abs() / mag).max())
            # the error of the operand model alone (float64 products of the rounded pairs, lo*lo dropped)
            xh, xl = (t.to(d) for t in split(x, dt))
            wh = w.to(dt).to(d)
            wl = (w - w.to(dt).float()).to(dt).to(d)
            model = conv(img(xh), wh) + conv(img(xh), wl) + conv(img(xl), wh)
            res[str(dt) + "_operand_model"] = float(((model - exact).abs() / mag).max())
            res[str(dt) + "_kernel_vs_model"] = float(((y.to(d) - model).abs() / mag).max())
        res["lo_w_subnormal_fraction"] = float(((w - w.half().float()).abs() < 2.0 ** -14).float().mean())
        out[name] = res
        print(name, res, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
