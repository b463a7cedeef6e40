#!/usr/bin/env python3
"""Correctness + timing probe of the hand-written trunk convolution (csrc/xq_conv.hip) against MIOpen.

    python tools/conv_probe.py [--n 32768] [--channels 128] [--out gpurun_out/conv_probe.json]

(The ablation / in-kernel time-stamp variants this script once drove were tuning aids that computed wrong results; they
were removed from csrc/xq_conv.hip in round 3 -- what they measured is recorded in EXPERIMENTS.md (Part II, section 7b).)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chinesechess-alphazero_amd"))

import torch
import torch.nn.functional as F

from cchess_alphazero import _native


def split(t, dtype, parts):
    hi = t.to(dtype)
    if parts == 1:
        return (hi,)
    return (hi, (t - hi.float()).to(dtype))


def reference(x_nhwc, w, bias, skip, relu):
    """fp64 on the GPU: x [N, 90, C] -> [N, 90, C]"""
    n, _, c = x_nhwc.shape
    x = x_nhwc.double().view(n, 10, 9, c).permute(0, 3, 1, 2)
    y = F.conv2d(x, w.double(), bias.double(), padding=1)
    y = y.permute(0, 2, 3, 1).reshape(n, 90, c)
    if skip is not None:
        y = y + skip.double()
    return torch.relu(y) if relu else y


def check(c, dtype, parts, n=37, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn((n, 90, c), device="cuda", generator=g).relu()
    sk = torch.randn((n, 90, c), device="cuda", generator=g)
    w = torch.randn((c, c, 3, 3), device="cuda", generator=g) / (3.0 * c ** 0.5)
    b = torch.randn((c,), device="cuda", generator=g)
    wp = _native.pack_conv3x3_weights(w, dtype, parts).cuda()
    xs, ss = split(x, dtype, parts), split(sk, dtype, parts)
    res = {}
    for name, skip, relu, f32out in (("plain", None, True, False), ("skip", ss, True, False),
                                     ("f32out", ss, False, True)):
        out = tuple(torch.full((n, 90, c), 7.0, device="cuda", dtype=dtype) for _ in range(parts))
        of = torch.full((n, 90, c), 7.0, device="cuda") if f32out else None
        _native.conv3x3(xs, wp, b, skip=skip, out=None if f32out else out, out_f32=of, relu=relu)
        torch.cuda.synchronize()
        got = of.double() if f32out else sum(o.double() for o in out)
        # the reference sees exactly the operands the kernel sees (hi + lo), so only the arithmetic differs
        x_eff = sum(t.double() for t in xs)
        s_eff = sum(t.double() for t in skip) if skip is not None else None
        w_eff = w if parts == 2 else w.to(dtype).float()
        ref = reference(x_eff, w_eff, b, s_eff, relu)
        err = (got - ref).abs().max().item()
        res[name] = dict(max_abs_err=err, ref_max=ref.abs().max().item())
    return res


def bench(c, dtype, parts, n, iters=10):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((n, 90, c), device="cuda", generator=g).relu()
    w = torch.randn((c, c, 3, 3), device="cuda", generator=g) / (3.0 * c ** 0.5)
    b = torch.randn((c,), device="cuda", generator=g)
    wp = _native.pack_conv3x3_weights(w, dtype, parts).cuda()
    xs = split(x, dtype, parts)
    out = tuple(torch.empty((n, 90, c), device="cuda", dtype=dtype) for _ in range(parts))
    for _ in range(2):
        _native.conv3x3(xs, wp, b, skip=xs, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        _native.conv3x3(xs, wp, b, skip=xs, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flop = 2.0 * n * 90 * c * c * 9
    return dict(ms=ms, tflops_algorithmic=flop / ms / 1e9, mfma_tflops=flop * (3 if parts == 2 else 1) / ms / 1e9)


def check_block(c=128, dtype=torch.bfloat16, n=300, seed=3):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn((n, 90, c), device="cuda", generator=g).relu()
    w1 = torch.randn((c, c, 3, 3), device="cuda", generator=g) / (3.0 * c ** 0.5)
    w2 = torch.randn((c, c, 3, 3), device="cuda", generator=g) / (3.0 * c ** 0.5)
    b1 = torch.randn((c,), device="cuda", generator=g)
    b2 = torch.randn((c,), device="cuda", generator=g)
    p1 = _native.pack_conv3x3_weights(w1, dtype, 2).cuda()
    p2 = _native.pack_conv3x3_weights(w2, dtype, 2).cuda()
    xs = split(x, dtype, 2)
    # two-launch path as the reference for the fused one (bit-identical arithmetic expected)
    t = tuple(torch.empty_like(xs[0]) for _ in range(2))
    o2 = tuple(torch.empty_like(xs[0]) for _ in range(2))
    _native.conv3x3(xs, p1, b1, out=t)
    _native.conv3x3(t, p2, b2, skip=xs, out=o2)
    o1 = tuple(torch.full_like(xs[0], 7.0) for _ in range(2))
    _native.resblock(xs, p1, b1, p2, b2, out=o1)
    of = torch.full((n, 90, c), 7.0, device="cuda")
    _native.resblock(xs, p1, b1, p2, b2, out_f32=of)
    torch.cuda.synchronize()
    d = (sum(a.double() for a in o1) - sum(a.double() for a in o2)).abs().max().item()
    d2 = (of.double() - sum(a.double() for a in o2)).abs().max().item()
    return dict(fused_vs_two_launches=d, bit_equal=bool(torch.equal(o1[0], o2[0]) and torch.equal(o1[1], o2[1])),
                f32_vs_two_launches=d2)


def bench_block(c, dtype, parts, n, iters=10):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((n, 90, c), device="cuda", generator=g).relu()
    w = torch.randn((c, c, 3, 3), device="cuda", generator=g) / (3.0 * c ** 0.5)
    b = torch.randn((c,), device="cuda", generator=g)
    wp = _native.pack_conv3x3_weights(w, dtype, parts).cuda()
    xs = split(x, dtype, parts)
    out = tuple(torch.empty((n, 90, c), device="cuda", dtype=dtype) for _ in range(parts))
    for _ in range(2):
        _native.resblock(xs, wp, b, wp, b, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        _native.resblock(xs, wp, b, wp, b, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flop = 2 * 2.0 * n * 90 * c * c * 9
    return dict(ms=ms, ms_per_conv=ms / 2, tflops_algorithmic=flop / ms / 1e9,
                mfma_tflops=(3 if parts == 2 else 1) * flop / ms / 1e9)


def bench_miopen(c, dtype, n, iters=5):
    x = torch.randn((n, c, 10, 9), device="cuda", dtype=dtype).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((c, c, 3, 3), device="cuda") / (3.0 * c ** 0.5)).to(dtype).contiguous(
        memory_format=torch.channels_last)
    for _ in range(3):
        F.conv2d(x, w, None, padding=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        F.conv2d(x, w, None, padding=1)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return dict(ms=ms, tflops=2.0 * n * 90 * c * c * 9 / ms / 1e9)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=32768)
    ap.add_argument("--channels", type=int, default=128)
    ap.add_argument("--out", default="")
    ap.add_argument("--no-miopen", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    a = ap.parse_args()
    res = {"n": a.n, "channels": a.channels, "check": {}, "bench": {}}
    t0 = time.time()
    for c in (() if a.no_check else (a.channels, 32)):
        for dtype, parts in ((torch.bfloat16, 2), (torch.bfloat16, 1), (torch.float16, 1)):
            key = f"c{c}_{str(dtype).split('.')[-1]}_x{parts}"
            res["check"][key] = check(c, dtype, parts)
            print(key, res["check"][key], flush=True)
    for dtype, parts in ((torch.bfloat16, 2), (torch.bfloat16, 1), (torch.float16, 1)):
        key = f"{str(dtype).split('.')[-1]}_x{parts}"
        res["bench"][key] = bench(a.channels, dtype, parts, a.n)
        print(key, res["bench"][key], flush=True)
    if a.channels == 128:
        if not a.no_check:
            res["check"]["resblock"] = check_block()
            print("resblock", res["check"]["resblock"], flush=True)
        for c, dtype, parts in ((128, torch.bfloat16, 2), (128, torch.bfloat16, 1), (256, torch.float16, 1)):
            key = f"resblock_c{c}_{str(dtype).split('.')[-1]}_x{parts}"
            res["bench"][key] = bench_block(c, dtype, parts, a.n if c == 128 else a.n // 2)
            print(key, res["bench"][key], flush=True)
    if not a.no_miopen:
        for dtype in (torch.float32, torch.bfloat16):
            key = "miopen_" + str(dtype).split(".")[-1]
            res["bench"][key] = bench_miopen(a.channels, dtype, a.n)
            print(key, res["bench"][key], flush=True)
    res["seconds"] = time.time() - t0
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
