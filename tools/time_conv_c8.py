#!/usr/bin/env python3
"""GPU: the c8 convolution (cz_conv3x3_c8: fp16 + two scaled-fp8 correction terms) against the split-bf16 one
(cz_conv3x3, three bf16 MFMAs per product), same shape (128 filters, two boards per workgroup, fp32 output), back to
back for a few seconds each so that both run in the power-capped state.

    python tools/time_conv_c8.py [--boards 32768] [--seconds 3]      ->  gpurun_out/time_conv_c8.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chinesechess-alphazero_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--boards", type=int, default=32768)
    ap.add_argument("--seconds", type=float, default=3.0)
    a = ap.parse_args()
    import torch
    from cchess_alphazero import _native
    c, n = 128, a.boards
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((n, 90, c), device="cuda", generator=g).relu()
    w = torch.randn((c, c, 3, 3), device="cuda", generator=g) / (3.0 * c ** 0.5)
    bias = torch.randn((c,), device="cuda", generator=g)
    out = torch.empty((n, 90, c), device="cuda")
    xb = (x.to(torch.bfloat16), (x - x.to(torch.bfloat16).float()).to(torch.bfloat16))
    pb = _native.pack_conv3x3_weights(w, torch.bfloat16, 2).cuda()
    xh, xc = _native.split_c8(x)
    pc = _native.pack_conv3x3_c8_weights(w).cuda()
    runs = {"bf16x3": lambda: _native.conv3x3(xb, pb, bias, out_f32=out, relu=True),
            "f16+2fp8": lambda: _native.conv3x3_c8((xh, xc), pc, bias, out_f32=out, relu=True)}
    res = {}
    for name in ("bf16x3", "f16+2fp8", "bf16x3", "f16+2fp8"):
        f = runs[name]
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0, k, ms = time.time(), 0, []
        while time.time() - t0 < a.seconds:
            e0.record()
            for _ in range(20):
                f()
            e1.record()
            e1.synchronize()
            ms.append(e0.elapsed_time(e1) / 20)
            k += 20
        res.setdefault(name, []).append({"launches": k, "ms_first": ms[0], "ms_settled": sum(ms[-5:]) / len(ms[-5:])})
        print(name, res[name][-1], flush=True)
    b = sum(r["ms_settled"] for r in res["bf16x3"]) / len(res["bf16x3"])
    f8 = sum(r["ms_settled"] for r in res["f16+2fp8"]) / len(res["f16+2fp8"])
    out_d = {"tool": "tools/time_conv_c8.py", "boards": n, "filters": c, "results": res, "ms_bf16x3": b, "ms_f16_2fp8": f8,
             "speedup": b / f8}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out_d, open(os.path.join(ROOT, "gpurun_out", "time_conv_c8.json"), "w"), indent=1)
    print(json.dumps({k: out_d[k] for k in ("ms_bf16x3", "ms_f16_2fp8", "speedup")}))


if __name__ == "__main__":
    main()
