#!/usr/bin/env python3
"""GPU probe 2: ways to make the 7x128 ResNet forward faster under PyTorch-ROCm without changing precision:
MIOpen find mode (cudnn.benchmark), NCHW vs NHWC, fused conv+bias+relu ops, batch size."""
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chinesechess-alphazero_amd"))
from cchess_alphazero.agent.model import CChessNet, InferenceNet, flops_per_position  # noqa: E402


def timeit(fn, it=5, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it


def main():
    torch.manual_seed(0)
    net = CChessNet(cnn_filter_num=128, res_layer_num=7)
    fl = flops_per_position(net.cfg)
    B = 32768
    res = []

    def report(name, dt, b=B):
        r = dict(name=name, ms=dt * 1e3, pos_per_s=b / dt, tflops=fl * b / dt / 1e12)
        print(json.dumps(r), flush=True)
        res.append(r)

    for dt_ in (torch.float32, torch.bfloat16):
        x = (torch.rand(B, 14, 10, 9, device="cuda") < 0.03).to(dt_)
        for bench in (False, True):
            torch.backends.cudnn.benchmark = bench
            inf = InferenceNet(net, dt_).cuda()
            report(f"{dt_} nhwc benchmark={bench}", timeit(lambda: inf(x)))
            # NCHW variant
            inf2 = InferenceNet(net, dt_).cuda().to(memory_format=torch.contiguous_format)

            def fwd_nchw():
                h = F.relu(inf2.input_conv(x))
                for c1, c2 in inf2.res:
                    y = F.relu(c1(h))
                    h = F.relu(h + c2(y))
                return h
            try:
                report(f"{dt_} nchw trunk-only benchmark={bench}", timeit(fwd_nchw))
            except Exception as e:
                print("nchw failed", e)
        # single conv layer timings: plain vs fused miopen op
        torch.backends.cudnn.benchmark = True
        w = torch.randn(128, 128, 3, 3, device="cuda", dtype=dt_) * 0.05
        b = torch.randn(128, device="cuda", dtype=dt_)
        for fmt, nm in ((torch.channels_last, "nhwc"), (torch.contiguous_format, "nchw")):
            h = torch.randn(B, 128, 10, 9, device="cuda", dtype=dt_).contiguous(memory_format=fmt)
            wf = w.contiguous(memory_format=fmt)
            report(f"{dt_} {nm} conv only", timeit(lambda: F.conv2d(h, wf, None, padding=1)))
            report(f"{dt_} {nm} conv+bias+relu (3 ops)", timeit(lambda: F.relu(F.conv2d(h, wf, b, padding=1))))
            if hasattr(torch, "miopen_convolution_relu"):
                try:
                    report(f"{dt_} {nm} miopen_convolution_relu",
                           timeit(lambda: torch.miopen_convolution_relu(h, wf, b, [1, 1], [1, 1], [1, 1], 1)))
                    z = torch.randn_like(h)
                    report(f"{dt_} {nm} miopen_convolution_add_relu",
                           timeit(lambda: torch.miopen_convolution_add_relu(h, wf, z, 1.0, b, [1, 1], [1, 1], [1, 1], 1)))
                except Exception as e:
                    print("fused op failed:", str(e)[:200])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "nn_probe2.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
