#!/usr/bin/env python3
"""GPU probe: forward throughput of the policy/value ResNet by dtype / batch (positions per second)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chinesechess-alphazero_amd"))
from cchess_alphazero.agent.model import CChessNet, InferenceNet, flops_per_position  # noqa: E402


def main():
    torch.manual_seed(0)
    res = []
    for (blocks, filt) in ((7, 128),):
        net = CChessNet(cnn_filter_num=filt, res_layer_num=blocks)
        fl = flops_per_position(net.cfg)
        for dt in (torch.float32, torch.bfloat16, torch.float16):
            inf = InferenceNet(net, dt).cuda()
            for B in (4096, 16384, 32768):
                x = (torch.rand(B, 14, 10, 9, device="cuda") < 0.03).to(dt)
                for _ in range(3):
                    inf(x)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                it = 5
                for _ in range(it):
                    p, v = inf(x)
                torch.cuda.synchronize()
                dtm = (time.perf_counter() - t0) / it
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    p, v = inf(x)
                g.replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(it):
                    g.replay()
                torch.cuda.synchronize()
                dtg = (time.perf_counter() - t0) / it
                r = dict(net=f"{blocks}x{filt}", dtype=str(dt), batch=B, ms=dtm * 1e3, ms_graph=dtg * 1e3,
                         pos_per_s=B / dtg, tflops=fl * B / dtg / 1e12)
                print(json.dumps(r), flush=True)
                res.append(r)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "nn_probe.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
