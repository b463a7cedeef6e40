#!/usr/bin/env python3
"""Per-launch times of the 7 x 128 tower (benchmark weights, calibration positions) for a list of arithmetics: HIP events
around every residual-block launch of InferenceNet (block_events), 32768 positions, mean of the timed repetitions.

    python tools/time_tower_launches.py [c8,c6] [32768] [masks]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chinesechess-alphazero_amd"))


def main():
    from cchess_alphazero.agent.model import CChessNet, calibration_planes, guarded_inference_net
    ariths = sys.argv[1].split(",") if len(sys.argv) > 1 else ["c8", "c6"]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
    use_masks = len(sys.argv) > 3 and sys.argv[3] == "masks"     # hand the occupancy boards in, as the engine does (round 5)
    torch.manual_seed(0)
    net = CChessNet(cnn_filter_num=128, res_layer_num=7).eval()
    base = calibration_planes(4096, 14, seed=1)
    planes = base.repeat((n + 4095) // 4096, 1, 1, 1)[:n].contiguous()
    masks = None
    if use_masks:
        bits = (planes.reshape(n, 14, 90) != 0).to(torch.int64)
        w = (1 << torch.arange(14, device=planes.device, dtype=torch.int64)).view(1, 14, 1)
        masks = torch.zeros((n, 96), dtype=torch.int64, device=planes.device)
        masks[:, :90] = (bits * w).sum(1)
        masks = masks.to(torch.int32).contiguous()
    out = {}
    for arith in ariths:
        g = guarded_inference_net(net, torch.float32, trunk="mfma", arith=arith, guard=False)
        for _ in range(6):
            g(planes, masks=masks)
        g.block_events = []
        reps = 8
        for _ in range(reps):
            g(planes, masks=masks)
        torch.cuda.synchronize()
        from cchess_alphazero.agent.model import events_ms
        ms = events_ms(g.block_events)
        g.block_events = None
        per = [sum(ms[i::7]) / reps for i in range(7)]
        out[arith] = {"per_launch_ms": per, "tower_ms": sum(per)}
        print(arith, " ".join(f"{x:.3f}" for x in per), f"sum {sum(per):.3f}", flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
