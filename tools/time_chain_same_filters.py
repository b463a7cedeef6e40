#!/usr/bin/env python3
"""Timing experiment (round 6; WRONG RESULTS in the second leg): how much of the chained c6 tower's time is
the filters' trips through the L2?  Leg "own": the 7 x 128 tower as shipped (FIRST + k_tower_c6<HEADS> over blocks 1 .. 6, twelve
packed filters of 0.5 MB cycling through every XCD's 4 MB L2).  Leg "same": the same launch with EVERY block of the chain
reading block 1's two filters (1 MB stays resident) -- same instruction stream, same LDS traffic, no filter refetches.  The
difference bounds what keeping the workgroups of an XCD on one block at a time could buy.

    python tools/time_chain_same_filters.py [32768] [reps]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chinesechess-alphazero_amd"))


def main():
    from cchess_alphazero import _native
    from cchess_alphazero.agent.model import CChessNet, calibration_planes, events_ms, guarded_inference_net
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    torch.manual_seed(0)
    net = CChessNet(cnn_filter_num=128, res_layer_num=7).eval()
    base = calibration_planes(4096, 14, seed=1)
    planes = base.repeat((n + 4095) // 4096, 1, 1, 1)[:n].contiguous()
    g = guarded_inference_net(net, torch.float32, trunk="mfma", arith="c6", guard=False)
    real = _native.tower_c6_heads
    out = {}
    for leg in ("own", "same", "own", "same"):
        if leg == "same":
            _native.tower_c6_heads = lambda x, blocks, *a, **k: real(x, [blocks[0]] * len(blocks), *a, **k)
        else:
            _native.tower_c6_heads = real
        for _ in range(4):
            g(planes)
        g.block_events = []
        for _ in range(reps):
            g(planes)
        torch.cuda.synchronize()
        ms = events_ms(g.block_events)
        g.block_events = None
        per = [sum(ms[i::7]) / reps for i in range(7)]
        out.setdefault(leg, []).append({"first_ms": per[0], "chain_ms": sum(per[1:]), "tower_ms": sum(per)})
        print(leg, f"first {per[0]:.3f} chain {sum(per[1:]):.3f} (per block {sum(per[1:]) / 6:.3f}) tower {sum(per):.3f}", flush=True)
    _native.tower_c6_heads = real
    print(json.dumps(out))


if __name__ == "__main__":
    main()
