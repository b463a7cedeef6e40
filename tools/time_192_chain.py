#!/usr/bin/env python3
"""Per-block times of the 10 x 192 tower (the reference's deployed topology) per arithmetic: one launch per block, chained
(cz_resblock_chain), and -- a timing experiment with wrong results -- chained with every block reading ONE block's filters (no L2
misses at the block switches).  32768 boards, HIP events around the tower launches.

    python tools/time_192_chain.py [c8,c6] [32768]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chinesechess-alphazero_amd"))


def main():
    from cchess_alphazero import _native
    from cchess_alphazero.agent.model import CChessNet, calibration_planes, events_ms, guarded_inference_net
    ariths = sys.argv[1].split(",") if len(sys.argv) > 1 else ["c8", "c6"]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
    torch.manual_seed(0)
    net = CChessNet(cnn_filter_num=192, res_layer_num=10).eval()
    base = calibration_planes(4096, 14, seed=1)
    planes = base.repeat((n + 4095) // 4096, 1, 1, 1)[:n].contiguous()
    real = _native.resblock_chain
    out = {}
    for arith in ariths:
        g = guarded_inference_net(net, torch.float32, trunk="mfma", arith=arith, guard=False)
        for leg in ("blocks", "chain", "same", "blocks", "chain", "same"):
            g.chain_blocks = leg != "blocks"
            _native.resblock_chain = (lambda x, bl, **k: real(x, _native.BlockList([bl.blocks[0]] * bl.n), **k)) if leg == "same" else real
            for _ in range(3):
                g(planes)
            g.block_events = []
            reps = 4
            for _ in range(reps):
                g(planes)
            torch.cuda.synchronize()
            ms = events_ms(g.block_events)
            g.block_events = None
            per = [sum(ms[i::10]) / reps for i in range(10)]
            out.setdefault(arith, {}).setdefault(leg, []).append(sum(per))
            print(arith, leg, " ".join(f"{x:.2f}" for x in per), f"sum {sum(per):.2f}", flush=True)
        _native.resblock_chain = real
    print(json.dumps(out))


if __name__ == "__main__":
    main()
