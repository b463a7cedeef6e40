#!/usr/bin/env python3
"""GPU (round 6): the four-wave pair kernel (k_resblock_ip4_c8, built for 192 filters) on the 128-filter tower's images against
k_tower on the same chain of blocks (cz_tower under CZ_TOWER4=1 / 0): equality of the exit image and time per block.

    python tools/time_ip4_128.py [c6|c8] [boards] [blocks]"""
import sys
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chinesechess-alphazero_amd"))


def main():
    from cchess_alphazero import _native
    from cchess_alphazero.agent.model import CChessNet, calibration_planes, guarded_inference_net
    arith = sys.argv[1] if len(sys.argv) > 1 else "c6"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
    nb = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    torch.manual_seed(0)
    net = CChessNet(cnn_filter_num=128, res_layer_num=7).eval()
    base = calibration_planes(4096, 14, seed=1)
    planes = base.repeat((n + 4095) // 4096, 1, 1, 1)[:n].contiguous()
    g = guarded_inference_net(net, torch.float32, trunk="mfma", arith=arith, guard=False)
    tag = torch.int8 if arith == "c6" else torch.uint8
    fmt = _native.IMG_C6 if arith == "c6" else _native.IMG_C8
    # the entry image: input layer + block 0
    x = (torch.empty((n, 90, 128), dtype=torch.float16, device="cuda"), torch.empty((n, 90, 256), dtype=tag, device="cuda"))
    w1, b1, w2, b2 = g._block_params(0)
    _native.input_resblock(planes, g.in_table32, g.in_bias32, w1, b1, w2, b2, out=x)
    blocks = [g._block_params(i) for i in range(1, 1 + nb)]
    bl_t = _native.BlockList(blocks, [fmt] * nb, [fmt] * nb)
    ya = tuple(torch.empty_like(t) for t in x)
    yb = tuple(torch.zeros_like(t) for t in x)

    def run_tower():                       # cz_tower on k_tower (round 6's first chain kernel)
        os.environ["CZ_TOWER4"] = "0"
        _native.tower(x, bl_t, fmt, out=ya)

    def run_ip4():                         # cz_tower on the four-wave pair kernel (the default since the end of round 6)
        os.environ["CZ_TOWER4"] = "1"
        _native.tower(x, bl_t, fmt, out=yb)
    run_tower()
    run_ip4()
    torch.cuda.synchronize()
    same = torch.equal(ya[0], yb[0]) and torch.equal(ya[1].view(torch.uint8), yb[1].view(torch.uint8))
    diff = (ya[0].float() - yb[0].float()).abs().max().item()
    print(f"{arith}: {nb} blocks, {n} boards: exit images equal: {same} (max |hi diff| {diff:.3e})", flush=True)
    for rep in range(3):
        for name, fn in (("k_tower", run_tower), ("ip4_128", run_ip4)):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            print(f"  {name}: {e0.elapsed_time(e1) / 10 / nb:.3f} ms per block", flush=True)


if __name__ == "__main__":
    main()
