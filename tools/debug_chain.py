#!/usr/bin/env python3
"""GPU debug: cz_tower_c6 against block-by-block cz_resblock launches on the operand pairs of a real c6 tower."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "chinesechess-alphazero_amd"), ROOT, os.path.join(ROOT, "tests")]
import torch
from cchess_alphazero import _native
from cchess_alphazero.agent.model import calibration_planes, guarded_inference_net
from test_gpu_guard import peaked_net

blocks_n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
net = peaked_net(20.0, blocks=blocks_n)
planes_all = calibration_planes(600, 14, seed=23)
g = guarded_inference_net(net, torch.float32, trunk="mfma", arith="c6", guard=False, planes=planes_all[:256])
real = _native.tower_c6


def dbg(x, blocks, out, count=None):
    cur = (x[0].clone(), x[1].clone())
    for (w1, b1, w2, b2) in blocks:
        o = (torch.zeros_like(cur[0]), torch.zeros_like(cur[1]))
        _native.resblock(cur, w1, b1, w2, b2, out=o, count=count)
        cur = o
    out[0].zero_(); out[1].zero_()
    real(x, blocks, out, count)
    torch.cuda.synchronize()
    n = x[0].shape[0]
    hi_ok = (out[0] == cur[0]).view(n, 90, 128)
    lo_ok = (out[1].view(torch.uint8) == cur[1].view(torch.uint8)).view(n, 90, 256)
    print(f"n={n} blocks={len(blocks)}: f16 equal {hi_ok.float().mean().item():.4f}, c6 image equal {lo_ok.float().mean().item():.4f}; "
          f"chain f16 nonzero {float((out[0] != 0).float().mean()):.3f} (seq {float((cur[0] != 0).float().mean()):.3f}); finite {bool(torch.isfinite(out[0].float()).all())}")
    if not hi_ok.all():
        bad_board = (~hi_ok).view(n, -1).any(1).nonzero().flatten()[:8].tolist()
        print("  boards with f16 differences:", bad_board, "of", n)
        b = bad_board[0]
        rows = (~hi_ok[b]).any(1).nonzero().flatten().tolist()
        print("  board", b, "rows differing:", rows[:30], "count", len(rows))
        r = rows[0]
        cols = (~hi_ok[b, r]).nonzero().flatten().tolist()
        print("  row", r, "channels differing:", cols[:40], "count", len(cols))
        print("  chain:", out[0][b, r, :8].tolist(), "\n  seq:  ", cur[0][b, r, :8].tolist())
    return out


_native.tower_c6 = dbg
import cchess_alphazero.agent.model as M
g.chain_blocks = True
for n in (2, 1, 3, 300):
    g(planes_all[:n].contiguous())
