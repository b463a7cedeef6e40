#!/usr/bin/env python3
"""GPU debug (round 6): cz_tower / cz_tower_pairs against block-by-block cz_resblock launches on the operand pairs of real towers.
    python tools/debug_tower.py [arith,...] [blocks]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "chinesechess-alphazero_amd"), ROOT, os.path.join(ROOT, "tests")]
import torch
from cchess_alphazero import _native
from cchess_alphazero.agent.model import calibration_planes, guarded_inference_net
from test_gpu_guard import peaked_net

ariths = sys.argv[1].split(",") if len(sys.argv) > 1 else ["c8", "c6>5", "c8>3", "f16x3"]
blocks_n = int(sys.argv[2]) if len(sys.argv) > 2 else 7
planes_all = calibration_planes(1100, 14, seed=23)
real_tower, real_pairs = _native.tower, _native.tower_pairs
TAG = {_native.IMG_C8: torch.uint8, _native.IMG_C6: torch.int8}


def cmp(name, a, b):
    n = a.shape[0]
    eq = (a.view(torch.uint8) == b.view(torch.uint8)).view(n, -1)
    print(f"   {name}: bytes equal {eq.float().mean().item():.5f}", end="")
    if not eq.all():
        bad = (~eq).any(1).nonzero().flatten()[:6].tolist()
        print(f"  boards differing {bad} of {n}", end="")
        b0 = bad[0]
        per_row = (~eq[b0]).view(90, -1)
        rows = per_row.any(1).nonzero().flatten().tolist()
        print(f"; board {b0}: {len(rows)} rows differ, first {rows[:8]}; bytes in row {rows[0]}: {per_row[rows[0]].nonzero().flatten()[:16].tolist()}", end="")
    print()


def dbg_tower(x, bl, exit_fmt, out=None, heads=None, count=None, fmt_x=None, fmt_y=None):
    n = x[0].shape[0]
    cur = (x[0].clone(), x[1].clone())
    fx = list(bl.fmt_x) if bl.fmt_x is not None else [1] * bl.n
    fy = list(bl.fmt_y) if bl.fmt_y is not None else [1] * bl.n
    print(f" tower: n={n} blocks={bl.n} fx={fx} fy={fy} exit={exit_fmt}")
    ref_f32 = None
    for k, (w1, b1, w2, b2) in enumerate(bl.blocks):
        tag = TAG[fy[k]]                       # (a block's kernel is selected by its intermediate image's format)
        xin = (cur[0], cur[1].view(tag))
        last = k + 1 == bl.n
        if last and exit_fmt in (_native.IMG_PAIR, _native.EXIT_HEADS):
            ref_f32 = torch.zeros((n, 90, 128), dtype=torch.float32, device="cuda")
            _native.resblock(xin, w1, b1, w2, b2, out_f32=ref_f32, count=count)
        else:
            o = (torch.zeros_like(cur[0]), torch.zeros_like(cur[1]).view(tag))
            _native.resblock(xin, w1, b1, w2, b2, out=o, count=count)
            cur = o
    if exit_fmt == _native.EXIT_HEADS:
        real_tower(x, bl, exit_fmt, out=out, heads=heads, count=count)
        return
    out[0].zero_(); out[1].zero_()
    real_tower(x, bl, exit_fmt, out=out, count=count)
    torch.cuda.synchronize()
    if exit_fmt == _native.IMG_PAIR:
        hi = ref_f32.half(); lo = (ref_f32 - hi.float()).half()
        cmp("pair hi", out[0], hi); cmp("pair lo", out[1], lo)
    else:
        cmp("f16", out[0], cur[0]); cmp("image", out[1], cur[1])
    return out


def dbg_pairs(x, bl, out=None, heads=None, count=None):
    n = x[0].shape[0]
    print(f" pairs: n={n} blocks={bl.n} heads={heads is not None}")
    if heads is not None:
        return real_pairs(x, bl, out=out, heads=heads, count=count)
    cur = (x[0].clone(), x[1].clone())
    for (w1, b1, w2, b2) in bl.blocks:
        o = (torch.zeros_like(cur[0]), torch.zeros_like(cur[1]))
        _native.resblock(cur, w1, b1, w2, b2, out=o, count=count)
        cur = o
    out[0].zero_(); out[1].zero_()
    real_pairs(x, bl, out=out, count=count)
    torch.cuda.synchronize()
    cmp("hi", out[0], cur[0]); cmp("lo", out[1], cur[1])
    return out


_native.tower, _native.tower_pairs = dbg_tower, dbg_pairs
for arith in ariths:
    net = peaked_net(20.0, blocks=blocks_n)
    g = guarded_inference_net(net, torch.float32, trunk="mfma", arith=arith, guard=False, planes=planes_all[:256])
    g.chain_heads = False
    print("==", arith, g.block_kinds())
    for n in [int(v) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else "2,1,3,300".split(","))]:
        g(planes_all[:n].contiguous())
