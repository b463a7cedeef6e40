#!/usr/bin/env python3
"""k_resblock_c8 (one board per iteration) against k_resblock_c8x2 (two boards per filter fragment) on 32 768 boards, c8 and c6
operands, alternating in one process (one box, one thermal state): ms per launch, and a byte comparison of the two kernels'
outputs (f16 part: every byte; image part: the bytes that carry data -- a c6 row has 8 unused bytes behind every piece).

    python tools/time_resblock_x2.py [rounds] [boards]
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chinesechess-alphazero_amd"))
from cchess_alphazero import _native  # noqa: E402


def c6_data_mask(device):
    """bool [256]: bytes of a c6 image row that carry data (piece heads: 16 bytes at chunk 8 kind + 4 b + 2 kb, tails: the
    first 8 bytes of the next chunk)"""
    m = torch.zeros(256, dtype=torch.bool, device=device)
    for kind in range(2):
        for blk in range(4):
            c = 8 * kind + 4 * (blk >> 1) + 2 * (blk & 1)
            m[16 * c:16 * c + 24] = True
    return m


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
    c = 128
    gw = torch.Generator().manual_seed(2)
    w1, w2 = (torch.randn((c, c, 3, 3), generator=gw) / (3.0 * c ** 0.5) for _ in range(2))
    w_in = torch.randn((c, 14, 5, 5), generator=gw) * 0.2
    b1 = (torch.randn(c, generator=gw) * 0.1).cuda()
    b2 = (torch.randn(c, generator=gw) * 0.1).cuda()
    bz = torch.zeros(c, device="cuda")
    table = _native.input_table(w_in).cuda()
    planes = torch.zeros((n, 14, 10, 9), dtype=torch.uint8)
    occ = torch.rand((n, 10, 9), generator=gw) < 0.3
    planes.scatter_(1, torch.randint(0, 14, (n, 1, 10, 9), generator=gw), occ.unsqueeze(1).to(torch.uint8))
    planes = planes.cuda()
    res = {"boards": n, "rounds": rounds}
    old = _native.resblock_x2(None)

    def timed(fn):
        for _ in range(12):
            fn()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(rounds):
            fn()
        ev[1].record()
        torch.cuda.synchronize()
        return ev[0].elapsed_time(ev[1]) / rounds

    try:
        for fmt in ("c6", "c8"):
            tag = torch.int8 if fmt == "c6" else torch.uint8
            mk = lambda: (torch.zeros((n, 90, c), dtype=torch.float16, device="cuda"),
                          torch.zeros((n, 90, 2 * c), dtype=tag, device="cuda"))
            x, ya, yb = mk(), mk(), mk()
            p8 = _native.pack_conv3x3_c8_weights(w1).cuda()
            if fmt == "c6":
                k = 3                                               # 2^3 * 28 = 224: generous for these activations
                p_first2 = _native.pack_conv3x3_c6_weights(w2, k, k).cuda()
                q1, q2 = _native.pack_conv3x3_c6_weights(w1, k, k).cuda(), _native.pack_conv3x3_c6_weights(w2, k, k).cuda()
            else:
                p_first2 = _native.pack_conv3x3_c8_weights(w2).cuda()
                q1, q2 = p8, p_first2
            _native.input_resblock(planes, table, bz, p8, b1, p_first2, b2, out=x)       # a real operand image of this format
            torch.cuda.synchronize()
            _native.resblock_x2(0)
            _native.resblock(x, q1, b1, q2, b2, out=ya)
            _native.resblock_x2(2)
            _native.resblock(x, q1, b1, q2, b2, out=yb)
            torch.cuda.synchronize()
            mask = c6_data_mask("cuda") if fmt == "c6" else torch.ones(256, dtype=torch.bool, device="cuda")
            dh = (ya[0].view(torch.int16) != yb[0].view(torch.int16))
            dc = (ya[1].view(torch.uint8) != yb[1].view(torch.uint8)) & mask
            res[f"{fmt}_f16_mismatches"] = int(dh.sum())
            res[f"{fmt}_image_mismatches"] = int(dc.sum())
            res[f"{fmt}_boards_with_mismatch"] = int((dh.flatten(1).any(1) | dc.flatten(1).any(1)).sum())
            res[f"{fmt}_output_nonzero_frac"] = float((ya[0] != 0).float().mean())
            if res[f"{fmt}_boards_with_mismatch"]:
                # where: by pixel tile (rows 0-31, 32-63, 64-89) and by the board's place in its pair
                for name, d in (("f16", dh), ("img", dc)):
                    per = d.flatten(2).any(2) if d.dim() == 3 else d
                    res[f"{fmt}_{name}_bad_rows_by_tile_even_board"] = [int(per[0::2, a:b].sum()) for a, b in ((0, 32), (32, 64), (64, 90))]
                    res[f"{fmt}_{name}_bad_rows_by_tile_odd_board"] = [int(per[1::2, a:b].sum()) for a, b in ((0, 32), (32, 64), (64, 90))]
                xa, xb = ya[0].float(), yb[0].float()
                res[f"{fmt}_f16_max_abs_diff"] = float((xa - xb).abs().max())
                res[f"{fmt}_f16_max_abs"] = float(xa.abs().max())
                bad = (dh.flatten(1).any(1) | dc.flatten(1).any(1)).nonzero().flatten()
                res[f"{fmt}_first_bad_boards"] = bad[:16].tolist()
                b0 = int(bad[0])
                res[f"{fmt}_bad_board_rows_f16"] = dh[b0].any(1).nonzero().flatten()[:20].tolist()
                res[f"{fmt}_bad_board_cols_f16"] = dh[b0].any(0).nonzero().flatten()[:20].tolist()
                res[f"{fmt}_bad_board_rows_img"] = dc[b0].any(1).nonzero().flatten()[:20].tolist()
                res[f"{fmt}_bad_board_cols_img"] = dc[b0].any(0).nonzero().flatten()[:40].tolist()
            for rep in range(2):
                _native.resblock_x2(0)
                res[f"{fmt}_one_board_ms_{rep}"] = timed(lambda: _native.resblock(x, q1, b1, q2, b2, out=ya))
                _native.resblock_x2(2)
                res[f"{fmt}_two_board_ms_{rep}"] = timed(lambda: _native.resblock(x, q1, b1, q2, b2, out=yb))
    finally:
        _native.resblock_x2(old)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
