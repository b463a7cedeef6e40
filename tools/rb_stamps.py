#!/usr/bin/env python3
"""Where a residual-block launch of the c8 tower spends its shader cycles: a -DCZ_RB_STAMPS build of the library (variant,
never the default) stamps s_memtime in one matrix wave and one copy wave of one workgroup around the sections of one
steady-state board.  Run on the MI355X:

    python chinesechess-alphazero_amd/build.py --out variants/libczero_stamps.so -DCZ_RB_STAMPS      (here, cross-compiles)
    CZ_LIB=variants/libczero_stamps.so python tools/rb_stamps.py
"""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chinesechess-alphazero_amd"))
from cchess_alphazero import _native  # noqa: E402


def main():
    n, c = 32768, 128
    g = torch.Generator(device="cuda").manual_seed(1)
    x = (torch.randn((n, 90, c), device="cuda", generator=g) * 1.5).relu()
    gw = torch.Generator().manual_seed(2)
    w1, w2 = (torch.randn((c, c, 3, 3), generator=gw) / (3.0 * c ** 0.5) for _ in range(2))
    w_in = torch.randn((c, 14, 5, 5), generator=gw) * 0.2
    b = torch.zeros(c, device="cuda")
    p1, p2 = _native.pack_conv3x3_c8_weights(w1).cuda(), _native.pack_conv3x3_c8_weights(w2).cuda()
    table = _native.input_table(w_in).cuda()
    planes = torch.zeros((n, 14, 10, 9), dtype=torch.uint8)
    occ = torch.rand((n, 10, 9), generator=gw) < 0.3
    planes.scatter_(1, torch.randint(0, 14, (n, 1, 10, 9), generator=gw), occ.unsqueeze(1).to(torch.uint8))
    planes = planes.cuda()
    xin = _native.split_c8(x)
    out = (torch.empty_like(xin[0]), torch.empty_like(xin[1]))
    # the c6 arithmetic (bf6 correction operands): its input image comes out of the fused first block
    q1, q2 = _native.pack_conv3x3_c6_weights(w1, 0, 0).cuda(), _native.pack_conv3x3_c6_weights(w2, 0, 0).cuda()
    x6 = (torch.empty_like(xin[0]), torch.empty_like(xin[1]).view(torch.int8))
    out6 = (torch.empty_like(x6[0]), torch.empty_like(x6[1]))
    _native.input_resblock(planes, table, b, p1, b, q2, b, out=x6)
    L = _native.lib()
    L.cz_debug_rb_stamps.argtypes = [C.c_void_p]
    res = {}
    last = torch.empty((n, 90, c), dtype=torch.float32, device="cuda")

    def knob(v, fn):
        def run():
            L.cz_debug_rb_knob(v)
            fn()
            L.cz_debug_rb_knob(0)
        return run
    c6b = lambda: _native.resblock(x6, q1, b, q2, b, out=out6)
    for name, fn in (("block", lambda: _native.resblock(xin, p1, b, p2, b, out=out)),
                     ("first_block_with_fused_input_layer", lambda: _native.input_resblock(planes, table, b, p1, b, p2, b, out=out)),
                     ("c6_block", lambda: _native.resblock(x6, q1, b, q2, b, out=out6)),
                     ("c6_first_block", lambda: _native.input_resblock(planes, table, b, p1, b, q2, b, out=out6)),
                     ("c6_block_f32_out", lambda: _native.resblock(x6, q1, b, q2, b, out_f32=last)),
                     ("c6_block_knob1_no_f16_stores", knob(1, c6b)), ("c6_block_knob2_no_piece_stores", knob(2, c6b)),
                     ("c6_block_knob3_no_stores", knob(3, c6b)), ("c6_block_knob4_no_store_pass", knob(4, c6b)),
                     ("c6_block_knob8_c8_store_pass", knob(8, c6b))):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for _ in range(30):                               # warm: the clock settles under the sustained load
            fn()
        ev[0].record()
        for _ in range(20):
            fn()
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / 20
        st = (C.c_longlong * 32)()
        assert L.cz_debug_rb_stamps(st) == 0
        s = list(st)
        m = {"wait_A": s[1] - s[0], "kloop_1_with_deferred_epilogue_2": s[2] - s[1], "epilogue_1_and_skip_init": s[3] - s[2],
             "wait_B": s[4] - s[3], "kloop_2": s[5] - s[4]}
        cw = {"wait_A": s[17] - s[16], "window_1_loads_or_gather": s[18] - s[17], "wait_B": s[19] - s[18],
              "store_previous_board": s[20] - s[19], "gather_rest": s[21] - s[20] if s[21] > s[20] else 0,
              "first_end_split": s[22] - max(s[21], s[20]), "write_X": s[23] - s[22]}
        if s[24] > s[17]:                                  # FIRST: the first window in detail
            cw["window_1_detail"] = {"first_begin_masks_and_tap_sets": s[24] - s[17], "planes_prefetch_issue": s[25] - s[24],
                                     "term_rounds": s[18] - s[25]}
        us_per_board = ms * 1e3 / (n / 256.0)
        res[name] = {"ms_per_launch": ms, "us_per_board": us_per_board, "matrix_wave_cycles": m, "copy_wave_cycles": cw,
                     "cycles_stamped": sum(m.values()), "effective_GHz": sum(m.values()) / us_per_board / 1e3}
    res["mfma_floor_cycles_per_kloop"] = 13824
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
