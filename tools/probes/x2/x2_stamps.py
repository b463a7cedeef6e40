#!/usr/bin/env python3
"""Shader-cycle stamps of one steady-state PAIR of k_resblock_c8x2 (a -DCZ_RB_STAMPS variant build; blockIdx 5, iteration 20):
    bash tools/build_variant.sh x2stamps -DCZ_RB_STAMPS
    CZ_LIB=variants/libczero_x2stamps.so python tools/x2_stamps.py
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
    gw = torch.Generator().manual_seed(2)
    w1, w2 = (torch.randn((c, c, 3, 3), generator=gw) / (3.0 * c ** 0.5) for _ in range(2))
    w_in = torch.randn((c, 14, 5, 5), generator=gw) * 0.2
    b = torch.zeros(c, device="cuda")
    table = _native.input_table(w_in).cuda()
    planes = torch.zeros((n, 14, 10, 9), dtype=torch.uint8)
    occ = torch.rand((n, 10, 9), generator=gw) < 0.3
    planes.scatter_(1, torch.randint(0, 14, (n, 1, 10, 9), generator=gw), occ.unsqueeze(1).to(torch.uint8))
    planes = planes.cuda()
    L = _native.lib()
    L.cz_debug_rb_stamps.argtypes = [C.c_void_p]
    res = {}
    p8a, p8b = _native.pack_conv3x3_c8_weights(w1).cuda(), _native.pack_conv3x3_c8_weights(w2).cuda()
    for fmt in ("c6", "c8"):
        tag = torch.int8 if fmt == "c6" else torch.uint8
        mk = lambda: (torch.zeros((n, 90, c), dtype=torch.float16, device="cuda"),
                      torch.zeros((n, 90, 2 * c), dtype=tag, device="cuda"))
        x, y = mk(), mk()
        if fmt == "c6":
            q1, q2 = _native.pack_conv3x3_c6_weights(w1, 3, 3).cuda(), _native.pack_conv3x3_c6_weights(w2, 3, 3).cuda()
        else:
            q1, q2 = p8a, p8b
        _native.input_resblock(planes, table, b, p8a, b, q2, b, out=x)
        _native.resblock_x2(2)
        for _ in range(10):
            _native.resblock(x, q1, b, q2, b, out=y)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(20):
            _native.resblock(x, q1, b, q2, b, out=y)
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / 20
        st = (C.c_longlong * 32)()
        assert L.cz_debug_rb_stamps(C.cast(st, C.c_void_p)) == 0
        s = list(st)
        m = {"wait_B0": s[1] - s[0], "kloop_1": s[2] - s[1], "wait_B1": s[3] - s[2], "epilogue_1_x2": s[4] - s[3],
             "wait_B2": s[5] - s[4], "kloop_2": s[6] - s[5], "wait_B3": s[7] - s[6], "epilogue_2_x2": s[8] - s[7],
             "wait_B4": s[9] - s[8]}
        cp = {"wait_B0": s[17] - s[16], "window_1_drain_F_and_issue_loads": s[18] - s[17], "wait_B1_B2": s[19] - s[18],
              "window_2_fill_F_and_issue_loads": s[20] - s[19], "wait_B3_B4": s[21] - s[20], "exposed_drain_A": s[22] - s[21],
              "exposed_fill_A": s[23] - s[22]}
        total = s[9] - s[0]
        res[fmt] = {"ms_per_launch": ms, "us_per_pair": ms * 1e3 / (n / 2 / 256), "matrix_wave_cycles": m, "copy_wave_cycles": cp,
                    "cycles_stamped_per_pair": total, "effective_GHz": total / (ms * 1e3 / (n / 2 / 256)) / 1e3}
    _native.resblock_x2(1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
