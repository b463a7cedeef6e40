#!/usr/bin/env python3
"""Launch times of the c8 residual-block kernels on 32 768 boards, back to back in one process (one box, one thermal state):
plain k_resblock<C8>, the pipelined k_resblock_pipe<C8>, the first block with the fused input layer, cz_input_conv<C8> +
plain.  CZ_LIB selects a variant build of the library (A/B of K-loop schedules).

    python tools/time_resblock_c8.py [rounds]
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chinesechess-alphazero_amd"))
from cchess_alphazero import _native  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    n, c = 32768, 128
    g = torch.Generator(device="cuda").manual_seed(1)
    x = (torch.randn((n, 90, c), device="cuda", generator=g) * 1.5).relu()
    gw = torch.Generator().manual_seed(2)
    w1, w2 = (torch.randn((c, c, 3, 3), generator=gw) / (3.0 * c ** 0.5) for _ in range(2))
    w_in = torch.randn((c, 14, 5, 5), generator=gw) * 0.2
    b = torch.zeros(c, device="cuda")
    p1, p2 = _native.pack_conv3x3_c8_weights(w1).cuda(), _native.pack_conv3x3_c8_weights(w2).cuda()
    table = _native.input_table(w_in).cuda()
    w_in_p = _native.pack_input_conv_weights(w_in, torch.float16, 2).cuda()
    planes = torch.zeros((n, 14, 10, 9), dtype=torch.uint8)
    occ = torch.rand((n, 10, 9), generator=gw) < 0.3
    planes.scatter_(1, torch.randint(0, 14, (n, 1, 10, 9), generator=gw), occ.unsqueeze(1).to(torch.uint8))
    planes = planes.cuda()
    xin = _native.split_c8(x)
    out = (torch.empty_like(xin[0]), torch.empty_like(xin[1]))
    outf = torch.empty((n, 90, c), device="cuda")
    res = {"lib": os.environ.get("CZ_LIB", "default"), "boards": n}
    old = _native.resblock_pipelined(None)

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
        for rep in range(2):                       # twice: the second pass runs in the settled thermal state
            _native.resblock_pipelined(False)
            res[f"plain_ms_{rep}"] = timed(lambda: _native.resblock(xin, p1, b, p2, b, out=out))
            res[f"plain_f32out_ms_{rep}"] = timed(lambda: _native.resblock(xin, p1, b, p2, b, out_f32=outf))
            _native.resblock_pipelined(True)
            res[f"pipe_ms_{rep}"] = timed(lambda: _native.resblock(xin, p1, b, p2, b, out=out))
            res[f"first_fused_ms_{rep}"] = timed(lambda: _native.input_resblock(planes, table, b, p1, b, p2, b, out=out))
            res[f"input_conv_ms_{rep}"] = timed(lambda: _native.input_conv(planes, w_in_p, b, out))
    finally:
        _native.resblock_pipelined(old)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
