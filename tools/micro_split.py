#!/usr/bin/env python3
"""GPU: where the 1 M-board rule micro-suite (k_rules_tpb) spends its time -- the same boards through cz_movegen (moves +
counts), cz_done (flags, need_check), cz_encode (planes only; wave-per-board kernel) and cz_rules_fused (everything)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "chinesechess-alphazero_amd"), ROOT]
from cchess_alphazero import _native  # noqa: E402
from cchess_alphazero.environment.static_env import state_to_array  # noqa: E402


def main():
    n = 1 << 20
    with open(os.path.join(ROOT, "tests", "golden", "positions_1k.json")) as f:
        states = [r["state"] for r in json.load(f)["positions"]]
    base = torch.from_numpy(np.stack([state_to_array(s) for s in states])).cuda()
    g = torch.Generator(device="cuda").manual_seed(1)
    boards = base[torch.randint(0, base.shape[0], (n,), device="cuda", generator=g)].contiguous()

    def timed(fn, iters=10):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / iters

    out = _native.rules_fused(boards, _native.F32)
    res = {"fused_f32_ms": timed(lambda: _native.rules_fused(boards, _native.F32, out=out)),
           "movegen_ms": timed(lambda: _native.movegen(boards)),
           "done_need_check_ms": timed(lambda: _native.done(boards, need_check=True)),
           "done_ms": timed(lambda: _native.done(boards, need_check=False)),
           "encode_f32_ms": timed(lambda: _native.encode(boards, _native.F32)),
           "encode_u8_ms": timed(lambda: _native.encode(boards, _native.U8))}
    out8 = _native.rules_fused(boards, _native.U8)
    res["fused_u8_ms"] = timed(lambda: _native.rules_fused(boards, _native.U8, out=out8))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
