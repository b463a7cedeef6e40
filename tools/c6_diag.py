#!/usr/bin/env python3
"""Diagnostics of the c6 tower (GPU) against the CPU emulations of the operand models: a table of max |centred logit|
differences between  f64 | emulated c6 | emulated c8 | emulated fp16-only | GPU c6 | GPU c8  on the same positions."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chinesechess-alphazero_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)


def main():
    import emulate_fp8_corrections as emu
    from test_gpu_guard import peaked_net
    from cchess_alphazero.agent.model import calibration_planes, guarded_inference_net, reference_forward_f64

    def cl(p):
        lg = torch.log(p.double().clamp_min(1e-300)).cpu()
        return lg - lg.mean(1, keepdim=True)
    out = {}
    for blocks, scale in ((2, 20.0), (3, 20.0), (3, 1.0)):
        net = peaked_net(scale, blocks=blocks)
        planes = calibration_planes(40, 14, seed=5)
        ref = reference_forward_f64(net, planes)
        g6 = guarded_inference_net(net, torch.float32, trunk="mfma", arith="c6", guard=False, planes=planes)
        g8 = guarded_inference_net(net, torch.float32, trunk="mfma", arith="c8", guard=False)
        rows = {"f64": cl(ref[0]), "gpu_c6": cl(g6(planes)[0]), "gpu_c8": cl(g8(planes)[0])}
        for mode in ("c6-kernel", "c8-kernel", "f16"):
            kw = {"exps": g6.act_exps} if mode == "c6-kernel" else {}
            rows["emu_" + mode] = cl(emu.run(net, planes.cpu(), mode, **kw)[2])
        names = list(rows)
        tab = {a: {b: float((rows[a] - rows[b]).abs().max()) for b in names} for a in names}
        out[f"{blocks}x{scale:g}"] = {"exps": g6.act_exps, "act_shift": g6.act_shift, "table": tab}
        print(f"--- {blocks} blocks, policy x{scale:g}; exps {g6.act_exps} shift {g6.act_shift}")
        print(" " * 14 + " ".join(f"{n:>13s}" for n in names))
        for a in names:
            print(f"{a:>14s} " + " ".join(f"{tab[a][b]:13.2e}" for b in names))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "c6_diag.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
