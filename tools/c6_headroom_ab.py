#!/usr/bin/env python3
"""c6_exponents keeps no headroom over the calibration sample's largest activation.  What does a bit of
headroom cost?  The benchmark network (7 x 128, seed 0) and a peaked one, c6 with 0 / 1 / 2 bits of headroom against float64
on the calibration positions and on fresh ones; plus how many activations of FRESH positions exceed the calibrated range
(what the headroom would protect)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chinesechess-alphazero_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cchess_alphazero.agent import model as M  # noqa: E402


def main():
    from test_gpu_guard import peaked_net
    res = {}
    torch.manual_seed(0)
    nets = {"benchmark_7x128_seed0": M.CChessNet(cnn_filter_num=128, res_layer_num=7).eval(), "peaked_x60": peaked_net(60.0)}
    cal = M.calibration_planes(256, 14)
    fresh = M.calibration_planes(1024, 14, seed=99)
    for name, net in nets.items():
        ref_c = M.reference_forward_f64(net, cal, with_activations=True)
        ref_f = None
        acts_f = []
        for lo in range(0, fresh.shape[0], 256):                   # (float64 unfold: 256 positions at a time)
            r = M.reference_forward_f64(net, fresh[lo:lo + 256], with_activations=True)
            acts_f.append(r[3])
            ref_f = r if ref_f is None else tuple(torch.cat([a, b]) if torch.is_tensor(a) else a for a, b in zip(ref_f, r))
        fresh_max = [max(col) for col in zip(*acts_f)]
        rec = {"calibration_activation_max": ref_c[3], "fresh_activation_max": fresh_max,
               "tensors_where_fresh_exceeds_calibration": sum(f > c for f, c in zip(fresh_max, ref_c[3])),
               "largest_fresh_over_calibration": max(f / c for f, c in zip(fresh_max, ref_c[3]))}
        for h in (0, 1, 2):
            exps = M.c6_exponents(ref_c[3], headroom=h)
            inf = M.InferenceNet(net, torch.float32, trunk="mfma", arith="c6", act_exps=exps).cuda()
            rec[f"headroom_{h}"] = {"calibration": M.measure_against_reference(inf, ref_c, cal),
                                    "fresh_1024": M.measure_against_reference(inf, ref_f[:3], fresh)}
        res[name] = rec
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
