#!/usr/bin/env python3
"""The reference's deployed topology (10 x 192, K = 10, c_puct 5) with the peaked-policy stand-in of a trained network: what the
load-time guard picks there and what it runs at.    python tools/leg_distribute_peaked.py [seconds]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

sec = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
r = bench.short_selfplay_leg("distribute_peaked", "normal", sec, lambda m: None, K=10, sharpen=True,
                             model=dict(cnn_filter_num=192, res_layer_num=10), play=dict(c_puct=5, noise_eps=0.2, max_game_length=200))
pk = r.get("peaked_policy") or {}
print(json.dumps({"effective": r["net_arith_effective"], "value": r["value"], "ms_per_step": r["ms_per_step"],
                  "by_block": [round(x, 2) for x in r["roofline"].get("launch_ms_by_block") or []],
                  "plan": r["roofline"].get("launch_plan"), "max_p": pk.get("max_policy_probability"),
                  "candidates": [(c["arith"], round(c["logit_max_abs"], 6)) for c in (pk.get("guard_candidates") or [])]}))
