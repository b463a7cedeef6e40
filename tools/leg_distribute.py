#!/usr/bin/env python3
"""A/B of the tower arithmetics on the reference's deployed topology (10 x 192, K = 10, c_puct 5: configs/distribute.py:33-51,84-87):
bench.py's `distribute_10x192_K10_cpuct5` leg for each requested arithmetic, same box, back to back.

    python tools/leg_distribute.py [seconds per leg] [arith ...]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    sec = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
    ariths = sys.argv[2:] or ["bf16x3", "c8", "f16x3", "c8"]
    for a in ariths:
        r = bench.short_selfplay_leg("distribute", "normal", sec, lambda m: None, K=10, arith=a,
                                     model=dict(cnn_filter_num=192, res_layer_num=10),
                                     play=dict(c_puct=5, noise_eps=0.2, max_game_length=200))
        n = r["numerics_check"]
        print(json.dumps({"arith": a, "effective": r["net_arith_effective"], "value": r["value"], "ms_per_step": r["ms_per_step"],
                          "block_ms": r["roofline"].get("avg_launch_ms"), "frac": r["roofline"]["frac"],
                          "by_block": [round(x, 2) for x in r["roofline"].get("launch_ms_by_block") or []],
                          "plan": r["roofline"].get("launch_plan"), "boards": r["roofline"].get("boards_per_launch"),
                          "logit": n["policy_logit_max_abs_diff"], "value_err": n["value_max_abs_diff"],
                          "within": n["within_tolerance"]}), flush=True)


if __name__ == "__main__":
    main()
