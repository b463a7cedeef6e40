#!/usr/bin/env python3
"""A/B of the deep tower's launches (BASELINE configs[4]: 20 x 256, fp16): bench.py's `deep_20x256_fp16_1600sims` leg, one process
per variant (the library reads its switches once), same box, alternating.

    python tools/leg_deep.py [seconds per leg] [reps]            # CZ_TOWER_PLAIN_PAIR = 0 / 1
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(sec):
    import bench
    r = bench.short_selfplay_leg("deep", "deep", sec, lambda m: None)
    n = r["numerics_check"]
    print(json.dumps({"pair": os.environ.get("CZ_TOWER_PLAIN_PAIR", "1"), "value": r["value"], "ms_per_step": r["ms_per_step"],
                      "block_ms": r["roofline"].get("avg_launch_ms"), "frac": r["roofline"]["frac"],
                      "logit": n["policy_logit_max_abs_diff"], "within": n["within_tolerance"]}), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        return one(float(sys.argv[2]))
    sec = sys.argv[1] if len(sys.argv) > 1 else "8"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    for _ in range(reps):
        for v in ("0", "1"):
            env = dict(os.environ, CZ_TOWER_PLAIN_PAIR=v)
            out = subprocess.run([sys.executable, __file__, "--one", sec], env=env, capture_output=True, text=True, timeout=600)
            line = [x for x in out.stdout.splitlines() if x.startswith("{")]
            print(line[-1] if line else "FAILED: " + out.stderr[-400:], flush=True)


if __name__ == "__main__":
    main()
