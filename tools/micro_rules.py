#!/usr/bin/env python3
"""GPU: run the 1M-board rule micro-suite a few times (for rocprofv3 counter passes)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "chinesechess-alphazero_amd"), ROOT]
import torch
sys.argv = ["bench.py"]
import bench
print(json.dumps(bench.micro_suite(iters=int(os.environ.get("ITERS", "5")))))
