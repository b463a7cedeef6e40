"""The collective path on ONE GPU: RCCL initialisation + the counter all-reduce are the only
multi-GPU code of the path (reference worker/self_play.py:55-60: independent workers per device).  No 8-GPU
node is available to the builder, so these tests make a single rank go through exactly that code on the MI355X: under
the driver's launcher (`torch.distributed.run --nproc-per-node 1`), and without one."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH_ARGS = ["--gpus", "1", "--steps", "2", "--warmup", "1", "--sustained-rounds", "0", "--no-micro",
              "--no-cpu-baseline", "--no-other-configs", "--games", "256"]


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return env


def _line(out):
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert out.returncode == 0 and lines, out.stderr[-3000:]
    # ONE JSON line on stdout and nothing else (RCCL's version banner, which it prints on stdout, is kept off it)
    assert len(lines) == 1 and lines[0].startswith("{"), lines[:8]
    assert len(lines[0]) < 4096                  # the compact line the driver parses (bench.py::compact_line)
    return json.loads(lines[0])


def _full(out):
    """the full record of the same run: bench.py prints it on stderr behind a tag (and writes bench_full.json)"""
    tag = "[bench full record] "
    rec = [l for l in out.stderr.splitlines() if l.startswith(tag)]
    assert len(rec) == 1, out.stderr[-2000:]
    return json.loads(rec[0][len(tag):])


def test_bench_under_the_launcher_with_one_rank_all_reduces_over_rccl():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + BENCH_ARGS
    out = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600, cwd=ROOT)
    d, full = _line(out), _full(out)
    c = d["collective"]
    assert c["backend"] == "nccl" and c["world"] == 1 and c["probe_ok"] is True, c
    assert full["collective"]["launched_by"] == "torch.distributed.run"
    assert d["n_gpus"] == 1 and d["per_rank_value"] == [d["value"]]   # the `ranks` entry of the all-reduced counter vector
    assert d["value"] > 0 and full["tree_shape"]["tree_resets"] == 0
    assert abs(full["value"] - d["value"]) <= 1e-5 * d["value"]      # the compact line rounds to 6 significant digits


def test_bench_without_a_launcher_still_brings_rccl_up():
    d = _line(subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + BENCH_ARGS, env=_env(),
                             capture_output=True, text=True, timeout=600, cwd=ROOT))
    c = d["collective"]
    assert c["backend"] == "nccl" and c["world"] == 1 and c["probe_ok"] is True, c
    assert d["n_gpus"] == 1 and c["all_reduce_int64x8_us"] > 0


def test_worker_counter_reduction_runs_on_rccl_with_one_rank():
    """worker/self_play.py::reduce_counters (the worker's report path) on the nccl backend, world size 1."""
    code = """
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(%r, "chinesechess-alphazero_amd"))
from cchess_alphazero.worker.self_play import reduce_counters, COUNTER_KEYS, game_id_partition
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
c = {k: 10 + i for i, k in enumerate(COUNTER_KEYS)}
r = reduce_counters(c)
assert r == c, (r, c)
assert dist.get_backend() == "nccl" and game_id_partition(0, 1, 4096) == (0, 4096)
dist.barrier(); dist.destroy_process_group()
print("RCCL_OK")
""" % ROOT
    env = dict(_env(), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "RCCL_OK" in out.stdout, out.stderr[-3000:]
