#!/usr/bin/env python3
"""GPU: the tree kernels alone in a sustained state, quickly.  4096 self-play games (normal search settings) driven by
the hash-stub network (tests/stub_net.py: microseconds per round instead of 30 ms), so that a few thousand rounds --
games in every phase, trees tens of thousands of nodes deep -- take seconds; then the search round is timed with HIP
events over the next rounds (per-launch: mean / median / p99).  A/B tool for changes to csrc/xq_search.hip.

    python tools/search_probe.py [--rounds 3000] [--timed 200] [--compact 1] [--masks-only 1]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "chinesechess-alphazero_amd"), ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3000)
    ap.add_argument("--timed", type=int, default=200)
    ap.add_argument("--compact", type=int, default=1)
    ap.add_argument("--games", type=int, default=4096)
    ap.add_argument("--masks-only", type=int, default=0, help="1: leaves written as occupancy boards only (cz_search_leaf_planes(0))")
    a = ap.parse_args()
    import types
    import stub_net
    from cchess_alphazero import _native
    from cchess_alphazero._native_search import Search
    pc = types.SimpleNamespace(simulation_num_per_move=800, search_threads=8, c_puct=1.5, noise_eps=0.15,
                               dirichlet_alpha=0.2, tau_decay_rate=0.9, virtual_loss=3, resign_threshold=-0.98,
                               min_resign_turn=40, max_game_length=100, enable_resign_rate=0.5)
    s = Search(pc, a.games, planes_dtype=_native.U8, seed=20260923)
    if a.masks_only:
        s.leaf_masks(True)
        s.leaf_planes(False)
    s.start_selfplay(seed=20260923)
    planes_of = (lambda: s.queue_planes()) if a.masks_only else (lambda: s.planes)   # (the same planes either way: same trees)

    def step():
        s.round(compact=bool(a.compact))
        # a near-uniform, position-dependent network: cheap, and the trees grow like the random-init net's
        p, v = stub_net.hash_stub_torch(planes_of(), 1)
        p = p * 0 + 1.0 / 2086 + p * 1e-3
        if a.compact:
            n = s.slots
            rows = s.q_rows.long().clamp_(0, n - 1)
            s.policy.copy_(p.index_select(0, rows))
            s.value.copy_((v * 0.05).index_select(0, rows))
        else:
            s.policy.copy_(p)
            s.value.copy_(v * 0.05)

    for _ in range(a.rounds):
        step()
    torch.cuda.synchronize()
    ev = []
    for _ in range(a.timed):
        e = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        e[0].record()
        s.round(compact=bool(a.compact))
        e[1].record()
        ev.append(e)
        p, v = stub_net.hash_stub_torch(planes_of(), 1)
        p = p * 0 + 1.0 / 2086 + p * 1e-3
        if a.compact:
            rows = s.q_rows.long().clamp_(0, s.slots - 1)
            s.policy.copy_(p.index_select(0, rows))
            s.value.copy_((v * 0.05).index_select(0, rows))
        else:
            s.policy.copy_(p)
            s.value.copy_(v * 0.05)
    torch.cuda.synchronize()
    t = sorted(x.elapsed_time(y) for x, y in ev)
    c, m = s.counters(), s.memory_info()
    out = {"rounds_before": a.rounds, "timed": a.timed, "compact": a.compact, "masks_only": a.masks_only,
           "search_round_ms": {"mean": sum(t) / len(t), "median": t[len(t) // 2], "p99": t[int(len(t) * 0.99)], "max": t[-1]},
           "mean_depth": c["sum_depth"] / max(1, c["sims"]), "plies": c["plies"], "games": c["games"],
           "tree_resets": c["tree_resets"], "nodes": m["nodes"], "tree_gb": m["tree_bytes"] / 1e9}
    if "cyc_select" in c:                  # CZ_SIM_PROFILE build: shader-clock cycles of wave time per section, per simulation
        n = max(1, c["sims"])
        out["cycles_per_sim"] = {k: c[k] / n for k in c if k.startswith("cyc_")}
        out["levels_per_sim"] = c["sum_depth"] / n
    print(json.dumps(out))
    s.close()


if __name__ == "__main__":
    main()
