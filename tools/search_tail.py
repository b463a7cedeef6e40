#!/usr/bin/env python3
"""GPU, CZ_SIM_PROFILE build: which game's wavefront does a search launch wait for, and in which section of a simulation?

A k_sim launch ends with its slowest wave (one wave per game).  With the per-game counters read before and after every
round (cz_search_game_counters) the wave with the most cycles in k_sim(SELECT) / k_sim(BACKUP) of that round is known,
together with what it did there: simulations that ended on terminal / repeated positions, expansions, resumed (parked)
simulations, levels walked, and the cycles per section (PUCT descent, rule code, hash, expansion, repetition scoring).
The sustained state is reached with the hash-stub network (tools/search_probe.py).

    bash tools/build_variant.sh prof -DCZ_SIM_PROFILE
    CZ_LIB=$PWD/variants/libczero_prof.so python tools/search_tail.py [--rounds 3000] [--timed 300]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "chinesechess-alphazero_amd"), ROOT, os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3000)
    ap.add_argument("--timed", type=int, default=300)
    ap.add_argument("--games", type=int, default=4096)
    a = ap.parse_args()
    import types
    import stub_net
    from cchess_alphazero import _native
    from cchess_alphazero._native_search import COUNTER_NAMES, Search
    pc = types.SimpleNamespace(simulation_num_per_move=800, search_threads=8, c_puct=1.5, noise_eps=0.15,
                               dirichlet_alpha=0.2, tau_decay_rate=0.9, virtual_loss=3, resign_threshold=-0.98,
                               min_resign_turn=40, max_game_length=100, enable_resign_rate=0.5)
    s = Search(pc, a.games, planes_dtype=_native.U8, seed=20260923)
    s.start_selfplay(seed=20260923)
    names = COUNTER_NAMES[:s.n_counters]
    if "cyc_kernel_select" not in names:
        raise SystemExit("this library has no section timers: build it with -DCZ_SIM_PROFILE and select it with CZ_LIB")
    col = {k: i for i, k in enumerate(names)}

    def feed():
        p, v = stub_net.hash_stub_torch(s.planes, 1)
        p = p * 0 + 1.0 / 2086 + p * 1e-3
        rows = s.q_rows.long().clamp_(0, s.slots - 1)
        s.policy.copy_(p.index_select(0, rows))
        s.value.copy_((v * 0.05).index_select(0, rows))

    for _ in range(a.rounds):
        s.round(compact=True)
        feed()
    torch.cuda.synchronize()
    keys = ["sims", "expansions", "terminal_sims", "repetition_sims", "parked", "sum_depth", "edges_visited", "leaf_moves",
            "plies", "cyc_select", "cyc_rules", "cyc_hash", "cyc_expand", "cyc_rep", "cyc_attach", "cyc_resume_load",
            "cyc_kernel_select", "cyc_kernel_backup"]
    slow = {"select": [], "backup": []}
    typical = {"select": [], "backup": []}
    ratio = {"select": [], "backup": []}
    prev = s.game_counters().astype(np.int64)
    for _ in range(a.timed):
        s.round(compact=True)
        cur = s.game_counters().astype(np.int64)
        d = cur - prev
        prev = cur
        feed()
        for which in ("select", "backup"):
            cyc = d[:, col["cyc_kernel_" + which]]
            g = int(cyc.argmax())
            busy = cyc > 0
            if not busy.any():
                continue
            med = float(np.median(cyc[busy]))
            slow[which].append({k: int(d[g, col[k]]) for k in keys})
            typical[which].append({k: float(np.median(d[busy, col[k]])) for k in keys})
            ratio[which].append(float(cyc[g]) / max(med, 1.0))

    def mean_of(rows):
        return {k: round(sum(r[k] for r in rows) / max(1, len(rows)), 1) for k in keys}

    def share(rows, pred):
        return round(sum(1 for r in rows if pred(r)) / max(1, len(rows)), 3)
    out = {"rounds_before": a.rounds, "timed": a.timed, "games": a.games}
    for which in ("select", "backup"):
        rows = slow[which]
        out[which] = {
            "slowest_wave_mean": mean_of(rows), "median_wave_mean": mean_of(typical[which]),
            "slowest_over_median_cycles": round(sum(ratio[which]) / max(1, len(ratio[which])), 2),
            "slowest_wave_has": {
                "a_repetition_sim": share(rows, lambda r: r["repetition_sims"] > 0),
                "a_terminal_sim": share(rows, lambda r: r["terminal_sims"] > 0),
                "a_ply_played (k_advance ran before it)": share(rows, lambda r: r["plies"] > 0),
                "more_than_8_sims (a chained batch)": share(rows, lambda r: r["sims"] > 8),
                "rep_section_over_30pct": share(rows, lambda r: r["cyc_rep"] > 0.3 * max(1, r["cyc_kernel_" + which])),
                "rules_section_over_50pct": share(rows, lambda r: r["cyc_rules"] > 0.5 * max(1, r["cyc_kernel_" + which])),
                "select_section_over_50pct": share(rows, lambda r: r["cyc_select"] > 0.5 * max(1, r["cyc_kernel_" + which])),
            }}
    print(json.dumps(out))
    s.close()


if __name__ == "__main__":
    main()
