"""Golden vectors for the UCI-facing part of the player, produced by running the REFERENCE's own
CChessPlayer with uci=True, debugging=True against the hash stub network (same environment control as
make_golden_mcts.py): `action(depth=...)`, the final `info depth ... pv ...` line of print_depth_info
(player.py:408-450) and the ponder move uci.py derives from the search tree (uci.py:308-318).

    python tests/golden/make_golden_uci.py   ->  tests/golden/uci_k1.json
"""
import contextlib
import io
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_mcts as G  # noqa: E402  (sets up the reference imports, stubs and thread timing)

senv, ref_player, stub_net = G.senv, G.ref_player, G.stub_net


def run(state, turns, salt, depth, hist=None):
    """hist: the game history [s0, m1, s1, ...] ending in `state` -> a 28-plane (use_history) player, as uci.py builds
    it for the history models (uci.py:179-200: CChessPlayer(..., use_history=self.use_history), action(hist=...))."""
    cfg = G.make_cfg(800, c_puct=1.0)                   # PlayWithHumanConfig: c_puct 1, tau_decay_rate 0, no noise
    pipe = stub_net.StubPipe(G.stub_fn(dict(kind="hash", salt=salt)))
    tree = G.ref_player.defaultdict(G.ref_player.VisitState)
    pl = ref_player.CChessPlayer(cfg, search_tree=tree, pipes=pipe, enable_resign=False, debugging=True, uci=True,
                                 use_history=hist is not None, side=turns % 2)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        action, _ = pl.action(state, turns, depth=depth, hist=list(hist) if hist is not None else None)
    infos = [l for l in buf.getvalue().splitlines() if l.startswith("info depth")]
    last = infos[-1].split()
    pv = last[last.index("pv") + 1:last.index("nps")]
    # ponder: the most visited reply in the node behind the chosen move (first maximum, uci.py:312-318)
    nxt = senv.step(state, action)
    ponder, cnt = None, 0
    if nxt in tree:
        for mov, a in tree[nxt].a.items():
            if a.n > cnt:
                ponder, cnt = mov, a.n
    pl.close(wait=False)
    return dict(state=state, turns=turns, salt=salt, depth=depth, hist=hist, action=action, info_lines=len(infos),
                final_depth=int(last[2]), pv=pv, ponder=ponder, done_tasks=depth,
                score=int(last[last.index("score") + 1]),            # network value of the END of the line, seen from `side`
                scores=[int(l.split()[l.split().index("score") + 1]) for l in infos])


def main():
    s1 = senv.step(senv.INIT_STATE, "7770")            # black to move after h2e2 (state is always in the mover's frame)
    cases = [run(senv.INIT_STATE, 0, 3, 300), run(s1, 1, 5, 200), run(senv.INIT_STATE, 0, 8, 100)]
    # 28-plane history models: a game history of two plies (hist[-5] exists: the root is evaluated with the position
    # two plies back) and the opening (no such position: planes 14-27 zero at the root)
    s2 = senv.step(s1, "7062")
    cases.append(run(s2, 2, 11, 200, hist=[senv.INIT_STATE, "7770", s1, "7062", s2]))
    cases.append(run(senv.INIT_STATE, 0, 12, 200, hist=[senv.INIT_STATE]))
    out = dict(meta=G.meta(), cases=cases)
    with open(os.path.join(HERE, "uci_k1.json"), "w") as f:
        json.dump(out, f, indent=1)
    for c in cases:
        print(c["action"], c["final_depth"], c["info_lines"], c["pv"][:6], c["ponder"])


if __name__ == "__main__":
    main()
