"""The oracle against the REFERENCE ITSELF, imported live from /root/reference (build container only).

The committed golden vectors (tests/golden/*.json) are fixed samples of the reference's behaviour; these tests draw
NEW random positions (seeds below) and compare the oracle's restatement (oracle/xq_rules.c, oracle/xq_mcts.c through
oracle/xq_oracle.py) with the reference's own functions on each of them:

    get_legal_moves (order included)   static_env.py:256-321      done(need_check=True)        static_env.py:14-62
    new_step for every legal move      static_env.py:77-108       state_to_planes / history    static_env.py:144-191
    fliped_state                       static_env.py:131-142      has_attack_chessman          static_env.py:431-438
    will_check_or_catch / be_catched   static_env.py:390-470      CChessPlayer.action (K = 1)  agent/player.py:139-330
    SelfPlayWorker.start_game          worker/self_play.py:95-212 (whole games: moves, result, resignation, visit CRCs)
    EvaluateWorker.start_game          worker/evaluator.py:147-250 (arena games over two players, via tests/arena_oracle.py)
    CChessPlayer.action, search_threads = 8 / 40, unmodified thread timing (agent/player.py:173-179,238-242)

The comparison runs in a child process (tests/live_reference_check.py): the reference's package is called
cchess_alphazero like this repository's host package.  /root/reference does not exist on the GPU box, so the tests
SKIP there (and they are not gpu tests): the -m gpu suites compare the HIP kernels with the oracle and with the
committed vectors, never with this import.
"""
import os
import subprocess
import sys

import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "cchess_alphazero")),
                                reason="the reference checkout is only present in the build container")
HERE = os.path.dirname(os.path.abspath(__file__))


def run_check(*args, timeout=600):
    r = subprocess.run([sys.executable, os.path.join(HERE, "live_reference_check.py")] + [str(a) for a in args],
                       capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    last = r.stdout.strip().splitlines()[-1].split()
    assert last[0] == "ok" and last[1] == args[0], r.stdout[-500:]
    return int(last[2])


@pytest.mark.parametrize("seed,games,max_plies,capture_bias", [(101, 20, 120, 0.3), (303, 24, 300, 0.9)])
def test_rules_match_the_reference_on_fresh_random_positions(seed, games, max_plies, capture_bias):
    assert run_check("rules", seed, games, max_plies, capture_bias) > 150


def test_history_planes_match_the_reference():
    assert run_check("history", 404, 10, 80, 0.5) > 100


def test_search_matches_the_reference_player_on_fresh_positions():
    assert run_check("mcts", 505, 8, 150) == 8


def test_selfplay_games_match_the_reference_worker_on_fresh_specs():
    assert run_check("games", 606, 6) == 6


def test_arena_games_match_the_reference_evaluator_on_fresh_specs():
    assert run_check("arena", 707, 3) == 3


def test_one_small_search_threads_8_search_matches_the_unmodified_reference():
    """the deferred terminal / repetition backup order of the canonical K > 1 schedule is pinned live by the opt-in
    test below only.  This is its smallest case in the DEFAULT run -- one fresh position, search_threads = 8, 120
    simulations, the reference's own thread timing, two runs -- under a hard timeout (100 s): the reference's threaded search
    needs 2 s or minutes for the same position (its sender thread holds the queue lock, SURVEY C-12), so a timeout is an
    expected failure, a mismatch is a real one."""
    # (one 100 s attempt ended in the xfail on most boxes.  Measured in round 5: the slow mode is a property of the
    #  HOST's state, not of the position -- right after a CPU-heavy test (the arena check before this one) the same command
    #  times out three times in a row, after 25 idle seconds it finishes in 2 s, every time.  So: cool down first, and again
    #  before each retry (a timed-out attempt is itself 8 busy threads).  Any attempt that finishes decides the test.)
    import time
    for attempt_timeout in (25, 25, 30):
        time.sleep(22)
        try:
            n = run_check("kgt1", 909, 1, 8, 120, 2, timeout=attempt_timeout)
        except subprocess.TimeoutExpired:
            continue
        assert n == 1
        return
    pytest.xfail("the unmodified reference's threaded search did not finish in three attempts (25-30 s each, 22 s of idle "
                 "before each; erratic by construction)")


@pytest.mark.skipif(os.environ.get("CZ_LIVE_KGT1") != "1",
                    reason="opt-in (CZ_LIVE_KGT1=1): the unmodified reference's threaded search takes 2 s or 3 minutes for the "
                           "same position, depending on how its sender thread happens to hold the queue lock (SURVEY C-12)")
@pytest.mark.parametrize("seed,positions,K,sims,runs", [(808, 4, 8, 200, 3), (811, 2, 40, 200, 3)])
def test_search_threads_gt_1_matches_the_unmodified_reference(seed, positions, K, sims, runs):
    """search_threads = K > 1 (the production regime: K = 8 in the benchmark, 40 in configs/normal.py): the reference
    with its own thread timing, run alone, is deterministic on these searches -- and the oracle's canonical order
     gives exactly its visit counts.  Not part of the default run (its duration is erratic);
    `CZ_LIVE_KGT1=1 python -m pytest tests/test_oracle_live_reference.py -k search_threads`, or the script directly:
    `python tests/live_reference_check.py kgt1 SEED N_POSITIONS K SIMS RUNS`."""
    assert run_check("kgt1", seed, positions, K, sims, runs, timeout=3600) == positions
