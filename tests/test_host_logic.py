"""CPU tests of the host side: C-ABI exports vs include/czero.h, config mirror, label tables, string<->board
bookkeeping, the play-record writer, and the multi-rank path (gloo, world_size 2)."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from oracle import xq_oracle as xo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_exports_every_declared_symbol():
    from cchess_alphazero import _native
    hdr = open(os.path.join(ROOT, "include", "czero.h")).read()
    names = sorted(set(re.findall(r"\b(cz_[a-z_0-9]+)\s*\(", hdr)))
    assert len(names) >= 22
    L = ctypes.CDLL(_native.LIB_PATH)
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/czero.h but not exported"
    assert L.cz_version() == 2


def test_no_cpu_fallback_without_gpu():
    import torch
    from cchess_alphazero import _native
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_native.NativeError):
        _native.movegen(torch.zeros((1, 90), dtype=torch.int8))
    import cchess_alphazero.environment.static_env as senv
    with pytest.raises(_native.NativeError):
        senv.get_legal_moves(senv.INIT_STATE)


def test_label_tables_match_oracle():
    from cchess_alphazero.environment import lookup_tables as lt
    assert lt.ActionLabelsRed == xo.labels()
    assert lt.Unflipped_index[:5] == [2026, 2025, 2024, 2023, 2022]
    assert lt.flip_move('7770') == xo.flip_move('7770')
    pol = np.arange(2086, dtype=np.float64)
    assert (lt.flip_policy(lt.flip_policy(pol)) == pol).all()


def test_string_bookkeeping(positions_1k, known_answers):
    import cchess_alphazero.environment.static_env as senv
    for r in positions_1k:
        s = r["state"]
        arr = senv.state_to_array(s)
        assert (arr == xo.state_to_board(s)).all()
        assert senv.array_to_state(arr) == s
        assert senv.fliped_state(s) == r["flip"]
        assert senv.board_to_state(senv.state_to_board(s)) == s
    c = known_answers["test_check_and_catch"]
    assert senv.fen_to_state(c["fen"]) == c["state"]
    assert senv.state_to_fen(known_answers["step_init_0001"], 1) == \
        'rnbakabnr/9/1c5c1/p1p1p1p1p/9/9/P1P1P1P1P/1C5C1/R8/1NBAKABNR b - - 0 1'
    assert senv.parse_ucci_move('b7b0') == '1710' and senv.to_uci_move('1710') == 'b7b0'
    assert senv.init('9999299949999999249999869999999958999999519999999999999999997699') == \
        '9/5s3/9/9/2R6/9/7pP/9/5r3/2E1S4'


def test_config_mirror():
    from cchess_alphazero.config import Config
    c = Config('normal')
    assert c.play.simulation_num_per_move == 800 and c.play.search_threads == 40 and c.play.virtual_loss == 3
    assert c.model.cnn_filter_num == 256 and c.play_data.nb_game_in_file == 5
    c.eval.update_play_config(c.play)
    assert c.play.search_threads == 8 and c.play.c_puct == 1
    assert Config('distribute').model.cnn_filter_num == 192
    with pytest.raises(RuntimeError):
        Config('nope')
    import cchess_alphazero.configs.mini as m
    assert m.PlayConfig().simulation_num_per_move == 100


def test_play_data_writer(tmp_path, monkeypatch):
    monkeypatch.setenv("DATA_DIR", str(tmp_path))
    from cchess_alphazero.config import Config
    from cchess_alphazero.lib.data_helper import PlayDataWriter, get_game_data_filenames, read_game_data_from_file
    cfg = Config('mini')
    cfg.play_data.max_file_num = 3
    w = PlayDataWriter(cfg)
    game = [xo.INIT_STATE, ['7062', 1], ['6042', -1], ['0001', 1]]
    paths = [w.add_game(game) for _ in range(5)]
    assert all(p is not None for p in paths)              # nb_game_in_file == 1
    files = get_game_data_filenames(cfg.resource)
    assert len(files) == 3                                # pruned to max_file_num
    assert read_game_data_from_file(files[-1]) == game
    assert re.match(r"play_\d{8}-\d{6}\.\d{6}\.json$", os.path.basename(files[-1]))
    cfg.play_data.nb_game_in_file = 2
    w2 = PlayDataWriter(cfg)
    assert w2.add_game(game) is None
    p2 = w2.add_game(game)
    assert read_game_data_from_file(p2) == game + game    # flat concatenation, as the reference writes it


def test_model_shapes_and_fold():
    import torch
    from cchess_alphazero.agent.model import CChessNet, InferenceNet, flops_per_position
    torch.manual_seed(0)
    net = CChessNet(cnn_filter_num=32, res_layer_num=2)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.5)
            m.running_var.uniform_(0.5, 2.0)
            m.weight.data.normal_(1, 0.2)
            m.bias.data.normal_(0, 0.2)
    net.eval()
    x = (torch.rand(6, 14, 10, 9) < 0.05).float()
    p, v = net(x)
    p2, v2 = InferenceNet(net)(x)
    assert p.shape == (6, 2086) and v.shape == (6,)
    assert (p - p2).abs().max() < 1e-5 and (v - v2).abs().max() < 1e-5       # tolerance: 1e-4 (north_star)
    assert abs(flops_per_position(CChessNet(cnn_filter_num=128).cfg) / 1e9 - 0.381) < 0.002
    # Keras topology of data/model/model_128f.json: 7 blocks, 4-filter policy conv, 2-filter value conv
    n128 = CChessNet(cnn_filter_num=128, res_layer_num=7)
    assert n128.policy_conv.out_channels == 4 and n128.value_conv.out_channels == 2 and len(n128.res) == 7
    assert n128.policy_out.in_features == 360 and n128.value_dense.in_features == 180


_GLOO_SCRIPT = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import torch.distributed as dist
from cchess_alphazero.worker.self_play import reduce_counters, game_id_partition, COUNTER_KEYS
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{os.environ['PORT']}", rank=rank, world_size=world)
local = {k: (rank + 1) * (i + 1) for i, k in enumerate(COUNTER_KEYS)}
tot = reduce_counters(local)
assert tot == {k: 3 * (i + 1) for i, k in enumerate(COUNTER_KEYS)}, tot
first, stride = game_id_partition(rank, world, 4096)
ids = {first + g + stride * r for g in range(4096) for r in range(3)}
print("OK", rank, min(ids), len(ids))
dist.destroy_process_group()
'''


def test_multi_rank_counters_gloo(tmp_path):
    script = tmp_path / "gloo_rank.py"
    script.write_text(_GLOO_SCRIPT)
    port = 29600 + os.getpid() % 300
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT,
                                       os.path.join(ROOT, "chinesechess-alphazero_amd")],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    mins = sorted(int(o.split()[2]) for o in outs if o.startswith("OK") or "OK" in o for o in [o[o.index("OK"):]])
    assert mins == [0, 4096]                               # disjoint game-id ranges per rank


def test_weight_packers_on_the_host():
    """cz_conv3x3_pack_weights / cz_input_conv_pack_weights run on the host: the MFMA fragment order
    [K-step][channel tile][lane][8] (k = 16 * step' + 8 * (lane >> 5) + j, output channel = 32 * tile + (lane & 31)),
    the (hi, lo) split with lo = bf16(w - hi), the zero K-steps the kernels prefetch past the end, and the argument
    checks -- without a GPU."""
    import torch
    from cchess_alphazero import _native
    g = torch.Generator().manual_seed(0)
    for c in (32, 128):
        w = torch.randn((c, c, 3, 3), generator=g)
        for dt in (torch.bfloat16, torch.float16):
            for parts in (1, 2):
                p = _native.pack_conv3x3_weights(w, dt, parts)
                kk, ct = c // 16, c // 32
                steps = 9 * kk + 3
                assert p.numel() == parts * steps * ct * 64 * 8 == _native.lib().cz_conv3x3_packed_elems(c, parts)
                v = p.view(parts, steps, ct, 2, 32, 8)                    # [part][step][tile][lane >> 5][lane & 31][j]
                want = w.permute(2, 3, 1, 0).reshape(9, kk, 2, 8, ct, 32)  # [tap][kk][half][j][tile][out]
                want = want.permute(0, 1, 4, 2, 5, 3).reshape(9 * kk, ct, 2, 32, 8)
                hi = want.to(dt)
                assert torch.equal(v[0, :9 * kk], hi)
                assert v[:, 9 * kk:].abs().sum() == 0
                if parts == 2:
                    assert torch.equal(v[1, :9 * kk], (want - hi.float()).to(dt))
    wi = torch.randn((128, 14, 5, 5), generator=g)
    p = _native.pack_input_conv_weights(wi, torch.bfloat16, 2)
    assert p.numel() == 2 * 28 * 4 * 64 * 8 == _native.lib().cz_input_conv_packed_elems(128, 14, 2)
    v = p.view(2, 28, 4, 2, 32, 8)
    pad = torch.zeros((128, 16, 5, 5))
    pad[:, :14] = wi
    want = pad.permute(2, 3, 1, 0).reshape(25, 2, 8, 4, 32).permute(0, 3, 1, 4, 2)      # [tap][tile][half][out][j]
    assert torch.equal(v[0, :25], want.to(torch.bfloat16))
    assert torch.equal(v[1, :25], (want - want.to(torch.bfloat16).float()).to(torch.bfloat16))
    assert v[:, 25:].abs().sum() == 0
    w28 = torch.randn((32, 28, 5, 5), generator=g)
    assert _native.pack_input_conv_weights(w28, torch.float16, 1).numel() == 28 * 2 * 1 * 64 * 8
    for bad in ((48, 2), (128, 3), (0, 1)):
        assert _native.lib().cz_conv3x3_packed_elems(*bad) == 0
    with pytest.raises(_native.NativeError):
        _native.pack_conv3x3_weights(torch.randn(48, 48, 3, 3), torch.bfloat16, 2)
    with pytest.raises(_native.NativeError):
        _native.pack_input_conv_weights(torch.randn(128, 40, 5, 5), torch.bfloat16, 2)


def test_tower_arithmetic_selection_on_the_host(monkeypatch):
    """engine.net_arith = "c6" is the self-play default (c8 where no c6 kernel exists); InferenceNet takes the c8 arithmetic only where its kernels exist
    (hand-written trunk, float32, 128 filters) and otherwise stays on the bf16 pairs; CZ_TOWER_ARITH overrides the
    configuration; the c8 network packs its block filters with cz_conv3x3_c8_pack_weights (byte count of the C-ABI) and
    its input-layer filters as fp16 pairs.  No GPU: construction and packing only."""
    import torch
    from cchess_alphazero import _native
    from cchess_alphazero.agent.model import CChessNet, InferenceNet
    from cchess_alphazero.config import Config
    monkeypatch.delenv("CZ_TOWER_ARITH", raising=False)
    assert Config("normal").engine.net_arith == "c6"          # (bf6 corrections; c8 where no c6 kernel exists)
    net = CChessNet(cnn_filter_num=128, res_layer_num=2)
    inf = InferenceNet(net, torch.float32, trunk="mfma", arith="c8")
    assert inf.arith == "c8" and inf.operand_dtype == torch.float16 and inf.parts == 2
    nb = _native.lib().cz_conv3x3_c8_packed_bytes(128)
    assert nb > 0 and inf.tw0a.numel() * 2 == nb and inf.tw1b.numel() * 2 == nb
    assert inf.in_w.numel() == _native.lib().cz_input_conv_packed_elems(128, 14, 2)
    assert InferenceNet(net, torch.float32, trunk="mfma").arith == "bf16x3"                 # explicit default of the class
    assert InferenceNet(net, torch.float32, trunk="library", arith="c8").arith == "bf16x3"  # no hand-written trunk
    assert InferenceNet(net, torch.float16, trunk="mfma", arith="c8").arith == "bf16x3"     # plain fp16 operands
    # (192 filters: c8 since round 4 -- k_resblock_ip_c8; 256 filters have no c8 kernels: the request degrades to the fp16 pairs)
    assert InferenceNet(CChessNet(cnn_filter_num=192, res_layer_num=1), torch.float32, trunk="mfma", arith="c8").arith == "c8"
    assert InferenceNet(CChessNet(cnn_filter_num=256, res_layer_num=1), torch.float32, trunk="mfma", arith="c8").arith == "f16x3"
    # c6: needs the activation images' exponents (the guard measures them); block 0's first convolution stays a c8 pack
    # (the fused input layer hands it a c8 image), every other filter carries the exponents of the images it reads / writes
    import numpy as np
    import pytest
    from cchess_alphazero.agent.model import c6_exponents
    with pytest.raises(ValueError):
        InferenceNet(net, torch.float32, trunk="mfma", arith="c6")
    assert c6_exponents([3.0, 28.0, 28.1, 0.2, 500.0], headroom=0) == ([0, -7], [1, 5])
    assert c6_exponents([3.0, 28.0, 28.1, 0.2, 500.0]) == ([1, -6], [2, 6])        # one bit kept over the sample's maximum
    i6 = InferenceNet(net, torch.float32, trunk="mfma", arith="c6", act_exps=([-4, -3], [-2, 1]))
    assert i6.c6 and i6.arith == "c8" and i6.arith_name == "c6" and i6.c8_blocks == 2
    tail = lambda t: t.numpy().view(np.uint8)[-16 - 2 * 128:-2 * 128].view(np.int32).tolist()    # (4 ints, then 2 x 128 row shifts)
    assert tail(i6.tw0a)[2:] == [0, 0] and (i6.tw0a == inf.tw0a).all()            # c8 pack
    assert tail(i6.tw0b)[2:] == [-4, -2] and tail(i6.tw1a)[2:] == [-2, -3] and tail(i6.tw1b)[2:] == [-3, 1]
    # 192 filters have c6 since round 6 (the same exponents contract) ...
    with pytest.raises(ValueError):
        InferenceNet(CChessNet(cnn_filter_num=192, res_layer_num=2), torch.float32, trunk="mfma", arith="c6")
    i192 = InferenceNet(CChessNet(cnn_filter_num=192, res_layer_num=2), torch.float32, trunk="mfma", arith="c6", act_exps=([-4, -3], [-2, 1]))
    assert i192.c6 and i192.arith_name == "c6" and i192.block_kinds() == ["c6", "c6"]
    tail192 = lambda t: t.numpy().view(np.uint8)[-16 - 2 * 192:-2 * 192].view(np.int32).tolist()
    assert tail192(i192.tw0a)[2:] == [0, 0] and tail192(i192.tw0b)[2:] == [-4, -2] and tail192(i192.tw1a)[2:] == [-2, -3]
    # ... 256 filters have neither c6 nor c8: the request degrades to the fp16 pairs
    assert InferenceNet(CChessNet(cnn_filter_num=256, res_layer_num=2), torch.float32, trunk="mfma", arith="c6").arith_name == "f16x3"
    assert InferenceNet(CChessNet(cnn_filter_num=128, res_layer_num=1), torch.float32, trunk="mfma", arith="c6").arith_name == "c8"
    monkeypatch.setenv("CZ_TOWER_ARITH", "c8")
    assert InferenceNet(net, torch.float32, trunk="mfma").arith == "c8"
    monkeypatch.setenv("CZ_TOWER_ARITH", "bf16x3")
    assert InferenceNet(net, torch.float32, trunk="mfma").arith == "bf16x3"
    assert _native.lib().cz_conv3x3_c8_packed_bytes(192) == ((9 * 12 + 3) * 6 * 64 + (9 * 3 + 1) * 2 * 6 * 2 * 64 + 1) * 16 + 2 * 192
    assert _native.lib().cz_conv3x3_c8_packed_bytes(256) == 0
    with pytest.raises(_native.NativeError):
        _native.pack_conv3x3_c8_weights(torch.randn(256, 256, 3, 3))


def test_uci_position_parsing_without_a_gpu():
    """The command parser of cchess_alphazero/uci.py (reference uci.py:116-170): `position fen ... w|b`, move counters,
    side to move, unknown commands ignored; nothing here needs the device."""
    import io
    from cchess_alphazero.config import Config
    from cchess_alphazero.environment import static_env as senv
    from cchess_alphazero.uci import UCI
    out = io.StringIO()
    u = UCI(Config(config_type="mini"), out=out)
    u.is_ready = True                                       # skip `uci` (it loads the network)
    u.handle("isready")
    assert out.getvalue().strip() == "readyok"
    fen = "rnbakabnr/9/1c5c1/p1p1p1p1p/9/9/P1P1P1P1P/1C5C1/9/RNBAKABNR"
    u.handle(f"position fen {fen} w - - 0 1")
    assert u.state == senv.INIT_STATE and u.is_red_turn and u.turns == 0 and u.history == [senv.INIT_STATE]
    u.handle(f"fen {fen} b - - 0 7")
    assert not u.is_red_turn and u.turns == 13 and u.state == senv.fliped_state(senv.INIT_STATE)
    u.handle("position fen broken")                          # malformed: the position is kept
    assert u.turns == 13
    u.handle("position startpos")
    assert u.is_red_turn and u.turns == 0 and u.state == senv.INIT_STATE
    u.handle("setoption name Threads value 16")
    assert u.config.play.search_threads == 16
    assert u.handle("no_such_command 1 2") is True
    assert u.handle("quit") is False


def test_bench_configs_follow_baseline():
    """bench.py's workloads are BASELINE.json's configs (network shape, simulations, concurrent games), and the engine
    defaults put the hand-written network kernels on the path."""
    import argparse
    import json
    import bench
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "800 sims/move" in base["configs"][1] and "4096 concurrent games" in base["configs"][1]
    want = {"mini": (2, 32, 50, 1), "normal": (7, 128, 800, 4096), "eval": (7, 128, 400, 200), "deep": (20, 256, 1600, None)}
    for name, (blocks, filters, sims, games) in want.items():
        cfg = bench.build_config(argparse.Namespace(config=name, games=None, sims_per_round=None, dtype=None, trunk=None))
        assert cfg.model.res_layer_num == blocks and cfg.model.cnn_filter_num == filters
        assert cfg.play.simulation_num_per_move == sims
        if games is not None:
            assert cfg.engine.games_per_gpu == games
        assert cfg.engine.net_trunk == "mfma"
    deep = bench.build_config(argparse.Namespace(config="deep", games=None, sims_per_round=None, dtype=None, trunk=None))
    assert deep.engine.net_dtype == "float16"
    lib = bench.build_config(argparse.Namespace(config="normal", games=64, sims_per_round=4, dtype="bfloat16", trunk="library"))
    assert (lib.engine.games_per_gpu, lib.play.search_threads, lib.engine.net_dtype, lib.engine.net_trunk) == \
        (64, 4, "bfloat16", "library")


def test_bench_gpus_flag_spawns_that_many_ranks():
    """`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run with N ranks (the
    driver's own command shape) and reports n_gpus = the number of ranks that took part in the counter all-reduce;
    a launcher that started a different number of ranks is refused.  --dry-run: gloo, no GPU work."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] == (1000 + 2000) / 0.6
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"],
                         env=dict(env, WORLD_SIZE="3", RANK="0"), capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "--gpus 2" in (bad.stderr + bad.stdout)


REQUIRED_LINE_KEYS = ("metric", "value", "unit", "n_gpus", "per_rank_value", "steps", "warmup", "ms_per_step",
                      "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "cpu_baseline")


def test_bench_line_is_compact():
    """the LAST stdout line of bench.py is a compact JSON object the driver can parse (round 4's 30 KB
    line came back `parsed: null`): below 4 KB (hard limit 8 KB), json.loads succeeds, the contract's keys are there --
    for `--dry-run` and `--gpus 2 --dry-run` (eight per_rank_value slots at --gpus 8 follow the same code)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    for extra, n in (([], 1), (["--gpus", "2"], 2)):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dry-run"] + extra, env=env,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        last = r.stdout.rstrip("\n").splitlines()[-1]
        assert len(last) < 4096
        line = json.loads(last)
        for k in REQUIRED_LINE_KEYS:
            assert k in line, k
        assert line["n_gpus"] == n and len(line["per_rank_value"]) == n
        assert os.path.exists(os.path.join(root, line["full_record"]))


def test_compact_line_of_a_fat_record():
    """bench.compact_line on a record shaped like a real N = 1 run with everything in it (600-character kernel
    descriptions, guard report, eight other_configs legs): the line stays below 4 KB, carries roofline and cpu_baseline with
    the keys SURVEY 8(d) / the contract name, the peaked-policy figure beside the headline, and no long strings."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    fat = {"metric": "mcts_node_expansions_per_sec", "value": 1548910.123456789, "unit": "expansions/s", "n_gpus": 8,
           "per_rank_value": [193613.765432] * 8, "steps": 20, "warmup": 5, "ms_per_step": 21.0543219, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f16+2xbf6corr-split/f32acc+f64/i32 tree", "dtype_note": "x" * 900,
           "data": "synthetic", "config": {"workload": "w" * 400, "games_per_gpu": 4096, "sims_per_round": 8,
                                           "queue_slots_per_gpu": 32768, "parallelism": "games sharded over 8 rank(s)"},
           "net_arith_guard": {"candidates": [{"x": 1.0}] * 40, "note": "y" * 3000},
           "roofline": {"kernel": "k" * 700, "kernel_short": "k_resblock_c8<C6>", "bound": "mfma", "achieved": 591.3,
                        "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.2365, "traffic": 3.38e9, "traffic_source": "profiles/x.json",
                        "avg_launch_ms": 2.93, "launches_timed": 140, "note": "n" * 2000},
           "roofline_search": {"kernel": "s" * 200, "bound": "hbm", "achieved": 2066.0, "peak": 8000.0, "unit": "GB/s",
                               "frac": 0.258, "avg_launch_ms": 0.245, "traffic": 2.37e8, "note": "n" * 900},
           "cpu_baseline": {"value": 1.6e6, "unit": "expansions/s", "cores": 16, "kind": "port", "sample": "s" * 500,
                            "cpu_model": "AMD EPYC 9575F 64-Core Processor", "with_network_estimate": {"value": 7425.0, "note": "n" * 400},
                            "reference_python_timing": {"summary": "z" * 2000}},
           "value_sustained": 1534550.0, "net_arith_requested": "c6", "net_arith_effective": "c6",
           "numerics_logit_max_abs": 2.4e-6, "numerics_peaked_arith": "c8>5", "value_peaked_policy": 1.3e6,
           "sustained": {"rounds": 3000, "seconds": 60.7, "note": "n" * 700, "tree_memory": {"a": list(range(200))}},
           "other_configs": {f"leg_{i}": {"value": 1e5 * i, "workload": "w" * 300, "numerics_check": {"t": "u" * 900}}
                             for i in range(9)},
           "numerics_check": {"sharpened": {"guard_candidates_on_calibration_positions": [{"a": 1}] * 30}},
           "micro_suite": {"achieved": 3678.6, "unit": "GB/s", "frac": 0.4598, "ms": 1.537, "boards": 1 << 20, "kernel": "k" * 90}}
    c = bench.compact_line(fat)
    text = json.dumps(c, separators=(",", ":"))
    assert len(text) < 4096, len(text)
    for k in REQUIRED_LINE_KEYS + ("roofline", "value_sustained", "value_peaked_policy", "numerics_peaked_arith",
                                   "net_arith_effective", "numerics_logit_max_abs"):
        assert k in c, k
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "launches_timed", "traffic", "traffic_source"):
        assert k in c["roofline"], k
    for k in ("value", "unit", "cores", "kind", "cpu_model", "with_network_estimate", "sample"):
        assert k in c["cpu_baseline"], k
    assert len(c["per_rank_value"]) == 8 and c["n_gpus"] == 8
    assert "net_arith_guard" not in c and "dtype_note" not in c and "other_configs" not in c

    def longest(o):
        if isinstance(o, str):
            return len(o)
        if isinstance(o, dict):
            return max([longest(v) for v in o.values()] + [0])
        if isinstance(o, list):
            return max([longest(v) for v in o] + [0])
        return 0
    assert longest(c) <= 200


def test_guard_chain_orders_the_candidates_by_exactness():
    """agent/model.py guard_search: which tower arithmetics the load-time guard measures, in which order, for a request, a
    tower's measured activation ranges (c8 image saturates at 448, fp16 pairs overflow at 65504) and the candidates that
    pass; next_more_exact: the request below a running arithmetic after a failed live audit."""
    from cchess_alphazero.agent.model import guard_search, next_more_exact

    def walk(arith, c8b, n, amax, good):
        return guard_search(arith, c8b, n, amax, lambda name: name in good)
    small = [3.0, 9.5]
    assert walk("c6", 7, 7, small, {"c6"}) == ("c6", ["c6"])
    assert walk("c8", 7, 7, small, {"c8", "f16x3"}) == ("c8", ["c8"])
    # round 6's stand-in for a trained network: passes from c8>4 down -- found one block at a time, the c6 hybrids skipped
    peaked = {"c8>4", "c8>3", "c8>2", "c8>1", "f16x3", "bf16x3"}
    assert walk("c6", 7, 7, small, peaked) == ("c8>4", ["c6", "c8", "c8>6", "c8>5", "c8>4"])
    assert walk("c8", 7, 7, small, peaked) == ("c8>4", ["c8", "c8>6", "c8>5", "c8>4"])
    assert walk("c8>3", 3, 7, small, peaked) == ("c8>3", ["c8>3"])
    # c8 passes, c6 does not: the c6 hybrids, most c6 blocks first; c8 itself where none of them does
    assert walk("c6", 7, 7, small, {"c8", "c6>2", "c6>1"}) == ("c6>2", ["c6", "c8", "c6>6", "c6>5", "c6>4", "c6>3", "c6>2"])
    assert walk("c6", 3, 3, small, {"c8"}) == ("c8", ["c6", "c8", "c6>2", "c6>1"])
    assert walk("c6>5", 7, 7, small, {"c8", "c6>4"}) == ("c6>4", ["c6>5", "c8", "c6>4"])
    assert walk("c6", 2, 2, [1.0], {"f16x3"}) == ("f16x3", ["c6", "c8", "f16x3"])
    assert walk("c8>1", 1, 7, small, {"c8>1"}) == ("c8>1", ["c8>1"])                # (not a candidate unless it is the request)
    # nothing passes: every family in turn, then None (the caller's fp32 library trunk)
    assert walk("c6", 3, 3, small, set()) == (None, ["c6", "c8", "c8>2", "f16x3", "bf16x3"])
    assert walk("c8", 5, 7, [3.0], set()) == (None, ["c8>5", "c8>4", "c8>3", "c8>2", "f16x3", "bf16x3"])
    # (bf6 images carry their own exponents: plain c6 is tried beyond 448, the hybrids -- a c8 hand-over image -- are not)
    assert walk("c6", 7, 7, [3.0, 500.0], {"bf16x3"}) == ("bf16x3", ["c6", "f16x3", "bf16x3"])
    assert walk("c6>5", 7, 7, [3.0, 500.0], {"f16x3"}) == ("f16x3", ["f16x3"])
    assert walk("c8", 7, 7, [3.0, 500.0], set()) == (None, ["f16x3", "bf16x3"])
    assert walk("c8", 7, 7, [4.0e4], {"bf16x3"}) == ("bf16x3", ["bf16x3"])
    assert walk("f16x3", 0, 7, [3.0], set()) == (None, ["f16x3", "bf16x3"])
    assert walk("bf16x3", 0, 7, [3.0], {"bf16x3"}) == ("bf16x3", ["bf16x3"])
    calls = []
    guard_search("c6", 7, 7, small, lambda name: calls.append(name) or False)
    assert len(calls) == len(set(calls))                                            # each candidate is measured once
    steps, cur = [], "c6"
    while cur is not None:
        steps.append(cur)
        cur = next_more_exact(cur, 3)
    assert steps == ["c6", "c6>2", "c6>1", "c8", "c8>2", "f16x3", "bf16x3"]
    assert next_more_exact("c8>4", 7) == "c8>3" and next_more_exact("c6", 1) == "c8"
    # power-of-two activation scales from the measured ranges [input, b0 mid, b0 out, b1 mid, b1 out]: a common shift for a
    # tower outside [2^-3, 224] (filters untouched), bounded per-tensor deviations against overflow only
    from cchess_alphazero.agent.model import choose_act_shift
    assert choose_act_shift([5.0, 9.0, 30.0, 3.0, 100.0], 2) == (0, [0, 0])
    assert choose_act_shift([5.0, 0.4, 30.0, 0.01, 100.0], 2) == (0, [0, 0])              # small tensors are left alone
    assert choose_act_shift([5.0, 3000.0, 30.0, 1.0, 100.0], 2) == (-5, [-5, -5])          # one large tensor moves the tower
    assert choose_act_shift([900.0, 9.0, 30.0, 3.0, 100.0], 2) == (-3, [-3, -3])
    sx, sm = choose_act_shift([1e-3, 1e-3, 2e-3, 1e-2, 1e-3], 2)                           # a tiny tower moves up as a whole
    assert sx == sm[0] == sm[1] == 13 and 64.0 <= 1e-2 * 2.0 ** sx <= 128.0


def test_reference_forward_f64_is_the_module_in_float64_on_the_cpu():
    import copy
    import torch
    from cchess_alphazero.agent.model import CChessNet, InferenceNet, reference_forward_f64
    torch.manual_seed(5)
    net = CChessNet(cnn_filter_num=32, res_layer_num=2).eval()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.3)
            m.running_var.uniform_(0.5, 2.0)
    planes = (torch.rand((5, 14, 10, 9)) < 0.07).to(torch.uint8)
    p, v, lg, acts, quants = reference_forward_f64(net, planes, with_activations=True)
    with torch.no_grad():
        pr, vr = copy.deepcopy(net).double()(planes.double())
    assert (p - pr).abs().max().item() < 1e-13 and (v - vr).abs().max().item() < 1e-13 and len(acts) == 5
    # arithmetic names: requests the constructor can serve, degraded where the kernels do not exist
    assert InferenceNet(net, torch.float32, trunk="library", arith="c8").arith_name == "bf16x3"
    n128 = CChessNet(cnn_filter_num=128, res_layer_num=3).eval()
    for req, name in (("c8", "c8"), ("c8>2", "c8>2"), ("c8>3", "c8"), ("c8>0", "f16x3"), ("f16x3", "f16x3"), ("bf16x3", "bf16x3")):
        assert InferenceNet(n128, torch.float32, trunk="mfma", arith=req).arith_name == name, req


def test_self_play_rank_rendezvous(monkeypatch):
    """worker/self_play.py: ranks spawned by `run.py self` meet through a FileStore in a private temporary directory (no
    port to lose between finding and binding it); the torchrun branch is taken only when a launcher really set up a
    rendezvous (RANK + WORLD_SIZE without MASTER_PORT used to make init_process_group fail)."""
    import torch.distributed as dist
    from cchess_alphazero.worker import self_play as sp
    a, b = sp.rendezvous_file(), sp.rendezvous_file()
    assert a != b and os.path.isdir(os.path.dirname(a)) and not os.path.exists(a)
    for k in ("RANK", "WORLD_SIZE", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        monkeypatch.delenv(k, raising=False)
    assert not sp.launched_by_torchrun()
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "1")
    assert not sp.launched_by_torchrun()                      # a scheduler's variables alone: single-process run
    monkeypatch.setenv("MASTER_PORT", "29511")
    assert sp.launched_by_torchrun()
    monkeypatch.delenv("MASTER_PORT")
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "x")
    assert sp.launched_by_torchrun() == dist.is_torchelastic_launched()
    # two processes do rendezvous through such a file (gloo here; nccl on the GPUs)
    import subprocess
    import sys
    store = sp.rendezvous_file()
    code = ("import sys, torch, torch.distributed as d; r = int(sys.argv[1]); "
            f"d.init_process_group('gloo', init_method='file://{store}', rank=r, world_size=2); "
            "t = torch.tensor([r + 1]); d.all_reduce(t); print(int(t)); d.destroy_process_group()")
    ps = [subprocess.Popen([sys.executable, "-c", code, str(r)], stdout=subprocess.PIPE, text=True) for r in range(2)]
    assert [p.communicate(timeout=120)[0].strip().splitlines()[-1] for p in ps] == ["3", "3"]


def test_bench_arithmetic_labels():
    """bench.py: the dtype label and the matrix-pipe cost of a tower arithmetic name (what `dtype` / `roofline` in the line say)."""
    sys.path.insert(0, ROOT) if ROOT not in sys.path else None
    import bench
    assert bench.arith_label("c8") == "f16+2xfp8corr-split/f32acc" and bench.arith_label("f16x3") == "f16x3-split/f32acc"
    assert "first 5 blocks" in bench.arith_label("c8>5") and bench.arith_label(None) is None
    assert bench.arith_mfma_equivalents("c8", 7) == 2.0 and bench.arith_mfma_equivalents("bf16x3", 7) == 3.0
    assert bench.arith_label("c6") == "f16+2xbf6corr-split/f32acc" and abs(bench.arith_mfma_equivalents("c6", 7) - 21.5 / 14) < 1e-12
    assert abs(bench.arith_mfma_equivalents("c8>5", 7) - (2.0 * 5 + 3.0 * 2) / 7) < 1e-12
    assert abs(bench.arith_mfma_equivalents("c6>3", 7) - (2.0 + 1.5 * 5 + 2.0 * 8) / 14) < 1e-12
    assert "first 3 blocks" in bench.arith_label("c6>3") and "bf6" in bench.arith_label("c6>3")
    assert bench.arith_mfma_equivalents("fp32-library", 7) == 1.0


def test_activation_shift_is_an_exact_reparametrisation():
    """InferenceNet(act_shift=(s_x, [s_mid ...])): power-of-two scales of the residual stream and of every block's
    intermediate tensor folded into filters and biases (agent/model.py; ReLU commutes with a positive scale) -- the
    network function is unchanged.  Checked on the CPU through the library trunk, whose folded convolutions are the ones the
    hand-written kernels pack."""
    import torch
    from cchess_alphazero.agent.model import CChessNet, InferenceNet
    torch.manual_seed(8)
    net = CChessNet(cnn_filter_num=32, res_layer_num=3).eval()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.3)
            m.running_var.uniform_(0.5, 2.0)
            m.weight.data.normal_(1, 0.2)
            m.bias.data.normal_(0, 0.2)
    x = (torch.rand((6, 14, 10, 9)) < 0.08).float()
    p0, v0 = InferenceNet(net, torch.float32, trunk="library")(x)
    for shift in ((-5, [3, -2, 0]), (7, [7, 7, 7]), (0, [0, 0, 0])):
        inf = InferenceNet(net, torch.float32, trunk="library", act_shift=shift)
        assert (inf.act_shift is None) == (shift == (0, [0, 0, 0]))
        p, v = inf(x)
        assert (p - p0).abs().max().item() < 1e-6 and (v - v0).abs().max().item() < 1e-6, shift
        # the tower really runs at the shifted scale: its first tensor is 2^s_x times the unshifted one
        a0 = torch.relu(InferenceNet(net, torch.float32, trunk="library").input_conv(x))
        a1 = torch.relu(inf.input_conv(x))
        assert torch.allclose(a1, a0 * 2.0 ** shift[0], rtol=1e-6, atol=1e-9)


def test_occupancy_boards_stand_for_the_planes():
    """cz_search_leaf_masks' convention, pinned on the CPU against the oracle's state_to_planes (static_env.py:137-156): word
    pos = plane position i * 9 + j, bit c = plane c shows a piece there; words 90 .. 95 zero; masks_to_planes (what
    Search.queue_planes rebuilds the audit positions with when the kernel writes boards only) is the exact inverse."""
    import random
    import torch
    from cchess_alphazero._native_search import masks_to_planes, planes_to_masks
    rng = random.Random(7)
    states = [xo.INIT_STATE, '3s5/4m4/9/9/4p4/2R6/9/4C4/4M4/3MS4', '3s5/9/9/9/9/9/9/9/9/3S5']
    s = xo.INIT_STATE
    for _ in range(60):                                  # a random playout: positions with captures and flipped sides
        mv = xo.get_legal_moves(s)
        if not mv:
            break
        s = xo.step(s, rng.choice(mv))
        states.append(s)
    boards = np.stack([xo.state_to_board(x) for x in states])
    planes = torch.from_numpy(xo.batch_rules(boards)["planes"])              # [n, 14, 10, 9] float32, 0 / 1
    masks = planes_to_masks(planes)
    assert masks.dtype == torch.int32 and masks.shape == (len(states), 96) and int(masks[:, 90:].abs().sum()) == 0
    # every square shows at most one piece: a word has at most one bit; the bits of a board = its pieces
    pop = torch.tensor([[bin(int(w)).count("1") for w in row] for row in masks.tolist()])
    assert int(pop.max()) == 1 and (pop.sum(1) == torch.from_numpy((boards != 0).sum(1))).all()
    # the convention itself, from the board: piece p (> 0 mover, < 0 opponent) on square y * 9 + x -> plane (p > 0 ? p - 1 : 6 - p),
    # plane position (9 - y) * 9 + x
    for b, row in zip(boards, masks.tolist()):
        for sq in range(90):
            p = int(b[sq])
            y, x = divmod(sq, 9)
            want = 0 if p == 0 else 1 << (p - 1 if p > 0 else 6 - p)
            assert row[(9 - y) * 9 + x] == want
    back = masks_to_planes(masks, 14, torch.float32)
    assert torch.equal(back, planes)
    # 28 planes: the history block in bits 14 .. 27
    both = torch.cat([planes, planes.flip(0)], 1)
    m28 = planes_to_masks(both)
    assert torch.equal(masks_to_planes(m28, 28, torch.uint8), both.to(torch.uint8))
    assert torch.equal(m28 & 0x3FFF, masks) and torch.equal(m28 >> 14, masks.flip(0))


def test_tower_plan_and_block_events():
    """agent/model.py: which launches a fused 128-filter tower runs as (round 6: every arithmetic chains), and how bench.py
    spreads a chained launch's time over its blocks (host logic only)."""
    from cchess_alphazero.agent.model import events_ms, tower_plan
    c6, c8, pr = "c6", "c8", "pair"
    # the benchmark tower: FIRST | blocks 1 .. 6 with the heads as the chain's exit
    assert tower_plan([c6] * 7) == [("first", 0), ("tower", [1, 2, 3, 4, 5, 6], "heads")]
    assert tower_plan([c6] * 7, chain_heads=False) == [("first", 0), ("tower", [1, 2, 3, 4, 5], "c6"), ("block", 6)]
    assert tower_plan([c6] * 7, heads_exit=False) == tower_plan([c6] * 7, chain_heads=False)
    # c6>5: one chain per arithmetic -- the c6 chain's exit writes the c8 image (the last c6 block hands over)
    assert tower_plan([c6] * 5 + [c8] * 2) == [("first", 0), ("tower", [1, 2, 3, 4], "c8"), ("tower", [5, 6], "heads")]
    assert tower_plan([c6] * 5 + [c8] * 2, chain_heads=False)[1:] == [("tower", [1, 2, 3, 4], "c8"), ("tower", [5], "c8"), ("block", 6)]
    assert tower_plan([c6] + [c8] * 6) == [("first", 0), ("tower", [1, 2, 3, 4, 5, 6], "heads")]
    # c8>3 (what the guard gives a peaked policy): FIRST | c8 blocks, exit = fp16 pairs | the f16x3 blocks with the heads
    assert tower_plan([c8] * 3 + [pr] * 4) == [("first", 0), ("tower", [1, 2], "pair"), ("pairs", [3, 4, 5, 6], True)]
    assert tower_plan([c8] * 3 + [pr] * 4, chain_heads=False) == [("first", 0), ("tower", [1, 2], "pair"),
                                                                   ("pairs", [3, 4, 5], False), ("block", 6)]
    assert tower_plan([pr] * 7) == [("first", 0), ("pairs", [1, 2, 3, 4, 5, 6], True)]
    assert tower_plan([c8] * 6 + [pr]) == [("first", 0), ("tower", [1, 2, 3, 4, 5], "pair"), ("pairs", [6], True)]
    assert tower_plan([c6] * 2) == [("first", 0), ("tower", [1], "heads")]
    assert tower_plan([c6] * 2, chain_heads=False) == [("first", 0), ("block", 1)]
    # at most 8 blocks per launch
    assert tower_plan([c6] * 12) == [("first", 0), ("tower", list(range(1, 9)), "c6"), ("tower", [9, 10, 11], "heads")]
    with pytest.raises(AssertionError):
        tower_plan([c8] + [pr] * 3)                                 # (that tower is not a fused-input tower: model.py n8 != 1)
    for nblk in range(2, 13):
        for a in range(0, nblk + 1):
            for b in range(a, nblk + 1):
                kinds = [c6] * a + [c8] * (b - a) + [pr] * (nblk - b)
                if b == 1 and nblk > 1:
                    continue
                for hx in (True, False):
                    steps = tower_plan(kinds, heads_exit=hx)
                    covered = [0]
                    for st in steps[1:]:
                        covered += st[1] if st[0] != "block" else [st[1]]
                        if st[0] == "tower":
                            assert kinds[st[1][0]] != pr and len({kinds[i] for i in st[1]}) == 1 and len(st[1]) <= 8
                            nxt = st[1][-1] + 1
                            assert st[2] == ("heads" if nxt == nblk else kinds[nxt])
                        if st[0] == "pairs":
                            assert all(kinds[i] == pr for i in st[1]) and st[2] == (st[1][-1] + 1 == nblk)
                    assert covered == list(range(nblk)), (kinds, steps)             # every block exactly once, in order
                    assert (steps[-1][0] == "block") == (not hx)

    # 192 filters (cz_resblock_chain): one launch per arithmetic; a c6 tower's block 0 reads the input layer's c8 image on its own
    from cchess_alphazero.agent.model import ip_segments
    assert ip_segments([c6] * 10) == [("chain", list(range(10)))]               # (the four-wave kernel: block 0 inside, CZ_F16C86)
    assert ip_segments([c6] * 10, first_alone=True) == [("block", [0]), ("chain", list(range(1, 10)))]     # (CZ_IP_PAIR=0)
    assert ip_segments([c8] * 10) == [("chain", list(range(10)))]
    assert ip_segments([c6] * 3 + [c8] * 7) == [("chain", [0, 1, 2]), ("chain", list(range(3, 10)))]
    assert ip_segments([c6] * 3 + [c8] * 7, first_alone=True) == [("block", [0]), ("chain", [1, 2]), ("chain", list(range(3, 10)))]
    assert ip_segments([c8] * 2 + [pr] * 2) == [("chain", [0, 1]), ("chain", [2, 3])]     # (round 6: pair blocks chain too)
    assert ip_segments([c8] * 14)[0] == ("chain", list(range(12)))        # at most 12 blocks per launch
    # pair blocks chain too (round 6: k_tower_pairs4<E, 192>; a chain that ends the tower writes fp32)
    assert ip_segments(["pair"] * 4) == [("chain", [0, 1, 2, 3])]
    assert ip_segments(["c8"] * 2 + ["pair"] * 3) == [("chain", [0, 1]), ("chain", [2, 3, 4])]
    assert ip_segments(["c8"] * 9 + ["pair"]) == [("chain", list(range(9))), ("chain", [9])]

    class Ev:
        def __init__(self, t): self.t = t
        def elapsed_time(self, other): return other.t - self.t
    ms = events_ms([(Ev(0.0), Ev(3.3)), (Ev(3.3), Ev(19.3), 6)])
    assert len(ms) == 7 and abs(ms[0] - 3.3) < 1e-12 and all(abs(x - 16.0 / 6) < 1e-12 for x in ms[1:])


def test_chained_tower_staging_layout_model():
    """The address arithmetic behind k_tower_c6's in-place conversion (csrc/xq_conv.hip, namespace tw), restated: a board's fp32
    result is staged INSIDE its slot's image -- channels 0 .. 63 of pixel q in row q of the f16 part, 64 .. 127 in row q of the
    c6 part -- and the four lanes (32-channel blocks) of a pixel turn the row into the operand triple in place.  What makes
    that safe without any synchronisation among the copy waves: per pixel row, the four lanes' reads cover exactly the row's
    512 bytes, their writes stay inside the same 512 bytes (f16: the whole part-0 row; c6: eight 24-byte pieces of the part-1
    row), and no two rows share a byte."""
    RB, PSTR = 256, (272 + 16) * 256
    stage_off = lambda q, ch: (ch >> 6) * PSTR + q * RB + ((((ch >> 2) & 15) ^ (q & 15)) << 4)
    c6_chunk = lambda kind, blk: 8 * kind + 4 * (blk >> 1) + 2 * (blk & 1)
    c6_lds_off = lambda row, chunk: row * RB + ((chunk ^ (row & 15)) << 4)
    seen = set()
    for q in range(90):
        row_bytes = set(range(q * RB, q * RB + RB)) | set(range(PSTR + q * RB, PSTR + q * RB + RB))
        reads, writes = set(), []
        for blk in range(4):
            for k in range(8):                                   # eight float4 of the lane's 32 channels
                o = stage_off(q, blk * 32 + 4 * k)
                reads |= set(range(o, o + 16))
            for k in range(4):                                   # f16: chunk 4 blk + k of the part-0 row
                o = q * RB + (((blk * 4 + k) ^ (q & 15)) << 4)
                writes.append(set(range(o, o + 16)))
            for kind in range(2):                                # a piece = 16-byte head in its chunk + 8-byte tail in the next
                c = c6_chunk(kind, blk)
                writes.append(set(range(PSTR + c6_lds_off(q, c), PSTR + c6_lds_off(q, c) + 16)))
                writes.append(set(range(PSTR + c6_lds_off(q, c + 1), PSTR + c6_lds_off(q, c + 1) + 8)))
        assert reads == row_bytes, q                             # the staging of a pixel IS its two image rows
        allw = set().union(*writes)
        assert sum(len(w) for w in writes) == len(allw) == 256 + 8 * 24 and allw <= row_bytes, q    # disjoint, inside the row
        assert not (row_bytes & seen), q
        seen |= row_bytes
    # the channel -> (row, chunk) map is one to one: 32 float4 per pixel on 32 distinct chunks
    assert len({stage_off(7, 4 * i) for i in range(32)}) == 32
