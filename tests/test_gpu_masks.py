"""-m gpu: the leaves' occupancy boards (round 5: cz_search_leaf_masks + cz_input_resblock_m).  The search kernel writes, beside
the input planes of every new leaf (state_to_planes / state_history_to_planes, environment/static_env.py:137-194), the same
position as 96 words -- word = plane position, bit c = plane c shows a piece there -- and the first residual block's fused input
layer (a gather over the occupied squares, agent/model.py:36-39 of the reference) takes them instead of scanning the 1260
(2520) plane bytes.  Exactness: the boards equal the planes bit for bit on every leaf row; the network's outputs with and
without them are IDENTICAL (the gather adds the same table rows in the same order)."""
import pytest

pytestmark = pytest.mark.gpu


def _masks_from_planes(planes):
    import torch
    n, c = planes.shape[0], planes.shape[1]
    bits = (planes.reshape(n, c, 90) != 0).to(torch.int64)
    w = (1 << torch.arange(c, device=planes.device, dtype=torch.int64)).view(1, c, 1)
    m = (bits * w).sum(1)                                       # [n, 90]
    out = torch.zeros((n, 96), dtype=torch.int64, device=planes.device)
    out[:, :90] = m
    return out.to(torch.int32)


@pytest.mark.parametrize("use_history", [False, True])
def test_leaf_masks_equal_the_planes(use_history):
    import types
    import torch
    import stub_net
    from cchess_alphazero import _native, _native_search
    pc = types.SimpleNamespace(simulation_num_per_move=48, search_threads=6, c_puct=1.5, noise_eps=0.0, dirichlet_alpha=0.2,
                               tau_decay_rate=0.0, virtual_loss=3, resign_threshold=-0.92, min_resign_turn=20,
                               max_game_length=40, enable_resign_rate=1.0)
    s = _native_search.Search(pc, 48, seed=5, planes_dtype=_native.U8, use_history=use_history)
    s.leaf_masks(True)
    assert s.masks.shape == (48 * 6, 96) and s.masks.dtype == torch.int32
    s.start_selfplay(seed=5)
    checked = 0
    for r in range(40):                                         # several plies: history planes become non-zero
        s.round(compact=True)
        cnt = int(s.q_count.item())
        rows = s.q_rows[:cnt].long()
        want = _masks_from_planes(s.planes[rows])
        assert torch.equal(s.masks[rows], want), (use_history, r)
        checked += cnt
        p, v = stub_net.hash_stub_torch(s.planes, 3)
        s.policy[:cnt].copy_(p[rows])
        s.value[:cnt].copy_(v[rows])
    assert checked > 2000
    if use_history:
        assert int((s.masks.long() >> 14).max()) > 0            # the second plane block was seen
    s.leaf_masks(False)
    assert s.masks is None
    s.close()


@pytest.mark.parametrize("arith", ["c6", "c8", "f16x3"])
def test_network_with_handed_in_masks_is_identical(arith):
    import torch
    from cchess_alphazero.agent.model import calibration_planes, guarded_inference_net
    from test_gpu_guard import peaked_net
    for depth in (14, 28):
        net = peaked_net(20.0, blocks=3)
        if depth == 28:
            from cchess_alphazero.agent.model import CChessNet
            torch.manual_seed(3)
            net = CChessNet(cnn_filter_num=128, res_layer_num=3, input_depth=28).eval()
        planes = calibration_planes(700, depth, seed=17)
        masks = _masks_from_planes(planes).contiguous()
        g = guarded_inference_net(net, torch.float32, trunk="mfma", arith=arith, guard=False, planes=planes[:256])
        p0, v0 = (t.clone() for t in g(planes))
        p1, v1 = g(planes, masks=masks)
        assert torch.equal(p0, p1) and torch.equal(v0, v1), (arith, depth)
        # the planes are not read at all then: poison them
        p2, v2 = g(torch.full_like(planes, 255), masks=masks)
        assert torch.equal(p0, p2) and torch.equal(v0, v2), (arith, depth)
        # compact queue: rows index planes and masks alike
        rows = torch.randperm(700, device="cuda")[:300].int()
        count = torch.tensor([211], dtype=torch.int32, device="cuda")
        pc, vc = g(planes, rows=rows, count=count, masks=masks)
        sel = rows[:211].long()
        assert torch.equal(pc[:211], p0[sel]) and torch.equal(vc[:211], v0[sel]), (arith, depth)


def test_masks_reach_the_fused_input_layer_with_other_head_widths():
    """a network whose heads do not have 4 + 2 filters (CChessNet(policy_filters=..., value_filters=...), reachable
    through lib/keras_io.py) runs the generic head path -- takes_masks() is still True for it, the engine switches the planes
    off, so that path must hand the occupancy boards to the fused input layer too: same outputs from the planes, from the
    boards with the planes poisoned, and within 1e-4 of the fp32 module."""
    import torch
    from cchess_alphazero.agent.model import CChessNet, InferenceNet, calibration_planes
    torch.manual_seed(5)
    net = CChessNet(cnn_filter_num=128, res_layer_num=3, policy_filters=2, value_filters=1).eval()
    planes = calibration_planes(300, 14, seed=19)
    masks = _masks_from_planes(planes).contiguous()
    g = InferenceNet(net, torch.float32, trunk="mfma", arith="f16x3").cuda()
    assert g.head_w32.shape[0] == 3 and g.takes_masks()
    p0, v0 = (t.clone() for t in g(planes))
    p1, v1 = g(torch.full_like(planes, 255), masks=masks)
    assert torch.equal(p0, p1) and torch.equal(v0, v1)
    with torch.no_grad():
        pr, vr = net(planes.float().cpu())
    assert (p0.cpu() - pr).abs().max().item() < 1e-4 and (v0.cpu() - vr).abs().max().item() < 1e-4


@pytest.mark.parametrize("use_history", [False, True])
def test_leaves_as_occupancy_boards_only(use_history):
    """cz_search_leaf_planes(0): the kernel writes the boards and leaves the planes alone; queue_planes() rebuilds the same
    planes; the search itself (driven by the same network outputs) is unchanged edge for edge."""
    import types
    import torch
    import stub_net
    from cchess_alphazero import _native, _native_search
    pc = types.SimpleNamespace(simulation_num_per_move=48, search_threads=6, c_puct=1.5, noise_eps=0.0, dirichlet_alpha=0.2,
                               tau_decay_rate=0.0, virtual_loss=3, resign_threshold=-0.92, min_resign_turn=20,
                               max_game_length=40, enable_resign_rate=1.0)

    def run(planes_on):
        s = _native_search.Search(pc, 32, seed=9, planes_dtype=_native.U8, use_history=use_history)
        with pytest.raises(_native.NativeError):
            s.leaf_planes(False)                               # not without the boards
        s.leaf_masks(True)
        s.leaf_planes(planes_on)
        s.planes.fill_(77)                                     # a sentinel no encoder writes
        s.start_selfplay(seed=9)
        seen = []
        for r in range(30):
            s.round(compact=True)
            cnt = int(s.q_count.item())
            rows = s.q_rows[:cnt].long()
            qp = s.queue_planes()
            if planes_on:
                assert torch.equal(qp[rows], s.planes[rows])
            else:
                assert int((s.planes != 77).sum()) == 0, r     # untouched
                assert torch.equal(_masks_from_planes(qp[rows]), s.masks[rows])
            seen.append((rows.clone(), s.masks[rows].clone()))
            p, v = stub_net.hash_stub_torch(qp, 3)
            s.policy[:cnt].copy_(p[rows])
            s.value[:cnt].copy_(v[rows])
        st = s.root_stats()
        c = s.counters()
        s.leaf_masks(False)
        assert s.planes_off is False
        s.close()
        return seen, st, c

    a, sa, ca = run(True)
    b, sb, cb = run(False)
    assert len(a) == len(b)
    for (ra, ma), (rb, mb) in zip(a, b):
        # (the compact queue's row order is arbitrary between launches: compare by slot)
        ia, ib = ra.argsort(), rb.argsort()
        assert torch.equal(ra[ia], rb[ib]) and torch.equal(ma[ia], mb[ib])
    for k in ("n", "w", "counts"):
        assert (sa[k] == sb[k]).all(), k
    assert ca["expansions"] == cb["expansions"] and ca["sims"] == cb["sims"] and ca["plies"] == cb["plies"]


def test_engine_switches_the_planes_off_and_plays_the_same_games(monkeypatch):
    """SelfPlayEngine: with the input layer fused into the first block the engine asks for boards only (Search.planes_off);
    CZ_LEAF_PLANES=1 keeps the planes.  Same seeds, same network -> the same visit counts either way; the audit positions
    (queue_planes) are real positions in both."""
    import torch as t
    from cchess_alphazero.config import Config
    from cchess_alphazero.engine import SelfPlayEngine

    def run(keep):
        monkeypatch.setenv("CZ_LEAF_PLANES", "1" if keep else "0")
        cfg = Config("mini")
        cfg.model.cnn_filter_num, cfg.model.res_layer_num = 128, 2
        cfg.play.simulation_num_per_move, cfg.play.search_threads, cfg.play.noise_eps = 40, 4, 0.0
        eng = SelfPlayEngine(cfg, 64, dtype=t.float32, seed=11)
        assert eng.search.masks is not None and eng.net.takes_masks()
        assert eng.search.planes_off == (not keep)
        eng.start()
        for _ in range(12):
            eng.step()
        st = eng.search.root_stats()
        qp = eng.queue_planes(64)
        assert qp.shape == (64, 14, 10, 9) and int(qp.sum()) > 64 * 10      # positions, not an empty queue
        au = eng.audit_network(32)
        assert au is None or au["ok"]
        return st, qp

    (a, qa), (b, qb) = run(True), run(False)
    for k in ("n", "w", "counts"):
        assert (a[k] == b[k]).all(), k
    assert t.equal(qa, qb)
