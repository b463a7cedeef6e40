"""-m gpu: the residual tower as CHAINS of blocks for every arithmetic (round 6: cz_tower, cz_tower_pairs; csrc/xq_tower.hip).

Reference: CChessModel.build's loop `for _ in range(res_layer_num): x = self._build_residual_block(x)` (agent/model.py:41-43,
blocks :68-83).  One launch per block makes every block read its input from HBM and write its output back; a chain keeps a pair
of boards in LDS through all its blocks.  What a chain must be: the SAME arithmetic in the same order as the one-block kernels
-- so the network's outputs are compared for EQUALITY with the block-by-block launches (CZ_TOWER_CHAIN=0 / chain_blocks False),
for every arithmetic the load-time guard can choose: c6, c6>N (c6 blocks, a hand-over, c8 blocks -- one launch), c8, c8>N (c8
blocks whose exit writes fp16 pairs, then the f16x3 blocks), f16x3, bf16x3; batch sizes that give the workgroups one board (the
odd-count path: the board runs in both slots), two, three and many; the compact queue.  Where the head convolutions are the last
chain's exit the head sums associate differently (four 32-channel partial sums per pixel): float32 rounding, bounded here."""
import pytest

from test_gpu_guard import peaked_net

pytestmark = pytest.mark.gpu

# (the guard walks the hybrids one block at a time since round 6: every split can be what a network runs on)
ARITHS = ["c6", "c6>6", "c6>5", "c6>4", "c6>3", "c6>2", "c6>1", "c8", "c8>6", "c8>5", "c8>4", "c8>3", "c8>2", "f16x3", "bf16x3"]


def _net(arith, blocks):
    import torch
    from cchess_alphazero.agent.model import calibration_planes, guarded_inference_net
    net = peaked_net(20.0, blocks=blocks)
    planes_all = calibration_planes(1100, 14, seed=23)
    g = guarded_inference_net(net, torch.float32, trunk="mfma", arith=arith, guard=False, planes=planes_all[:256])
    assert g.arith_name == arith, (g.arith_name, arith)
    return g, planes_all


def _launches(g, planes, **kw):
    g.block_events = []
    out = tuple(t.clone() for t in g(planes, **kw))
    launches = [e[2] if len(e) > 2 else 1 for e in g.block_events]
    g.block_events = None
    return out, launches


@pytest.mark.parametrize("tower4", ["1", "0"])          # cz_tower on k_resblock_ip4_c8<128> (default) / on k_tower (CZ_TOWER4=0)
@pytest.mark.parametrize("arith", ARITHS)
def test_chained_tower_is_bit_identical_to_block_by_block(arith, tower4, monkeypatch):
    monkeypatch.setenv("CZ_TOWER4", tower4)
    import torch
    from cchess_alphazero.agent.model import tower_plan
    blocks = 7
    g, planes_all = _net(arith, blocks)
    g.chain_heads = False                     # (the chains proper: the heads-as-exit variants reorder the head sums, tested below)
    want = [len(st[1]) if st[0] in ("tower", "pairs") else 1 for st in tower_plan(g.block_kinds(), chain_heads=False)]
    for n in (1, 37, 256, 300, 700, 1100):
        planes = planes_all[:n].contiguous()
        g.chain_blocks = False
        (p0, v0), l0 = _launches(g, planes)
        assert l0 == [1] * blocks, l0
        g.chain_blocks = True
        (p1, v1), l1 = _launches(g, planes)
        assert l1 == want, (l1, want)
        assert torch.isfinite(p1).all() and torch.isfinite(v1).all()
        assert torch.equal(p0, p1) and torch.equal(v0, v1), (arith, n, (p0 - p1).abs().max().item(), (v0 - v1).abs().max().item())
    # compact queue: rows / count on the device
    planes = planes_all[:900].contiguous()
    rows = torch.randperm(900, device="cuda")[:640].int()
    count = torch.tensor([517], dtype=torch.int32, device="cuda")
    g.chain_blocks = False
    p0, v0 = (t.clone() for t in g(planes, rows=rows, count=count))
    g.chain_blocks = True
    p1, v1 = g(planes, rows=rows, count=count)
    assert torch.equal(p0[:517], p1[:517]) and torch.equal(v0[:517], v1[:517])


@pytest.mark.parametrize("arith,blocks", [("c8", 4), ("c8>2", 4), ("c6>2", 4), ("f16x3", 3), ("c8>2", 3), ("c6", 2), ("f16x3", 2)])
def test_short_towers_chain_too(arith, blocks):
    """Chains of one and two blocks, hand-overs right behind the first block, a chain that is only the last block."""
    import torch
    g, planes_all = _net(arith, blocks)
    for heads_in_chain in (False, True):
        g.chain_heads = heads_in_chain
        for n in (1, 3, 300, 513):
            planes = planes_all[:n].contiguous()
            g.chain_blocks = False
            p0, v0 = (t.clone() for t in g(planes))
            g.chain_blocks = True
            p1, v1 = g(planes)
            if heads_in_chain:
                assert (p0 - p1).abs().max().item() < 5e-6 and (v0 - v1).abs().max().item() < 3e-5, (arith, blocks, n)
            else:
                assert torch.equal(p0, p1) and torch.equal(v0, v1), (arith, blocks, n)


@pytest.mark.parametrize("tower4", ["1", "0"])
@pytest.mark.parametrize("arith", ["c6>5", "c8", "c8>3", "f16x3"])
def test_heads_as_the_last_chains_exit(arith, tower4, monkeypatch):
    """The default: the tower's last block is inside the last chain and the 1 x 1 head convolutions are its exit pass.  The head
    dot products are summed over four 32-channel partial sums per pixel (the one-block HEADS kernels: sixteen 8-channel ones);
    the pair chains take the block's value as hi + lo of its operand pair (k_resblock<HEADS> keeps the fp32 value: an fp16 pair
    stands for it to 2^-22; bf16 pairs, 2^-17, keep their HEADS launch -- last test).  Everything else is identical: policy /
    value agree to float32 rounding."""
    import torch
    from cchess_alphazero.agent.model import tower_plan
    monkeypatch.setenv("CZ_TOWER4", tower4)
    g, planes_all = _net(arith, 7)
    want = [len(st[1]) if st[0] in ("tower", "pairs") else 1 for st in tower_plan(g.block_kinds())]
    for n in (1, 37, 300, 700):
        planes = planes_all[:n].contiguous()
        g.chain_heads = False
        p0, v0 = (t.clone() for t in g(planes))
        g.chain_heads = True
        (p1, v1), l1 = _launches(g, planes)
        assert l1 == want and sum(l1) == 7, (l1, want)
        assert torch.isfinite(p1).all() and (p0 - p1).abs().max().item() < 5e-6 and (v0 - v1).abs().max().item() < 3e-5, \
            (arith, n, (p0 - p1).abs().max().item(), (v0 - v1).abs().max().item())


def test_the_guards_choice_for_a_peaked_policy_runs_as_three_launches():
    """What VERDICT r05 item 1 asks for: the arithmetic the load-time guard gives a peaked-policy network (c8>N: c8 blocks, then
    f16x3 blocks) as FIRST | the c8 blocks, exit = fp16 pairs | the f16x3 blocks with the heads -- with the guard untouched and
    the outputs within north_star's tolerance of the float64 network."""
    import torch
    from cchess_alphazero.agent.model import (GUARD_TOL, LOGIT_TOL, calibration_planes, guarded_inference_net,
                                              measure_against_reference, reference_forward_f64, within_guard)
    assert GUARD_TOL == 5e-5 and LOGIT_TOL == 2e-4
    from cchess_alphazero.agent.model import CChessNet
    torch.manual_seed(0)                                         # bench.py's stand-in for a trained network (sharpened_copy): the
    net = CChessNet(cnn_filter_num=128, res_layer_num=7).eval()  # benchmark's random-init weights, policy layer x 240 (max p 0.86)
    net.policy_out.weight.data.mul_(240.0)
    g = guarded_inference_net(net, torch.float32, trunk="mfma", arith="c6")
    name = g.arith_effective
    assert name.startswith("c8>"), (name, g.calibration["candidates"])
    planes = calibration_planes(512, 14, seed=77)
    (p, v), launches = _launches(g, planes)
    n8 = int(name[3:])
    assert launches == [1, n8 - 1, 7 - n8], (name, launches)
    m = measure_against_reference(g, reference_forward_f64(net, planes), planes)
    assert within_guard(m, tol=1e-4, logit_tol=LOGIT_TOL * 1.5), m           # (fresh positions, not the calibration set)


@pytest.mark.parametrize("arith", ["c6", "c8", "c6>3", "c8>3"])
def test_both_chain_kernels_give_the_same_bits(arith, monkeypatch):
    """cz_tower on k_tower (CZ_TOWER4=0) and on the four-wave pair kernel k_resblock_ip4_c8<128> (round 6, default): same
    products in the same order per accumulator tile, the exits (operand image, fp16 pairs, head features) computed item for item
    the same way -- identical network outputs, heads exit included."""
    import torch
    g, planes_all = _net(arith, 7)
    for n in (1, 37, 300, 1100):
        planes = planes_all[:n].contiguous()
        monkeypatch.setenv("CZ_TOWER4", "0")
        p0, v0 = (t.clone() for t in g(planes))
        monkeypatch.setenv("CZ_TOWER4", "1")
        p1, v1 = g(planes)
        assert torch.equal(p0, p1) and torch.equal(v0, v1), (arith, n, (p0 - p1).abs().max().item())


def test_bf16_pairs_keep_their_heads_launch():
    import torch
    g, planes_all = _net("bf16x3", 7)
    (p, v), launches = _launches(g, planes_all[:300].contiguous())
    assert launches == [1, 5, 1], launches                        # FIRST | the pair chain | HEADS (fp32 value)
    g.chain_blocks = False
    p0, v0 = g(planes_all[:300].contiguous())
    assert torch.equal(p0, p) and torch.equal(v0, v)


@pytest.mark.parametrize("arith,blocks", [("c8", 10), ("c6", 10), ("c6>3", 10), ("c8>6", 10), ("c6", 2), ("c8", 2), ("c6>1", 4), ("c8>2", 4),
                                          ("f16x3", 10), ("bf16x3", 4), ("f16x3", 3), ("c8>2", 10)])
def test_192_filter_tower_chains_are_bit_identical(arith, blocks, monkeypatch):
    """cz_resblock_chain (the reference's deployed width, configs/distribute.py:84-87): consecutive 192-filter blocks of one
    arithmetic in one launch -- a PAIR of boards per workgroup, one LDS image per board, four matrix waves of three channel tiles
    (k_resblock_ip4_c8), or CZ_IP_PAIR=0: one board in two images on six matrix waves (k_resblock_ip_c8).  Both equal to one
    launch per block for c8, c6 (block 0 reads the input layer's c8 image on a launch of its own), the c6>N hand-over (a c6 block
    writing a c8 image) and c8>N (fp32 out of the c8 chain, re-split into fp16 pairs)."""
    import torch
    from cchess_alphazero.agent.model import calibration_planes, guarded_inference_net, ip_segments
    net = peaked_net(20.0, blocks=blocks, filters=192)
    planes_all = calibration_planes(700, 14, seed=31)
    g = guarded_inference_net(net, torch.float32, trunk="mfma", arith=arith, guard=False, planes=planes_all[:256])
    assert g.arith_name == arith and g.filters == 192
    want = {pair: [len(b) for _, b in ip_segments(g.block_kinds(), first_alone=pair == "0")] for pair in ("1", "0")}
    for n in (1, 37, 300, 700):
        planes = planes_all[:n].contiguous()
        g.chain_blocks = False
        monkeypatch.setenv("CZ_IP_PAIR", "0")                # (the reference: one launch per block of the six-wave kernel)
        (p0, v0), l0 = _launches(g, planes)
        assert l0 == [1] * blocks, l0
        monkeypatch.setenv("CZ_IP_PAIR", "1")
        (p2, v2), _ = _launches(g, planes)                   # one launch per block of the four-wave kernel
        assert torch.equal(p0, p2) and torch.equal(v0, v2), (arith, blocks, n)
        g.chain_blocks = True
        for pair in ("1", "0"):
            monkeypatch.setenv("CZ_IP_PAIR", pair)
            (p1, v1), l1 = _launches(g, planes)
            assert l1 == want[pair], (l1, want[pair])
            assert torch.isfinite(p1).all() and torch.equal(p0, p1) and torch.equal(v0, v1), \
                (arith, blocks, n, pair, (p0 - p1).abs().max().item())
    rows = torch.randperm(700, device="cuda")[:500].int()
    count = torch.tensor([333], dtype=torch.int32, device="cuda")
    planes = planes_all.contiguous()
    g.chain_blocks = False
    monkeypatch.setenv("CZ_IP_PAIR", "0")
    p0, v0 = (t.clone() for t in g(planes, rows=rows, count=count))
    for chain in (False, True):
        g.chain_blocks = chain
        for pair in ("1", "0"):
            monkeypatch.setenv("CZ_IP_PAIR", pair)
            p1, v1 = g(planes, rows=rows, count=count)
            assert torch.equal(p0[:333], p1[:333]) and torch.equal(v0[:333], v1[:333]), (arith, blocks, chain, pair)


@pytest.mark.parametrize("dtype,blocks", [("float16", 20), ("float16", 3), ("bfloat16", 5), ("float16", 26)])
def test_deep_tower_on_plain_operands_is_one_launch(dtype, blocks, monkeypatch):
    """cz_tower_plain (BASELINE configs[4]: 20 x 256, fp16 MFMA evaluation): all blocks of a 256-filter tower on plain 2-byte
    operands in one launch (24 at most) -- a pair of boards per workgroup with ONE LDS image per board (k_tower_plain2: the
    skip values wait in registers while the intermediate activation overwrites them), or CZ_TOWER_PLAIN_PAIR=0: one board in
    two images (k_tower_plain).  Both equal to one k_resblock launch per block."""
    import torch
    from cchess_alphazero.agent.model import CChessNet, InferenceNet, calibration_planes
    torch.manual_seed(13)
    net = CChessNet(cnn_filter_num=256, res_layer_num=blocks).eval()
    planes_all = calibration_planes(600, 14, seed=41)
    g = InferenceNet(net, getattr(torch, dtype), trunk="mfma").cuda()
    assert g.parts == 1 and g.filters == 256
    for n in (1, 37, 300, 600):
        planes = planes_all[:n].contiguous()
        g.chain_blocks = False
        (p0, v0), l0 = _launches(g, planes)
        assert l0 == [1] * blocks, l0
        g.chain_blocks = True
        for pair in ("1", "0"):
            monkeypatch.setenv("CZ_TOWER_PLAIN_PAIR", pair)
            (p1, v1), l1 = _launches(g, planes)
            assert l1 == ([24, blocks - 24] if blocks > 24 else [blocks]), l1
            assert torch.isfinite(p1).all() and torch.equal(p0, p1) and torch.equal(v0, v1), \
                (dtype, blocks, n, pair, (p0 - p1).abs().max().item())
    # the board count on the device (cz_tower_plain's n_dev; an odd count: the last pair is half empty, rows beyond it untouched)
    from cchess_alphazero import _native
    bl = [g._block_params(i) for i in range(min(blocks, 3))]
    x = (torch.randn((300, 90, 256), device="cuda") * 0.5).to(getattr(torch, dtype))
    want = _native.tower_plain(x[:189].contiguous(), bl, torch.empty_like(x[:189]))
    count = torch.tensor([189], dtype=torch.int32, device="cuda")
    for pair in ("1", "0"):
        monkeypatch.setenv("CZ_TOWER_PLAIN_PAIR", pair)
        y = torch.full_like(x, 7.0)
        _native.tower_plain(x, bl, y, count=count)
        assert torch.equal(y[:189], want) and bool((y[189:] == 7.0).all()), (dtype, pair)


def test_tower_entry_points_reject_what_they_cannot_run():
    import torch
    from cchess_alphazero import _native
    g, planes_all = _net("c8", 3)
    x = (torch.zeros((4, 90, 128), dtype=torch.float16, device="cuda"), torch.zeros((4, 90, 256), dtype=torch.uint8, device="cuda"))
    blk = [g._block_params(1)]
    with pytest.raises(_native.NativeError):
        _native.tower(x, blk * 9, _native.IMG_C8, out=x, fmt_x=[0] * 9, fmt_y=[0] * 9)      # more than 8 blocks
    with pytest.raises(_native.NativeError):
        _native.tower(x, blk, _native.IMG_C8, out=x, fmt_x=[2], fmt_y=[0])                   # pair blocks belong to cz_tower_pairs
    with pytest.raises(_native.NativeError):
        _native.tower(x, blk * 2, _native.IMG_C8, out=x, fmt_x=[1, 0], fmt_y=[1, 0])         # one arithmetic per launch
    with pytest.raises(_native.NativeError):
        _native.tower(x, blk, _native.IMG_C6, out=x, fmt_x=[0], fmt_y=[0])                   # a c8 chain cannot write a c6 image
    with pytest.raises(_native.NativeError):
        _native.tower(x, blk, 7, out=x, fmt_x=[0], fmt_y=[0])                                # no such exit
    with pytest.raises(_native.NativeError):
        _native.tower(x, blk, _native.EXIT_HEADS, heads=(g.head_w32[:5], g.head_b32, 4, x[0], x[0]), fmt_x=[0], fmt_y=[0])
