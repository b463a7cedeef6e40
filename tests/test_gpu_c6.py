"""-m gpu: the c6 tower arithmetic (k_resblock_c8<.., C6>, csrc/xq_conv.hip; K loop csrc/xq_c8_kloop.h FMT = 1): the c8
sum with bf6 (e3m2) correction operands, which the matrix pipe retires in half the time of e4m3 ones.

Pinned three ways, all against float64:
  * its OPERAND MODEL: the tower's policy against a CPU emulation that rounds every operand as the kernels do
    (tools/emulate_fp8_corrections.py mode "c6-kernel": float64 products) -- the kernels' error against float64 must be
    of the size the model predicts (measured: 0.7x; the c8 kernels against the c8 model: the same) and an order of
    magnitude below fp16 alone, which is what a wrong element order, piece address or scale would give (the correction
    terms turn into noise);
  * batch independence: a position's result does not depend on which workgroup, which pipeline slot (first board of a
    workgroup, later boards with the deferred epilogue in flight, the last one) or which queue row it is computed in;
  * the tolerance north_star states (policy / value within 1e-4 of the reference network) through the load-time guard
    (GUARD_TOL 5e-5 on calibration positions, 1e-4 with margin on fresh ones), chain c6 -> c8 -> c8>N -> f16x3 -> bf16x3.
"""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from test_gpu_guard import peaked_net  # noqa: E402


def centred_logits(p):
    import torch
    lg = torch.log(p.double().clamp_min(1e-300))
    return lg - lg.mean(1, keepdim=True)


@pytest.mark.parametrize("blocks,filters", [(2, 128), (3, 128), (2, 192), (3, 192)])
def test_c6_tower_matches_its_operand_model(blocks, filters):
    """128 filters, 2 blocks: fused-input block + heads block; 3: + a plain block in between (all three kernel variants).
    192 filters (round 6; the reference's deployed width, configs/distribute.py:84-87): k_resblock_ip_c8<192, XF, YF> -- block 0
    reads the input layer's c8 image (<0, 1>), the others are c6 blocks (<1, 1>), the last one writes fp32."""
    import torch
    import emulate_fp8_corrections as emu
    from cchess_alphazero.agent.model import calibration_planes, guarded_inference_net, reference_forward_f64
    net = peaked_net(20.0, blocks=blocks, filters=filters)
    planes = calibration_planes(40, 14, seed=5)
    g = guarded_inference_net(net, torch.float32, trunk="mfma", arith="c6", guard=False, planes=planes)
    assert g.arith_name == "c6" and g.c6 and g.arith_effective == "c6"
    p, v = g(planes)
    ref = reference_forward_f64(net, planes)
    x, lg, pe, ve, _ = emu.run(net, planes.cpu(), "c6-kernel", exps=g.act_exps)
    d_model = (centred_logits(pe) - centred_logits(ref[0].cpu())).abs().max().item()       # the arithmetic's own error
    d_kernel = (centred_logits(p.cpu()) - centred_logits(pe)).abs().max().item()            # kernels vs their model
    d_total = (centred_logits(p.cpu()) - centred_logits(ref[0].cpu())).abs().max().item()
    print(f"c6, {blocks} blocks: model vs f64 {d_model:.2e}, kernels vs model {d_kernel:.2e}, kernels vs f64 {d_total:.2e}; "
          f"exponents {g.act_exps}")
    assert d_total < 1.5 * d_model + 2e-6 and d_kernel < 1.5 * d_model + 2e-6, (d_total, d_kernel, d_model)
    dv_model = (ve - ref[1].cpu()).abs().max().item()
    dv_total = (v.cpu().double() - ref[1].cpu()).abs().max().item()
    assert dv_total < 3.0 * dv_model + 2e-6, (dv_total, dv_model)       # (one stretched scalar per position: noisier than the logits)
    # fp16 alone (no corrections) on the same positions: what a broken correction path would look like
    xf, lgf, pf, vf, _ = emu.run(net, planes.cpu(), "f16")
    d_f16 = (centred_logits(pf) - centred_logits(ref[0].cpu())).abs().max().item()
    assert d_f16 > 8.0 * d_total, (d_f16, d_total)


@pytest.mark.parametrize("filters", [128, 192])
def test_c6_results_do_not_depend_on_the_batch(filters):
    import torch
    from cchess_alphazero.agent.model import calibration_planes, guarded_inference_net
    net = peaked_net(20.0, blocks=3, filters=filters)
    planes = calibration_planes(1500, 14, seed=9)                # > 5 boards per workgroup on 256 CUs
    g = guarded_inference_net(net, torch.float32, trunk="mfma", arith="c6", guard=False, planes=planes[:256])
    p_all, v_all = (t.clone() for t in g(planes))
    for lo, n in ((0, 1), (0, 40), (700, 300), (1499, 1), (3, 1497)):
        p, v = g(planes[lo:lo + n].contiguous())
        assert torch.equal(p, p_all[lo:lo + n]) and torch.equal(v, v_all[lo:lo + n]), (lo, n)
    # the compact queue: rows gathered by index, count on the device
    rows = torch.randperm(1500, device="cuda")[:900].int()
    count = torch.tensor([777], dtype=torch.int32, device="cuda")
    p, v = g(planes, rows=rows, count=count)
    sel = rows[:777].long()
    assert torch.equal(p[:777], p_all[sel]) and torch.equal(v[:777], v_all[sel])


def test_guard_keeps_c6_on_the_benchmark_network_and_leaves_it_where_it_must():
    import torch
    from cchess_alphazero.agent.model import (CChessNet, calibration_planes, guarded_inference_net,
                                              measure_against_reference, reference_forward_f64)
    torch.manual_seed(0)
    net = CChessNet(cnn_filter_num=128, res_layer_num=7).eval()
    g = guarded_inference_net(net, torch.float32, trunk="mfma", arith="c6")
    c = g.calibration["candidates"]
    print("benchmark network:", c, g.calibration["c6_exponents"])
    assert g.arith_effective == "c6" and len(c) == 1 and c[0]["policy_max_abs"] < 5e-5 and c[0]["value_max_abs"] < 5e-5
    fresh = calibration_planes(192, 14, seed=777)
    ref = reference_forward_f64(net, fresh)
    m6 = measure_against_reference(g, ref, fresh)
    m8 = measure_against_reference(guarded_inference_net(net, torch.float32, trunk="mfma", arith="c8"), ref, fresh)
    print(f"fresh positions: c6 logit {m6['logit_max_abs']:.2e} policy {m6['policy_max_abs']:.2e} value {m6['value_max_abs']:.2e}; "
          f"c8 logit {m8['logit_max_abs']:.2e} policy {m8['policy_max_abs']:.2e} value {m8['value_max_abs']:.2e}")
    assert m6["policy_max_abs"] < 5e-5 and m6["value_max_abs"] < 5e-5 and m6["logit_max_abs"] < 6.0 * m8["logit_max_abs"] + 1e-6
    # a peaked policy: whatever the chain ends on is inside the tolerance, and c6 is its first candidate
    for scale in (60.0, 150.0):
        sharp = peaked_net(scale)
        gs = guarded_inference_net(sharp, torch.float32, trunk="mfma", arith="c6")
        refs = reference_forward_f64(sharp, fresh)
        ms = measure_against_reference(gs, refs, fresh)
        names = [r["arith"] for r in gs.calibration["candidates"]]
        print(f"policy x{scale:g}: c6 -> {gs.arith_effective} via {names}: policy {ms['policy_max_abs']:.2e} value {ms['value_max_abs']:.2e}")
        assert names[0] == "c6" and gs.arith_requested == "c6"
        assert ms["policy_max_abs"] < 1e-4 and ms["value_max_abs"] < 1e-4
    # 192 filters have c6 since round 6 (the benchmark's random-init weights keep it there too) ...
    g192 = guarded_inference_net(CChessNet(cnn_filter_num=192, res_layer_num=2).eval(), torch.float32, trunk="mfma", arith="c6")
    assert g192.arith_effective == "c6" and g192.c6, g192.calibration["candidates"]
    # ... shapes without a c6 kernel degrade to the c8 family
    g256 = guarded_inference_net(CChessNet(cnn_filter_num=256, res_layer_num=2).eval(), torch.float32, trunk="mfma", arith="c6")
    assert g256.arith_effective in ("f16x3", "bf16x3"), g256.arith_effective
    g1 = guarded_inference_net(CChessNet(cnn_filter_num=128, res_layer_num=1).eval(), torch.float32, trunk="mfma", arith="c6")
    assert g1.arith_effective == "c8"


def test_c6_saturation_is_graceful():
    """Activations beyond the calibration range saturate the bf6 images (the conversion clamps at 28 * 2^k): only the
    correction terms of those elements are lost, the result degrades towards fp16's, nothing overflows."""
    import torch
    from cchess_alphazero.agent.model import InferenceNet, calibration_planes, measure_against_reference, reference_forward_f64
    net = peaked_net(1.0, blocks=3)
    planes = calibration_planes(64, 14, seed=3)
    ref = reference_forward_f64(net, planes, with_activations=True)
    from cchess_alphazero.agent.model import c6_exponents
    kmid, kout = c6_exponents(ref[3], headroom=0)
    tight = InferenceNet(net, torch.float32, trunk="mfma", arith="c6", act_exps=(kmid, kout)).cuda()
    low = InferenceNet(net, torch.float32, trunk="mfma", arith="c6", act_exps=([k - 3 for k in kmid], [k - 3 for k in kout])).cuda()
    m_t = measure_against_reference(tight, ref, planes)
    m_l = measure_against_reference(low, ref, planes)
    print("exponents", kmid, kout, "fitted:", m_t, " 3 too small:", m_l)
    assert m_t["finite"] and m_l["finite"]
    assert m_t["logit_max_abs"] < 1e-4 and m_l["logit_max_abs"] < 5e-3


@pytest.mark.parametrize("blocks", [4, 7])
def test_chained_c6_tower_is_bit_identical_to_block_by_block(blocks):
    """cz_tower_c6 (k_tower_c6, round 5): the consecutive c6 inner blocks in ONE launch, activations staying in LDS (a workgroup
    takes a pair of boards through the chain; results staged inside the dead image of the other slot and converted in place).
    Same arithmetic, accumulation order and conversions as one k_resblock_c8<.., C6> launch per block: the network's outputs are
    IDENTICAL -- for batch sizes that give the workgroups one board (the odd-count path: the board runs in both slots), two,
    three, and many; through the compact queue as well."""
    import torch
    from cchess_alphazero.agent.model import calibration_planes, guarded_inference_net
    net = peaked_net(20.0, blocks=blocks)
    planes_all = calibration_planes(1100, 14, seed=23)
    g = guarded_inference_net(net, torch.float32, trunk="mfma", arith="c6", guard=False, planes=planes_all[:256])
    assert g.c6 and g.c6_blocks == blocks
    g.chain_heads = False                     # (the chain proper: the heads-as-exit variant reorders the head sums, tested below)
    for n in (1, 37, 256, 300, 700, 1100):
        planes = planes_all[:n].contiguous()
        g.chain_blocks = False
        p0, v0 = (t.clone() for t in g(planes))
        g.chain_blocks = True
        g.block_events = []
        p1, v1 = g(planes)
        launches = [len(e) > 2 and e[2] or 1 for e in g.block_events]
        g.block_events = None
        assert launches == [1, blocks - 2, 1], launches               # FIRST | the chain | HEADS
        assert torch.equal(p0, p1) and torch.equal(v0, v1), (blocks, n, (p0 - p1).abs().max().item())
    # compact queue: rows / count on the device
    planes = planes_all[:900].contiguous()
    rows = torch.randperm(900, device="cuda")[:640].int()
    count = torch.tensor([517], dtype=torch.int32, device="cuda")
    g.chain_blocks = False
    p0, v0 = (t.clone() for t in g(planes, rows=rows, count=count))
    g.chain_blocks = True
    p1, v1 = g(planes, rows=rows, count=count)
    assert torch.equal(p0[:517], p1[:517]) and torch.equal(v0[:517], v1[:517])


@pytest.mark.parametrize("blocks", [3, 7])
def test_chain_through_the_last_block_with_the_heads_as_its_exit(blocks):
    """cz_tower_c6_heads (the default where the whole tower is c6; CZ_TOWER_HEADS=0 switches it off): the chain ends on the tower's last block and the 1 x 1 head convolutions are
    its exit pass.  The head dot products are summed in a different order than the unchained launch's (four 32-channel partial
    sums per pixel instead of sixteen 8-channel ones), everything else is identical: policy / value agree to float32 rounding."""
    import torch
    from cchess_alphazero.agent.model import calibration_planes, guarded_inference_net
    net = peaked_net(20.0, blocks=blocks)
    planes_all = calibration_planes(700, 14, seed=29)
    g = guarded_inference_net(net, torch.float32, trunk="mfma", arith="c6", guard=False, planes=planes_all[:256])
    for n in (1, 37, 300, 700):
        planes = planes_all[:n].contiguous()
        g.chain_heads = False
        p0, v0 = (t.clone() for t in g(planes))
        g.chain_heads = True
        g.block_events = []
        p1, v1 = g(planes)
        launches = [len(e) > 2 and e[2] or 1 for e in g.block_events]
        g.block_events = None
        assert launches == [1, blocks - 1], launches                  # FIRST | the chain through the last block
        # (128-term fp32 sums in two association orders, through the dense heads of a hostile test network: policy a few
        #  1e-7, value up to 1e-5 -- the float32 noise either order has against float64; north_star's tolerance is 1e-4)
        assert torch.isfinite(p1).all() and (p0 - p1).abs().max().item() < 5e-6 and (v0 - v1).abs().max().item() < 3e-5, \
            (blocks, n, (p0 - p1).abs().max().item(), (v0 - v1).abs().max().item())
    g.chain_heads = False
