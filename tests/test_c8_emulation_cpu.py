"""The numerical design margin of the c8 and c6 tower arithmetics, on the CPU (tools/emulate_fp8_corrections.py): the whole
7 x 128 policy / value network with every tower product formed from operands rounded exactly as the matrix instructions
see them (fp16 main term, e4m3 correction operands, one power-of-two scale per tensor), float64 products -- against the
float64 network.  The GPU tests measure the kernels; this one pins the arithmetic itself, wherever the suite runs:
the c8 form must stay well inside the north_star tolerance (policy / value within 1e-4) on a network with a sharpened
policy, and the fp16 main term alone must NOT (which is why the correction terms exist)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "chinesechess-alphazero_amd"))
sys.path.insert(0, ROOT)


def test_c8_arithmetic_stays_inside_the_tolerance_where_fp16_alone_does_not():
    import torch
    import emulate_fp8_corrections as emu
    from cchess_alphazero.agent.model import CChessNet
    import oracle.xq_oracle as xo
    emu.UNIFORM = True                                     # constant scale operands, as in the kernels
    torch.manual_seed(3)
    net = CChessNet(cnn_filter_num=128, res_layer_num=7)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.3)
            m.running_var.uniform_(0.5, 2.0)
            m.weight.data.normal_(1, 0.2)
            m.bias.data.normal_(0, 0.2)
    net.policy_out.weight.data.mul_(20.0)                  # largest probability ~0.1 instead of 1e-3
    net.eval()
    rng = np.random.default_rng(3)
    boards, state = [], xo.INIT_STATE
    while len(boards) < 4:
        mv = xo.get_legal_moves(state)
        boards.append(xo.state_to_board(state))
        state = xo.step(state, mv[rng.integers(len(mv))])
    planes = torch.from_numpy(np.stack([xo.planes_board(b) for b in boards]))
    ref = emu.run(net, planes, "f64")
    assert float(ref[2].max()) > 0.03
    err = {}
    for mode in ("bf16x3", "f16+fp8", "c8-kernel", "c6-kernel", "f16"):
        x, lg, p, v, _ = emu.run(net, planes, mode)
        err[mode] = (float((x - ref[0]).norm() / ref[0].norm()), float((p - ref[2]).abs().max()),
                     float((v - ref[3]).abs().max()))
    assert err["f16+fp8"][0] < 4e-5 and err["f16+fp8"][1] < 4e-5 and err["f16+fp8"][2] < 2e-5, err
    # 'c8-kernel': exactly the kernels' operand model (fixed activation scales 2^11 / 1, saturation at 448, one scale per filter)
    assert err["c8-kernel"][0] < 4e-5 and err["c8-kernel"][1] < 4e-5 and err["c8-kernel"][2] < 2e-5, err
    assert err["bf16x3"][0] < 2e-5 and err["bf16x3"][1] < 2e-5, err
    # 'c6-kernel' (round 4): the same sum with bf6 (e3m2) correction operands and per-image exponents, exactly as
    # k_resblock_c8<.., C6> forms it (block 0's first convolution on the fused input layer's c8 image): two mantissa bits
    # instead of three in the corrections -- about twice c8's error, the same class, an order of magnitude below fp16 alone
    assert err["c6-kernel"][0] < 6e-5 and err["c6-kernel"][1] < 6e-5 and err["c6-kernel"][2] < 3e-5, err
    assert err["c6-kernel"][0] < 4 * err["c8-kernel"][0] and err["f16"][0] > 5 * err["c6-kernel"][0], err
    assert err["f16+fp8"][0] < 8 * err["bf16x3"][0], err                     # the same class as the split-bf16 form
    assert err["f16"][1] > 1e-4 and err["f16"][0] > 10 * err["f16+fp8"][0], err   # fp16 alone: outside the tolerance
