"""-m gpu: the hand-written MFMA trunk convolution (csrc/xq_conv.hip) through the C-ABI (cz_conv3x3,
cz_conv3x3_pack_weights, cz_split_bias_act) against a plain PyTorch float64 reference of the same op, and the whole
policy/value network with trunk="mfma" against the plain fp32 PyTorch module.

Tolerances (floating point, stated here as the contract asks):
  * split mode (parts = 2, the fp32 network): the kernel sees operands hi + lo with |x - hi - lo| <= 2^-17 |x| and
    drops the lo*lo product, so each product carries <= ~3 * 2^-17 relative error; accumulation is fp32.
    Against the float64 convolution OF THE SAME (hi + lo) OPERANDS: max |err| <= 2e-5 * max|y|.
  * plain bf16 / fp16: the only difference to the float64 reference on the same rounded operands is fp32
    accumulation and the final rounding of the result to 2 bytes: <= 2^-8 (bf16) / 2^-10 (fp16) relative.
  * whole network, split mode: policy and value within 1e-4 of the fp32 module (north_star tolerance).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _split(t, dtype, parts):
    hi = t.to(dtype)
    return (hi,) if parts == 1 else (hi, (t - hi.float()).to(dtype))


def _reference(x, w, b, skip, relu):
    import torch
    import torch.nn.functional as F
    n, _, c = x.shape
    y = F.conv2d(x.double().view(n, 10, 9, c).permute(0, 3, 1, 2), w.double(), b.double(), padding=1)
    y = y.permute(0, 2, 3, 1).reshape(n, 90, c)
    if skip is not None:
        y = y + skip.double()
    return torch.relu(y) if relu else y


CASES = [(128, "bfloat16", 2), (128, "bfloat16", 1), (128, "float16", 1), (32, "bfloat16", 2), (32, "float16", 1),
         (256, "float16", 1), (256, "bfloat16", 2), (192, "bfloat16", 2), (192, "float16", 1)]


@pytest.mark.parametrize("c,dt,parts", CASES)
@pytest.mark.parametrize("n", [1, 2, 7, 64])
def test_conv3x3_against_float64(c, dt, parts, n):
    import torch
    from cchess_alphazero import _native
    dtype = getattr(torch, dt)
    g = torch.Generator(device="cuda").manual_seed(1000 * c + n)
    x = torch.randn((n, 90, c), device="cuda", generator=g).relu()
    sk = torch.randn((n, 90, c), device="cuda", generator=g)
    w = torch.randn((c, c, 3, 3), device="cuda", generator=g) / (3.0 * c ** 0.5)
    b = torch.randn((c,), device="cuda", generator=g)
    wp = _native.pack_conv3x3_weights(w, dtype, parts).cuda()
    xs, ss = _split(x, dtype, parts), _split(sk, dtype, parts)
    x_eff, s_eff = sum(t.double() for t in xs), sum(t.double() for t in ss)
    w_eff = w if parts == 2 else w.to(dtype).float()
    rel = 2e-5 if parts == 2 else (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -10)
    for skip, relu, f32out in ((None, True, False), (ss, True, False), (ss, False, True), (None, False, False)):
        out = tuple(torch.full((n, 90, c), 7.0, device="cuda", dtype=dtype) for _ in range(parts))
        of = torch.full((n, 90, c), 7.0, device="cuda") if f32out else None
        _native.conv3x3(xs, wp, b, skip=skip, out=None if f32out else out, out_f32=of, relu=relu)
        got = of.double() if f32out else sum(o.double() for o in out)
        ref = _reference(x_eff, w_eff, b, s_eff if skip is not None else None, relu)
        tol = (2e-5 if f32out else rel) * ref.abs().max().item()
        assert (got - ref).abs().max().item() <= tol, (c, dt, parts, n, skip is not None, relu, f32out)


def test_conv3x3_identity_filter_is_a_shift():
    """Asymmetric known answer: a filter that copies input channel (o + 1) % C of the pixel one step up-left
    (ky = 0, kx = 0) must reproduce the shifted board exactly, with zeros where the tap leaves the board."""
    import torch
    from cchess_alphazero import _native
    c, n = 128, 3
    x = torch.arange(n * 90 * c, device="cuda", dtype=torch.float32).reshape(n, 90, c) % 251 - 125.0   # exact in bf16
    w = torch.zeros((c, c, 3, 3), device="cuda")
    for o in range(c):
        w[o, (o + 1) % c, 0, 0] = 1.0
    wp = _native.pack_conv3x3_weights(w, torch.bfloat16, 1).cuda()
    out = (torch.empty((n, 90, c), device="cuda", dtype=torch.bfloat16),)
    _native.conv3x3((x.to(torch.bfloat16),), wp, torch.zeros(c, device="cuda"), out=out, relu=False)
    want = torch.zeros((n, 10, 9, c), device="cuda")
    want[:, 1:, 1:, :] = x.view(n, 10, 9, c)[:, :-1, :-1, :].roll(-1, dims=3)
    assert torch.equal(out[0].float().view(n, 10, 9, c), want)


@pytest.mark.parametrize("c,dt,parts", [(128, "bfloat16", 2), (128, "bfloat16", 1), (128, "float16", 1),
                                        (256, "float16", 1), (256, "bfloat16", 1), (192, "float16", 1),
                                        (192, "bfloat16", 2), (192, "float16", 2)])      # (192 split: k_resblock_ip)
@pytest.mark.parametrize("n", [1, 3, 257, 700])
def test_resblock_equals_two_convolutions(c, dt, parts, n):
    """cz_resblock (one launch, intermediate in LDS) must be BIT-identical to two cz_conv3x3 launches: same
    operands, same MFMA order, same epilogue arithmetic."""
    import torch
    from cchess_alphazero import _native
    dtype = getattr(torch, dt)
    g = torch.Generator(device="cuda").manual_seed(n + c)
    x = torch.randn((n, 90, c), device="cuda", generator=g).relu()
    ws = [torch.randn((c, c, 3, 3), device="cuda", generator=g) / (3.0 * c ** 0.5) for _ in range(2)]
    bs = [torch.randn((c,), device="cuda", generator=g) for _ in range(2)]
    ps = [_native.pack_conv3x3_weights(w, dtype, parts).cuda() for w in ws]
    xs = _split(x, dtype, parts)
    t = tuple(torch.empty_like(xs[0]) for _ in range(parts))
    want = tuple(torch.empty_like(xs[0]) for _ in range(parts))
    _native.conv3x3(xs, ps[0], bs[0], out=t)
    _native.conv3x3(t, ps[1], bs[1], skip=xs, out=want)
    got = tuple(torch.full_like(xs[0], 7.0) for _ in range(parts))
    _native.resblock(xs, ps[0], bs[0], ps[1], bs[1], out=got)
    assert all(torch.equal(a, b) for a, b in zip(got, want))
    if parts == 2:
        want_f = torch.empty((n, 90, c), device="cuda")
        _native.conv3x3(t, ps[1], bs[1], skip=xs, out_f32=want_f)
        got_f = torch.full((n, 90, c), 7.0, device="cuda")
        _native.resblock(xs, ps[0], bs[0], ps[1], bs[1], out_f32=got_f)
        assert torch.equal(got_f, want_f)
    # in place (y aliases x): every workgroup has its boards in LDS before it writes them back
    xi = tuple(a.clone() for a in xs)
    _native.resblock(xi, ps[0], bs[0], ps[1], bs[1], out=xi)
    assert all(torch.equal(a, b) for a, b in zip(xi, want))


@pytest.mark.parametrize("c", [128, 192])
@pytest.mark.parametrize("n", [1, 2, 5, 301])
def test_conv3x3_c8_matches_its_operands_and_the_float64_convolution(n, c):
    """cz_conv3x3_c8 (the c8 tower arithmetic: one fp16 and two scaled-fp8 matrix instructions per 64 input
    channels).  (1) Against float64 arithmetic on EXACTLY the operand values the kernel is given (decoded fragments and
    images): only the instructions' accumulation differs -- bounded by 1e-5 of the sum of |terms| of an output (the fp8
    instruction accumulates to ~6e-5 of ITS terms, which are 2^-12 of the main ones).  (2) Against the float64
    convolution of the unrounded tensors: the arithmetic's own error, which must stay in the class of the split-bf16
    kernel's (both are checked on the same data)."""
    import torch
    import torch.nn.functional as F
    from cchess_alphazero import _native
    from test_c8_pack_cpu import decode_c8_pack
    g = torch.Generator(device="cuda").manual_seed(40 + n)
    x = (torch.randn((n, 90, c), device="cuda", generator=g) * 1.5).relu()
    x[0, 0, :4] = torch.tensor([300.0, 2e-3, 5e-5, 0.0], device="cuda")           # large / tiny activations
    w = torch.randn((c, c, 3, 3), device="cuda", generator=g) / (3.0 * c ** 0.5)
    # output channels of very different magnitude (folded BatchNorm scales): every row has its own correction shifts, which
    # the matrix instruction must take per lane
    w[::3] *= 0.004
    w[1::7] *= 50.0
    bias = torch.randn((c,), device="cuda", generator=g)
    bias[::3] *= 0.004                                     # (the accumulators start at the bias: a bias far above a row's
    bias[1::7] *= 50.0                                     #  products would set the fp32 rounding floor of that row)
    packed = _native.pack_conv3x3_c8_weights(w)
    assert len(set(decode_c8_pack(packed, c)[3].tolist())) >= 3
    x_hi, x_c8 = _native.split_c8(x)
    out = torch.full((n, 90, c), 7.0, device="cuda")
    _native.conv3x3_c8((x_hi, x_c8), packed.cuda(), bias, out_f32=out, relu=False)

    d = torch.float64
    img = lambda t: t.to(d).view(n, 10, 9, c).permute(0, 3, 1, 2)
    conv = lambda a, ww: F.conv2d(a, ww, None, padding=1).permute(0, 2, 3, 1).reshape(n, 90, c)
    w_hi, k0, k1, sh, sl = decode_c8_pack(packed, c)
    tw = lambda a: torch.from_numpy(a).to("cuda", d).view(c, c, 3, 3)
    l8 = x_c8[..., :c].contiguous().view(torch.float8_e4m3fn).to(d)
    h8 = x_c8[..., c:].contiguous().view(torch.float8_e4m3fn).to(d)
    main = conv(img(x_hi), tw(w_hi))
    row = lambda a: torch.from_numpy(2.0 ** (-a.astype(np.float64))).to("cuda", d).view(1, 1, c)       # per output channel
    corr = conv(img(l8), tw(k0)) * row(sh) * 2.0 ** -_native.C8_X_LO_SHIFT + conv(img(h8), tw(k1)) * row(sl)
    want_ops = main + corr + bias.to(d)
    mag = conv(img(x_hi).abs(), tw(w_hi).abs()) + 1e-30
    err_ops = ((out.to(d) - want_ops).abs() / mag).max().item()
    assert err_ops < 1e-5, err_ops

    exact = conv(img(x), w.to(d)) + bias.to(d)
    mag_x = conv(img(x).abs(), w.to(d).abs())
    err_c8 = ((out.to(d) - exact).abs() / mag_x).max().item()
    ps = _native.pack_conv3x3_weights(w, torch.bfloat16, 2).cuda()
    ref = torch.empty((n, 90, c), device="cuda")
    _native.conv3x3(_split(x, torch.bfloat16, 2), ps, bias, out_f32=ref, relu=False)
    err_bf16x3 = ((ref.to(d) - exact).abs() / mag_x).max().item()
    assert err_c8 < 3e-5 and err_c8 < 16 * err_bf16x3 + 1e-6, (err_c8, err_bf16x3)
    # ReLU flag
    out2 = torch.empty_like(out)
    _native.conv3x3_c8((x_hi, x_c8), packed.cuda(), bias, out_f32=out2, relu=True)
    assert torch.equal(out2, out.relu())


@pytest.mark.parametrize("filters,blocks", [(128, 7), (128, 2), (128, 1), (192, 10), (192, 2)])
def test_network_with_c8_tower_matches_fp32_module(filters, blocks):
    """The whole policy / value network with the c8 tower arithmetic (InferenceNet(arith="c8"): fp16 main term +
    two scaled-fp8 correction terms per product; reference architecture agent/model.py:32-83) against the plain PyTorch
    fp32 module on the CPU: the north_star tolerance (policy / value within 1e-4), the logit and relative bounds of the
    split-bf16 test, and agreement with the split-bf16 network itself.  (192, 10) is the reference's deployed topology
    (configs/distribute.py:84-87) on k_resblock_ip_c8."""
    import torch
    from cchess_alphazero.agent.model import CChessNet, InferenceNet
    import oracle.xq_oracle as xo
    torch.manual_seed(11)
    net = CChessNet(cnn_filter_num=filters, res_layer_num=blocks)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.3)
            m.running_var.uniform_(0.5, 2.0)
            m.weight.data.normal_(1, 0.2)
            m.bias.data.normal_(0, 0.2)
    net.policy_out.weight.data.mul_(10.0)                     # a sharper policy than random initialisation gives
    net.eval()
    rng = np.random.default_rng(5)
    boards, state = [], xo.INIT_STATE
    while len(boards) < 40:
        mv = xo.get_legal_moves(state)
        if not mv or xo.done(state)[0]:
            state = xo.INIT_STATE
            continue
        boards.append(xo.state_to_board(state))
        state = xo.step(state, mv[rng.integers(len(mv))])
    x = torch.from_numpy(np.stack([xo.planes_board(b) for b in boards]))
    with torch.no_grad():
        p_ref, v_ref = net.double()(x.double())
    net.float()
    inf = InferenceNet(net, torch.float32, trunk="mfma", arith="c8").cuda()
    assert inf.arith == "c8"
    for planes in (x.cuda(), x.to(torch.uint8).cuda()):
        p, v = inf(planes)
        p, v = p.cpu().double(), v.cpu().double()
        assert (p - p_ref).abs().max().item() < 1e-4 and (v - v_ref).abs().max().item() < 1e-4
        lg, lr = torch.log(p.clamp_min(1e-300)), torch.log(p_ref.clamp_min(1e-300))
        assert ((lg - lg.mean(1, keepdim=True)) - (lr - lr.mean(1, keepdim=True))).abs().max().item() < 1e-3
        assert ((p - p_ref).abs() / p_ref.clamp_min(1e-12)).max().item() < 1e-3
    ref = InferenceNet(net, torch.float32, trunk="mfma").cuda()
    assert ref.arith == "bf16x3"
    p2, v2 = ref(x.cuda())
    assert (p2.cpu().double() - p).abs().max().item() < 1e-4 and (v2.cpu().double() - v).abs().max().item() < 1e-4
    inf.fused_blocks = False                                  # per-convolution launches: the same tower arithmetic (the head
    p3, v3 = inf(x.to(torch.uint8).cuda())                    # convolutions then run as their own kernel: summation order)
    assert (p3.cpu().double() - p).abs().max().item() < 1e-6 and (v3.cpu().double() - v).abs().max().item() < 1e-5
    # compact queue: rows + device-side count
    inf.fused_blocks = True
    n = x.shape[0]
    rows = torch.arange(n - 1, -1, -1, dtype=torch.int32, device="cuda")
    cnt = torch.tensor([n - 7], dtype=torch.int32, device="cuda")
    pq, vq = inf(x.to(torch.uint8).cuda(), rows=rows, count=cnt)
    assert torch.equal(pq[:n - 7].cpu().double(), p.flip(0)[:n - 7]) and torch.equal(vq[:n - 7].cpu().double(), v.flip(0)[:n - 7])


@pytest.mark.parametrize("n", [1, 3, 257])
def test_conv3x3_c8_operand_pair_output_and_skip(n):
    """The operand-pair output of cz_conv3x3_c8 is the split of its own fp32 output (device conversions = PyTorch's:
    round to nearest even, saturating at 448), and the skip input adds the value the pair stands for.  Since round 4 the
    accumulators START at bias (+ skip) and the products are added on top (no additions left in the epilogue), so the
    skip result equals  (conv + bias) + skip  up to the fp32 rounding of a different summation order -- every one of the
    ~110 accumulation steps rounds at the magnitude of a different partial sum: 2^-18 of the largest, not bit for bit."""
    import torch
    from cchess_alphazero import _native
    c = 128
    g = torch.Generator(device="cuda").manual_seed(70 + n)
    x = (torch.randn((n, 90, c), device="cuda", generator=g) * 2.0).relu()
    x[0, 1, :3] = torch.tensor([700.0, 3e-4, 1.0], device="cuda")
    skip = torch.randn((n, 90, c), device="cuda", generator=g) * 3.0
    w = torch.randn((c, c, 3, 3), device="cuda", generator=g) / (3.0 * c ** 0.5)
    bias = torch.randn((c,), device="cuda", generator=g)
    pk = _native.pack_conv3x3_c8_weights(w).cuda()
    xs, ss = _native.split_c8(x), _native.split_c8(skip)
    f = torch.empty((n, 90, c), device="cuda")
    _native.conv3x3_c8(xs, pk, bias, out_f32=f, relu=False)
    for relu in (False, True):
        pair = (torch.full((n, 90, c), 7.0, device="cuda", dtype=torch.float16),
                torch.full((n, 90, 2 * c), 7, device="cuda", dtype=torch.uint8))
        _native.conv3x3_c8(xs, pk, bias, out=pair, relu=relu)
        want = _native.split_c8(f.relu() if relu else f)
        assert torch.equal(pair[0], want[0]) and torch.equal(pair[1], want[1])
        fs = torch.empty_like(f)
        _native.conv3x3_c8(xs, pk, bias, skip=ss, out_f32=fs, relu=relu)
        ws = f + _native.join_c8(ss)
        ws = ws.relu() if relu else ws
        scale = (f.abs() + _native.join_c8(ss).abs()).max().item()
        assert (fs - ws).abs().max().item() <= 2.0 ** -18 * scale, ((fs - ws).abs().max().item(), scale)
    # the pair reproduces the value to 2^-16 of its magnitude (f16 hi + 4-bit lo; the lo image's smallest step is
    # 2^-9 / 2^11 = 2^-20, which is what small values get), below e4m3's saturation
    v = torch.randn((4, 90, c), device="cuda", generator=g) * 50.0
    back = _native.join_c8(_native.split_c8(v))
    assert ((back - v).abs() <= torch.maximum(v.abs() * 2.0 ** -15, torch.tensor(2.0 ** -20, device="cuda"))).all()


@pytest.mark.parametrize("c", [128, 192])
@pytest.mark.parametrize("n", [1, 3, 257, 700])
def test_resblock_c8_equals_two_c8_convolutions(n, c):
    """cz_resblock with the c8 arithmetic (dtype CZ_F16C8: k_resblock_c8 for 128 filters, k_resblock_ip_c8 for 192) is
    bit-identical to two cz_conv3x3_c8 launches: pair output, fp32 output, in place, device-side count."""
    import torch
    from cchess_alphazero import _native
    g = torch.Generator(device="cuda").manual_seed(90 + n)
    x = torch.randn((n, 90, c), device="cuda", generator=g).relu()
    ws = [torch.randn((c, c, 3, 3), device="cuda", generator=g) / (3.0 * c ** 0.5) for _ in range(2)]
    bs = [torch.randn((c,), device="cuda", generator=g) for _ in range(2)]
    ps = [_native.pack_conv3x3_c8_weights(w).cuda() for w in ws]
    xs = _native.split_c8(x)
    empty = lambda: (torch.full((n, 90, c), 7.0, device="cuda", dtype=torch.float16),
                     torch.full((n, 90, 2 * c), 7, device="cuda", dtype=torch.uint8))
    t, want, got = empty(), empty(), empty()
    _native.conv3x3_c8(xs, ps[0], bs[0], out=t)
    _native.conv3x3_c8(t, ps[1], bs[1], skip=xs, out=want)
    _native.resblock(xs, ps[0], bs[0], ps[1], bs[1], out=got)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    want_f = torch.empty((n, 90, c), device="cuda")
    _native.conv3x3_c8(t, ps[1], bs[1], skip=xs, out_f32=want_f)
    got_f = torch.full((n, 90, c), 7.0, device="cuda")
    _native.resblock(xs, ps[0], bs[0], ps[1], bs[1], out_f32=got_f)
    assert torch.equal(got_f, want_f)
    xi = tuple(a.clone() for a in xs)
    _native.resblock(xi, ps[0], bs[0], ps[1], bs[1], out=xi)
    assert torch.equal(xi[0], want[0]) and torch.equal(xi[1], want[1])
    if n > 3:
        cnt = n // 2
        y = empty()
        _native.resblock(xs, ps[0], bs[0], ps[1], bs[1], out=y, count=torch.tensor([cnt], dtype=torch.int32, device="cuda"))
        assert torch.equal(y[0][:cnt], want[0][:cnt]) and torch.equal(y[1][:cnt], want[1][:cnt])
        assert (y[0][cnt:] == 7.0).all() and (y[1][cnt:] == 7).all()
    if c != 128:
        return
    # the fused head convolutions on this arithmetic: the same block, then the 1x1 convolutions of its fp32 output
    hw = torch.randn((6, c), device="cuda", generator=g) / c ** 0.5
    hb = torch.randn((6,), device="cuda", generator=g)
    pf, vf = torch.empty((n, 4 * 90), device="cuda"), torch.empty((n, 2 * 90), device="cuda")
    _native.resblock_heads(xs, ps[0], bs[0], ps[1], bs[1], hw, hb, 4, pf, vf)
    hd = (want_f.double() @ hw.double().t() + hb.double()).relu()              # [n, 90, 6]
    want_p = hd[..., :4].permute(0, 2, 1).reshape(n, 360)
    want_v = hd[..., 4:].permute(0, 2, 1).reshape(n, 180)
    assert (pf.double() - want_p).abs().max().item() < 2e-5 and (vf.double() - want_v).abs().max().item() < 2e-5


@pytest.mark.parametrize("dt", ["float16", "bfloat16"])
def test_resblock_256_two_channel_tiles_per_wave_is_bit_identical(dt):
    """256 filters, plain operands (BASELINE configs[4] 'deep'): the default schedule gives every matrix wave two
    channel tiles (half the LDS reads per MFMA); the tuning hook's 0 selects the one-tile schedule.  Same MFMA order per
    accumulator, so the results must be identical bit for bit."""
    import torch
    from cchess_alphazero import _native
    dtype = getattr(torch, dt)
    c = 256
    g = torch.Generator(device="cuda").manual_seed(11)
    ws = [torch.randn((c, c, 3, 3), device="cuda", generator=g) / (3.0 * c ** 0.5) for _ in range(2)]
    bs = [torch.randn((c,), device="cuda", generator=g) for _ in range(2)]
    ps = [_native.pack_conv3x3_weights(w, dtype, 1).cuda() for w in ws]
    old = _native.resblock_pipelined(None)
    try:
        for n in (1, 2, 255, 513, 1500):
            xs = _split(torch.randn((n, 90, c), device="cuda", generator=g).relu(), dtype, 1)
            outs = {}
            for mode in (False, True):
                _native.resblock_pipelined(mode)
                y = (torch.full_like(xs[0], 7.0),)
                _native.resblock(xs, ps[0], bs[0], ps[1], bs[1], out=y)
                outs[mode] = y[0]
            assert torch.equal(outs[True], outs[False]), n
    finally:
        _native.resblock_pipelined(old)


@pytest.mark.parametrize("n,npol", [(1, 4), (5, 2), (300, 4)])
def test_resblock_heads_equals_resblock_then_head_convs(n, npol):
    """cz_resblock_heads = cz_resblock (fp32 out) followed by cz_head_convs; only the summation order of the 128-term
    dot products differs: <= 2e-6 relative."""
    import torch
    from cchess_alphazero import _native
    c, dtype = 128, torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(n)
    x = torch.randn((n, 90, c), device="cuda", generator=g).relu()
    ws = [torch.randn((c, c, 3, 3), device="cuda", generator=g) / (3.0 * c ** 0.5) for _ in range(2)]
    bs = [torch.randn((c,), device="cuda", generator=g) for _ in range(2)]
    ps = [_native.pack_conv3x3_weights(w, dtype, 2).cuda() for w in ws]
    hw = torch.randn((6, c), device="cuda", generator=g) / c ** 0.5
    hb = torch.randn((6,), device="cuda", generator=g)
    xs = _split(x, dtype, 2)
    mid = torch.empty((n, 90, c), device="cuda")
    _native.resblock(xs, ps[0], bs[0], ps[1], bs[1], out_f32=mid)
    pf0 = torch.empty((n, npol * 90), device="cuda")
    vf0 = torch.empty((n, (6 - npol) * 90), device="cuda")
    _native.head_convs(mid, hw, hb, npol, pf0, vf0)
    pf = torch.full_like(pf0, 7.0)
    vf = torch.full_like(vf0, 7.0)
    _native.resblock_heads(xs, ps[0], bs[0], ps[1], bs[1], hw, hb, npol, pf, vf)
    scale = max(pf0.abs().max().item(), vf0.abs().max().item())
    assert (pf - pf0).abs().max().item() <= 2e-6 * scale and (vf - vf0).abs().max().item() <= 2e-6 * scale


def test_resblock_rejects_unsupported_shapes():
    import torch
    from cchess_alphazero import _native
    w = _native.pack_conv3x3_weights(torch.randn(32, 32, 3, 3), torch.bfloat16, 2).cuda()
    b = torch.zeros(32, device="cuda")
    x32 = tuple(torch.zeros((1, 90, 32), device="cuda", dtype=torch.bfloat16) for _ in range(2))
    with pytest.raises(_native.NativeError):
        _native.resblock(x32, w, b, w, b, out=x32)


@pytest.mark.parametrize("c,in_planes,dt,parts", [(128, 14, "bfloat16", 2), (128, 28, "bfloat16", 2),
                                                  (128, 14, "float16", 1), (32, 14, "bfloat16", 2),
                                                  (256, 14, "float16", 1), (256, 28, "bfloat16", 2),
                                                  (192, 14, "bfloat16", 2)])
@pytest.mark.parametrize("pdt", ["float32", "uint8"])
def test_input_conv_against_float64(c, in_planes, dt, parts, pdt):
    """cz_input_conv on real feature planes (0/1) in the search kernel's layout against a float64 conv2d of the same
    weights: split mode <= 2e-5 relative (weights carry 2^-17), plain mode the 2-byte rounding of weights + result."""
    import torch
    import torch.nn.functional as F
    from cchess_alphazero import _native
    dtype = getattr(torch, dt)
    n = 9
    g = torch.Generator(device="cuda").manual_seed(c + in_planes)
    planes = (torch.rand((n, in_planes, 10, 9), device="cuda", generator=g) < 0.15).float()
    w = torch.randn((c, in_planes, 5, 5), device="cuda", generator=g) / (5.0 * in_planes ** 0.5)
    b = torch.randn((c,), device="cuda", generator=g)
    wp = _native.pack_input_conv_weights(w, dtype, parts).cuda()
    out = tuple(torch.full((n, 90, c), 7.0, device="cuda", dtype=dtype) for _ in range(parts))
    _native.input_conv(planes.to(getattr(torch, pdt)), wp, b, out)
    got = sum(o.double() for o in out)
    w_eff = w if parts == 2 else w.to(dtype).float()
    ref = torch.relu(F.conv2d(planes.double(), w_eff.double(), b.double(), padding=2)).permute(0, 2, 3, 1)
    ref = ref.reshape(n, 90, c)
    rel = 2e-5 if parts == 2 else (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -10)
    assert (got - ref).abs().max().item() <= rel * ref.abs().max().item()


@pytest.mark.parametrize("c,dt", [(128, "float32"), (256, "float16"), (32, "bfloat16"), (192, "float32")])
def test_head_convs_against_float64(c, dt):
    import torch
    from cchess_alphazero import _native
    n = 37
    g = torch.Generator(device="cuda").manual_seed(c)
    x = torch.randn((n, 90, c), device="cuda", generator=g).relu().to(getattr(torch, dt))
    w = torch.randn((6, c), device="cuda", generator=g) / c ** 0.5
    b = torch.randn((6,), device="cuda", generator=g)
    pf = torch.full((n, 360), 7.0, device="cuda")
    vf = torch.full((n, 180), 7.0, device="cuda")
    _native.head_convs(x, w, b, 4, pf, vf)
    ref = torch.relu(torch.einsum("npc,oc->nop", x.double(), w.double()) + b.double()[None, :, None])
    assert (pf.double() - ref[:, :4].reshape(n, 360)).abs().max().item() <= 2e-6 * ref.abs().max().item()
    assert (vf.double() - ref[:, 4:].reshape(n, 180)).abs().max().item() <= 2e-6 * ref.abs().max().item()


def test_split_bias_act():
    import torch
    from cchess_alphazero import _native
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn((11, 90, 128), device="cuda", generator=g) * 3
    b = torch.randn((128,), device="cuda", generator=g)
    out = tuple(torch.empty((11, 90, 128), device="cuda", dtype=torch.bfloat16) for _ in range(2))
    _native.split_bias_act(x, b, out, relu=True)
    want = torch.relu(x + b)
    hi = want.to(torch.bfloat16)
    assert torch.equal(out[0], hi) and torch.equal(out[1], (want - hi.float()).to(torch.bfloat16))
    assert ((out[0].double() + out[1].double()) - want.double()).abs().max() <= 2.0 ** -16 * want.abs().max()


def test_pack_weights_layout_and_errors():
    import torch
    from cchess_alphazero import _native
    w = torch.randn(128, 128, 3, 3)
    p = _native.pack_conv3x3_weights(w, torch.bfloat16, 2)
    kk_n, ct_n = 8, 4
    part = (9 * kk_n + 3) * ct_n * 64 * 8
    assert p.numel() == 2 * part
    hi = p[:part].view(9 * kk_n + 3, ct_n, 64, 8)
    lo = p[part:].view(9 * kk_n + 3, ct_n, 64, 8)
    for tap, kk, ct, lane, j in ((0, 0, 0, 0, 0), (5, 3, 2, 45, 6), (8, 7, 3, 63, 7)):
        o, ci = ct * 32 + (lane & 31), kk * 16 + (lane >> 5) * 8 + j
        v = w[o, ci, tap // 3, tap % 3]
        assert hi[tap * kk_n + kk, ct, lane, j] == v.to(torch.bfloat16)
        assert lo[tap * kk_n + kk, ct, lane, j] == (v - v.to(torch.bfloat16).float()).to(torch.bfloat16)
    assert hi[72:].abs().sum() == 0                      # prefetch padding
    with pytest.raises(_native.NativeError):
        _native.pack_conv3x3_weights(torch.randn(48, 48, 3, 3), torch.bfloat16, 2)
    with pytest.raises(_native.NativeError):
        x = (torch.zeros((1, 90, 64), device="cuda", dtype=torch.bfloat16),)
        _native.conv3x3(x, p.cuda(), torch.zeros(64, device="cuda"), out=x)


@pytest.mark.parametrize("filters,blocks", [(128, 7), (32, 2), (192, 3), (256, 2)])
def test_network_with_mfma_trunk_matches_fp32_module(filters, blocks):
    """The whole policy/value network with the hand-written split-precision trunk against the plain PyTorch fp32
    module (CPU): policy and value within 1e-4 (north_star tolerance)."""
    import torch
    from cchess_alphazero.agent.model import CChessNet, InferenceNet
    import oracle.xq_oracle as xo
    torch.manual_seed(11)
    net = CChessNet(cnn_filter_num=filters, res_layer_num=blocks)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.3)
            m.running_var.uniform_(0.5, 2.0)
            m.weight.data.normal_(1, 0.2)
            m.bias.data.normal_(0, 0.2)
    net.eval()
    states = [xo.INIT_STATE, '3s5/4m4/9/9/4p4/2R6/9/4C4/4M4/3MS4', 'rkemsmek1/8r/1c5c1/p1p1p1p1p/9/9/P1P1P1P1P/1C2C4/9/RKEMSMEKR']
    boards = np.stack([xo.state_to_board(s) for s in states] * 4 + [xo.state_to_board(states[0])])
    x = torch.from_numpy(np.stack([xo.planes_board(b) for b in boards]))
    with torch.no_grad():
        p_ref, v_ref = net(x)
    inf = InferenceNet(net, torch.float32, trunk="mfma").cuda()
    p, v = inf(x.cuda())
    assert (p.cpu() - p_ref).abs().max().item() < 1e-4
    assert (v.cpu() - v_ref).abs().max().item() < 1e-4
    # the softmax outputs are ~5e-4 each, so the absolute bound alone says little: also a bound in LOGIT space
    # (log-probabilities up to the common shift) and a relative bound on every probability
    lg, lr = torch.log(p.cpu().clamp_min(1e-30)), torch.log(p_ref.clamp_min(1e-30))
    assert ((lg - lg.mean(1, keepdim=True)) - (lr - lr.mean(1, keepdim=True))).abs().max().item() < 1e-3
    assert ((p.cpu() - p_ref).abs() / p_ref.clamp_min(1e-12)).max().item() < 1e-3
    lib = InferenceNet(net, torch.float32, trunk="library").cuda()
    p2, v2 = lib(x.cuda())
    assert (p - p2).abs().max().item() < 1e-4 and (v - v2).abs().max().item() < 1e-4
    inf.fused_heads = False                  # separate head-convolution launch
    p6, v6 = inf(x.cuda())
    assert (p6 - p).abs().max().item() < 1e-6 and (v6 - v).abs().max().item() < 1e-5
    inf.fused_blocks = False                 # per-convolution launches: identical trunk arithmetic (the library
    p5, v5 = inf(x.cuda())                   # kernels around it may change solver between calls, hence not torch.equal)
    assert (p5 - p).abs().max().item() < 1e-6 and (v5 - v).abs().max().item() < 1e-5
    for dt, tol in ((torch.bfloat16, 3e-2), (torch.float16, 5e-3)):
        lo = InferenceNet(net, dt, trunk="mfma").cuda()
        p3, v3 = lo(x.cuda())
        assert (p3.cpu() - p_ref).abs().max().item() < tol and (v3.cpu() - v_ref).abs().max().item() < tol * 10
    with pytest.raises(RuntimeError):
        InferenceNet(net, torch.float32, trunk="mfma")(x)          # no CPU implementation of the HIP trunk


def test_deep_network_fp16_matches_fp32_module(positions_1k):
    """BASELINE configs[4] ("deep net stress: 20-block x 256-filter ResNet, fp16 MFMA eval"; architecture: reference
    agent/model.py:32-83): the whole 20 x 256 network on plain fp16 operands (k_resblock<256>, fp32 accumulate) against
    the plain PyTorch fp32 module on the CPU, BatchNorm statistics perturbed so that the folding is exercised, on 64
    real positions of the 1k suite.  The bound is the derived one bench.py prints for this configuration
    (bench.py::fp16_tolerance: 2^-11 operand rounding, random-sign accumulation, 41 layers in quadrature:
    4.5e-3 on the value and on the logits, 1e-4 on the probabilities; measured 2.3e-4 / 8.4e-4 / 6e-7) -- it is fp16's bound, not north_star's 1e-4, which
    is the split-precision default's (tested above for 256 filters as well)."""
    import torch
    from cchess_alphazero.agent.model import CChessNet, InferenceNet
    import oracle.xq_oracle as xo
    import bench                                      # (repo root is on sys.path: tests/conftest.py)
    torch.manual_seed(13)
    net = CChessNet(cnn_filter_num=256, res_layer_num=20)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.7, 1.4)
            m.weight.data.normal_(1, 0.1)
            m.bias.data.normal_(0, 0.1)
    net.eval()
    states = [r["state"] for r in positions_1k if not r["done"][0]][::14][:64]
    assert len(states) == 64
    x = torch.from_numpy(np.stack([xo.planes_board(xo.state_to_board(s)) for s in states]))
    with torch.no_grad():
        p_ref, v_ref = net(x)
    tol = bench.fp16_tolerance(20, 256)
    assert 3e-3 < tol["value_abs"] < 6e-3
    inf = InferenceNet(net, torch.float16, trunk="mfma").cuda()
    p, v = inf(x.to(torch.uint8).cuda())
    lg, lr = torch.log(p.cpu().clamp_min(1e-30)), torch.log(p_ref.clamp_min(1e-30))
    dlogit = ((lg - lg.mean(1, keepdim=True)) - (lr - lr.mean(1, keepdim=True))).abs().max().item()
    dp, dv = (p.cpu() - p_ref).abs().max().item(), (v.cpu() - v_ref).abs().max().item()
    print(f"deep fp16 vs fp32 module: policy {dp:.3e}, logit {dlogit:.3e}, value {dv:.3e}; bound {tol}")
    assert dp < tol["policy_abs"] and dlogit < tol["policy_logit_abs"] and dv < tol["value_abs"], (dp, dlogit, dv)
    assert v_ref.abs().max() > 0.05                  # (the value head is not saturated at 0: the check means something)
    # the same network with split-precision operands stays inside north_star's 1e-4
    sp = InferenceNet(net, torch.float32, trunk="mfma").cuda()
    p2, v2 = sp(x.to(torch.uint8).cuda())
    assert (p2.cpu() - p_ref).abs().max().item() < 1e-4 and (v2.cpu() - v_ref).abs().max().item() < 1e-4


@pytest.mark.parametrize("arith", ["bf16x3", "c8"])
def test_network_with_history_planes_and_reference_head_shapes(arith):
    """28 input planes (use_history) and the 2-policy / 4-value head filters of the reference's published topologies
    (data/model/model_128f.json) through the hand-written path, uint8 planes as the engine feeds them; both tower
    arithmetics."""
    import torch
    from cchess_alphazero.agent.model import CChessNet, InferenceNet
    torch.manual_seed(5)
    net = CChessNet(cnn_filter_num=128, res_layer_num=3, input_depth=28, policy_filters=2, value_filters=4).eval()
    g = torch.Generator().manual_seed(1)
    x = (torch.rand((9, 28, 10, 9), generator=g) < 0.12).float()
    with torch.no_grad():
        p_ref, v_ref = net(x)
    inf = InferenceNet(net, torch.float32, trunk="mfma", arith=arith).cuda()
    assert inf.arith == arith
    p, v = inf(x.to(torch.uint8).cuda())
    assert (p.cpu() - p_ref).abs().max().item() < 1e-4 and (v.cpu() - v_ref).abs().max().item() < 1e-4
    p2, v2 = inf(x.cuda())                                   # fp32 planes take the separate input-layer kernel (split-bf16
    assert (p - p2).abs().max().item() < 1e-6 and (v - v2).abs().max().item() < 1e-5      # products): same answer to rounding
    inf.fused_input = False                                  # ... which uint8 planes take too when the fusion is off: identical
    p3, v3 = inf(x.to(torch.uint8).cuda())
    assert torch.equal(p2, p3) and torch.equal(v2, v3)


@pytest.mark.parametrize("filters,arith", [(128, "bf16x3"), (128, "c8"), (192, "bf16x3")])
def test_network_on_a_compact_queue_matches_the_gathered_batch(filters, arith):
    """cz_*_q (compact evaluation queue): rows / count live on the device.  The network evaluated on
    (planes, rows, count) gives, in its first `count` result rows, what it gives on the gathered batch
    planes[rows[:count]] -- for count = 0, 1, an odd number and the whole queue -- and leaves the launch shapes alone.
    128 filters (k_resblock_pipe / k_resblock with fused heads; the c8 tower: k_input_conv<C8> + k_resblock<C8>) and 192
    (k_resblock_ip + cz_head_convs)."""
    import torch
    from cchess_alphazero.agent.model import CChessNet, InferenceNet
    torch.manual_seed(11)
    raw = CChessNet(cnn_filter_num=filters, res_layer_num=3).eval()
    for m in raw.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    net = InferenceNet(raw, torch.float32, trunk="mfma", arith=arith).cuda()
    assert net.supports_compact_queue() and net.arith == arith
    n = 77
    planes = (torch.rand((n, 14, 10, 9), device="cuda") < 0.07).to(torch.uint8)
    perm = torch.randperm(n, device="cuda").to(torch.int32)
    for count in (0, 1, 33, n):
        cnt = torch.tensor([count], dtype=torch.int32, device="cuda")
        p, v = net(planes, rows=perm, count=cnt)
        assert p.shape == (n, 2086) and v.shape == (n,)
        if count:
            pg, vg = net(planes[perm[:count].long()].contiguous())
            # (the convolutional part is bit-identical; the dense layers may pick another GEMM tiling for another M)
            assert (p[:count] - pg).abs().max() < 1e-6 and (v[:count] - vg).abs().max() < 1e-6
    # a count larger than the queue is clamped to it
    p, v = net(planes, rows=perm, count=torch.tensor([10 * n], dtype=torch.int32, device="cuda"))
    pg, vg = net(planes[perm.long()].contiguous())
    assert (p - pg).abs().max() < 1e-6 and (v - vg).abs().max() < 1e-6


def test_resblock_c8_deferred_epilogue_is_bit_identical_to_two_convolutions():
    """k_resblock_c8 (round 4: accumulators start at bias / bias + skip, the second epilogue of a board deferred into the
    fp8 slots of the next board's first K loop, X rewritten under K loop 2) against two cz_conv3x3_c8 launches: bit for
    bit, for one board per workgroup, several, one more than the CUs, an odd many (every workgroup's first, middle and
    last board take different paths); large / tiny activations; in place; device-side count."""
    import torch
    from cchess_alphazero import _native
    c = 128
    g = torch.Generator(device="cuda").manual_seed(17)
    ws = [torch.randn((c, c, 3, 3), device="cuda", generator=g) / (3.0 * c ** 0.5) for _ in range(2)]
    bs = [torch.randn((c,), device="cuda", generator=g) * 0.3 for _ in range(2)]
    ps = [_native.pack_conv3x3_c8_weights(w).cuda() for w in ws]
    for n in (1, 2, 5, 257, 1031, 2100):
        x = (torch.randn((n, 90, c), device="cuda", generator=g) * 1.5).relu()
        x[0, 0, :4] = torch.tensor([300.0, 2e-3, 5e-5, 0.0], device="cuda")
        xs = _native.split_c8(x)
        empty = lambda: (torch.full((n, 90, c), 7.0, device="cuda", dtype=torch.float16),
                         torch.full((n, 90, 2 * c), 7, device="cuda", dtype=torch.uint8))
        got = _native.resblock(xs, ps[0], bs[0], ps[1], bs[1], out=empty())
        t, want = empty(), empty()
        _native.conv3x3_c8(xs, ps[0], bs[0], out=t)
        _native.conv3x3_c8(t, ps[1], bs[1], skip=xs, out=want)
        for part in range(2):
            assert torch.equal(got[part], want[part]), (n, part)
    n, cnt = 300, 123
    xs = _native.split_c8((torch.randn((n, 90, c), device="cuda", generator=g) * 1.5).relu())
    ref = _native.resblock(xs, ps[0], bs[0], ps[1], bs[1], out=(torch.empty_like(xs[0]), torch.empty_like(xs[1])))
    y = tuple(t.clone() for t in xs)
    _native.resblock(y, ps[0], bs[0], ps[1], bs[1], out=y, count=torch.tensor([cnt], dtype=torch.int32, device="cuda"))
    for part in range(2):
        assert torch.equal(y[part][:cnt], ref[part][:cnt]) and torch.equal(y[part][cnt:], xs[part][cnt:])


@pytest.mark.parametrize("dt", ["bfloat16", "float16"])
def test_pipelined_resblock_is_bit_identical_to_the_plain_schedule(dt):
    """k_resblock_pipe (epilogue 2 of a board under the next board's first K loop, result written in place over the skip
    operand, two barriers per board) against k_resblock: the same arithmetic in the same order, so the (hi, lo) outputs
    are equal bit for bit -- for one board, a few, one more than the CUs, an odd many; also in place and with the
    board count taken from the device."""
    import torch
    from cchess_alphazero import _native
    dtype = getattr(torch, dt)
    torch.manual_seed(5)
    c = 128
    w1, w2 = torch.randn(c, c, 3, 3) * 0.05, torch.randn(c, c, 3, 3) * 0.05
    b1, b2 = (torch.randn(c) * 0.1).cuda(), (torch.randn(c) * 0.1).cuda()
    p1, p2 = _native.pack_conv3x3_weights(w1, dtype, 2).cuda(), _native.pack_conv3x3_weights(w2, dtype, 2).cuda()
    old = _native.resblock_pipelined(None)
    try:
        for n in (1, 2, 5, 257, 1031):
            x = _split(torch.randn(n, 90, c).cuda(), dtype, 2)
            outs = {}
            for mode in (False, True):
                _native.resblock_pipelined(mode)
                y = tuple(torch.full_like(t, 7.0) for t in x)
                _native.resblock(x, p1, b1, p2, b2, out=y)
                outs[mode] = y
            assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1]), n
        # in place, and a device-side count smaller than the buffers (the tail stays untouched)
        _native.resblock_pipelined(True)
        n, cnt = 300, 123
        x = _split(torch.randn(n, 90, c).cuda(), dtype, 2)
        ref = tuple(torch.empty_like(t) for t in x)
        _native.resblock_pipelined(False)
        _native.resblock(x, p1, b1, p2, b2, out=ref)
        _native.resblock_pipelined(True)
        y = tuple(t.clone() for t in x)
        _native.resblock(y, p1, b1, p2, b2, out=y, count=torch.tensor([cnt], dtype=torch.int32, device="cuda"))
        for part in range(2):
            assert torch.equal(y[part][:cnt], ref[part][:cnt]) and torch.equal(y[part][cnt:], x[part][cnt:])
    finally:
        _native.resblock_pipelined(old)


@pytest.mark.parametrize("pair", ["bfloat16", "float16"])
@pytest.mark.parametrize("fp,fv", [(360, 180), (180, 360)])
def test_heads_tail_matches_float64(fp, fv, pair):
    """cz_heads_tail (csrc/xq_heads.hip: Dense(2086) + softmax, Dense(256) + ReLU + Dense(1) + tanh; reference
    agent/model.py:58-59,64-66) against float64 PyTorch on the same fp32 inputs.  Tolerance, derived: every product is
    formed from (hi, lo) bf16 pairs with the lo*lo term dropped -- relative error <= 3 * 2^-17 < 2^-15 per product -- and
    accumulated in fp32, so |d logit| <= B = 2^-15 * sum_k |w_k| |x_k| (+ 1e-6 for the accumulation); a softmax output
    then moves by at most 2 B relative, tanh is 1-Lipschitz.  With (hi, lo) fp16 pairs (the default since round 4: 22 bits per
    operand, the weights' lo parts are fp16 subnormals here and must be honoured) the per-product bound is 2^-19.  Sizes
    around the 64-position tile, the reference's two head shapes (4 + 2 and 2 + 4 filters), and the compact queue's
    device-side count."""
    import torch
    from cchess_alphazero import _native
    torch.manual_seed(3)
    pdt = getattr(torch, pair)
    eps = 2.0 ** -15 if pair == "bfloat16" else 2.0 ** -19
    n_lab, n_hid = 2086, 256
    wp, bp = torch.randn(n_lab, fp) * 0.08, torch.randn(n_lab) * 0.5
    w1, b1 = torch.randn(n_hid, fv) * 0.1, torch.randn(n_hid) * 0.2
    w2, b2 = torch.randn(n_hid) * 0.2, 0.13
    pk_p, pk_1 = _native.pack_fc_weights(wp, pdt).cuda(), _native.pack_fc_weights(w1, pdt).cuda()
    for n, cnt in ((1, None), (63, None), (64, None), (65, None), (1000, None), (300, 123), (70, 0)):
        pf = torch.relu(torch.randn(n, fp)) * 1.5
        vf = torch.relu(torch.randn(n, fv)) * 1.5
        pol = torch.full((n, n_lab), 7.0, device="cuda")
        val = torch.full((n,), 7.0, device="cuda")
        stats = torch.empty((n, 2), device="cuda")
        count = None if cnt is None else torch.tensor([cnt], dtype=torch.int32, device="cuda")
        _native.heads_tail(pf.cuda(), vf.cuda(), pk_p, bp.cuda(), pk_1, b1.cuda(), w2.cuda(), b2, pol, val, stats, count=count)
        m = n if cnt is None else cnt
        assert torch.all(pol[m:] == 7.0) and torch.all(val[m:] == 7.0)            # rows beyond the count are untouched
        if m == 0:
            continue
        lg = pf[:m].double() @ wp.double().T + bp.double()
        p_ref = torch.softmax(lg, dim=1)
        bound = eps * (pf[:m].abs().double() @ wp.abs().double().T).max().item() + 1e-6
        p = pol[:m].cpu().double()
        assert torch.isfinite(p).all() and (p.sum(1) - 1).abs().max() < 1e-5
        assert ((p - p_ref).abs() <= 2.5 * bound * p_ref + 1e-12).all(), ((p - p_ref).abs() / p_ref).max()
        hid = torch.relu(vf[:m].double() @ w1.double().T + b1.double())
        v_ref = torch.tanh(hid @ w2.double() + b2)
        vb = ((eps * (vf[:m].abs().double() @ w1.abs().double().T) + 1e-6) @ w2.abs().double()).max().item() + 1e-5
        assert (val[:m].cpu().double() - v_ref).abs().max().item() <= vb
        # and against what the library path computes in fp32 (the path it replaces): well inside 1e-5
        p32 = torch.softmax(pf[:m].cuda() @ wp.cuda().T + bp.cuda(), dim=1)
        assert (pol[:m] - p32).abs().max().item() < 1e-5
        # normalize=False (the engine's queue; cz_search_policy_logits): the same rows as raw logits, whose float32
        # softmax is the normalised output to rounding
        raw = torch.full((n, n_lab), 7.0, device="cuda")
        _native.heads_tail(pf.cuda(), vf.cuda(), pk_p, bp.cuda(), pk_1, b1.cuda(), w2.cuda(), b2, raw, val, stats, count=count,
                           normalize=False)
        assert torch.all(raw[m:] == 7.0)
        assert (raw[:m].cpu().double() - lg).abs().max().item() <= bound
        assert (torch.softmax(raw[:m], dim=1) - pol[:m]).abs().max().item() < 1e-6


def test_network_tail_switch_gives_the_same_outputs():
    """InferenceNet with the hand-written dense tail (default) vs the hipBLASLt / PyTorch tail it replaces."""
    import torch
    from cchess_alphazero.agent.model import CChessNet, InferenceNet
    torch.manual_seed(4)
    raw = CChessNet(cnn_filter_num=128, res_layer_num=2).eval()
    net = InferenceNet(raw, torch.float32, trunk="mfma").cuda()
    planes = (torch.rand((130, 14, 10, 9), device="cuda") < 0.07).to(torch.uint8)
    assert net.fused_tail
    p, v = net(planes)
    net.fused_tail = False
    p0, v0 = net(planes)
    assert (p - p0).abs().max().item() < 2e-7 and (v - v0).abs().max().item() < 2e-6
    with torch.no_grad():
        pr, vr = raw(planes.float().cpu())
    assert (p.cpu() - pr).abs().max().item() < 1e-4 and (v.cpu() - vr).abs().max().item() < 1e-4


@pytest.mark.parametrize("in_planes", [14, 28])
def test_input_resblock_c8_matches_input_layer_then_resblock(in_planes):
    """cz_input_resblock with dtype CZ_F16C8 (k_resblock_c8<FIRST>: the copy waves' fp32 gather produces the c8 operand
    triple) against the float64 input layer, split with split_c8, followed by cz_resblock on the c8
    arithmetic.  The input layers differ by fp32 summation only (<= 4e-6 relative), the blocks are the same arithmetic:
    outputs agree to 3e-5 of the largest value; the compact queue's gathered rows reproduce the plain batch bit for bit."""
    import torch
    import torch.nn.functional as F
    from cchess_alphazero import _native
    c = 128
    g = torch.Generator().manual_seed(100 + in_planes)
    w_in = torch.randn((c, in_planes, 5, 5), generator=g) * 0.2
    b_in = torch.randn((c,), generator=g) * 0.1
    ws = [torch.randn((c, c, 3, 3), generator=g) / (3.0 * c ** 0.5) for _ in range(2)]
    bs = [(torch.randn((c,), generator=g) * 0.1).cuda() for _ in range(2)]
    ps = [_native.pack_conv3x3_c8_weights(w).cuda() for w in ws]
    table, bias = _native.input_table(w_in).cuda(), b_in.cuda()
    for n in (1, 2, 255, 257, 700):
        planes = torch.zeros((n, in_planes, 10, 9), dtype=torch.uint8)
        for grp in range(in_planes // 14):
            occ = torch.rand((n, 10, 9), generator=g) < 0.36
            which = torch.randint(0, 14, (n, 10, 9), generator=g)
            planes[:, grp * 14:(grp + 1) * 14].scatter_(1, which.unsqueeze(1), occ.unsqueeze(1).to(torch.uint8))
        x = F.relu(F.conv2d(planes.double(), w_in.double(), b_in.double(), padding=2)).float()
        xs = _native.split_c8(x.permute(0, 2, 3, 1).reshape(n, 90, c).contiguous().cuda())
        empty = lambda: (torch.full((n, 90, c), 7.0, device="cuda", dtype=torch.float16),
                         torch.full((n, 90, 2 * c), 7, device="cuda", dtype=torch.uint8))
        want = _native.resblock(xs, ps[0], bs[0], ps[1], bs[1], out=empty())
        got = _native.input_resblock(planes.cuda(), table, bias, ps[0], bs[0], ps[1], bs[1], out=empty())
        y, y_ref = _native.join_c8(got), _native.join_c8(want)
        assert torch.isfinite(y).all()
        assert (y - y_ref).abs().max().item() <= 3e-5 * y_ref.abs().max().item(), n
        # the value bytes are the e4m3 of the value the pair stands for (what the next block's correction term reads)
        h8 = got[1][..., c:].contiguous().view(torch.float8_e4m3fn).float()
        assert (h8 - y).abs().max().item() <= 2.0 ** -4 * y.abs().max().item() + 2.0 ** -9
        if n == 700:
            perm = torch.randperm(n, generator=g).to(torch.int32).cuda()
            cnt = torch.tensor([333], dtype=torch.int32, device="cuda")
            got2 = _native.input_resblock(planes.cuda(), table, bias, ps[0], bs[0], ps[1], bs[1], out=empty(), rows=perm, count=cnt)
            for part in range(2):
                assert torch.equal(got2[part][:333], got[part][perm[:333].long()])
                assert torch.all(got2[part][333:] == 7)


@pytest.mark.parametrize("in_planes,dt", [(14, "bfloat16"), (28, "bfloat16"), (14, "float16")])
def test_input_resblock_matches_input_layer_then_resblock(in_planes, dt):
    """cz_input_resblock (k_resblock_pipe<FIRST>: the 5x5 input layer as an fp32 gather over the occupied squares, done by
    the first block's copy waves; reference agent/model.py:36-45) against the float64 input layer followed by
    cz_resblock on its (hi, lo) split.  The only difference is the rounding of the input layer: fp32 sums of <= 51 terms
    (<= 4e-6 relative to the largest term sum) against float64 rounded to fp32 -- the block's output, two split-precision
    convolutions later, agrees to 2e-5 of its largest value.  One-hot planes as the engine writes them (and a second bit
    per square for the history planes), board counts around the CU count, the compact queue's rows / count."""
    import torch
    import torch.nn.functional as F
    from cchess_alphazero import _native
    dtype = getattr(torch, dt)
    c = 128
    g = torch.Generator().manual_seed(in_planes)
    w_in = torch.randn((c, in_planes, 5, 5), generator=g) * 0.2
    b_in = torch.randn((c,), generator=g) * 0.1
    ws = [torch.randn((c, c, 3, 3), generator=g) / (3.0 * c ** 0.5) for _ in range(2)]
    bs = [(torch.randn((c,), generator=g) * 0.1).cuda() for _ in range(2)]
    ps = [_native.pack_conv3x3_weights(w, dtype, 2).cuda() for w in ws]
    table, bias = _native.input_table(w_in).cuda(), b_in.cuda()
    for n in (1, 2, 255, 257, 700):
        planes = torch.zeros((n, in_planes, 10, 9), dtype=torch.uint8)
        for grp in range(in_planes // 14):                       # one plane per occupied square and 14-plane group
            occ = torch.rand((n, 10, 9), generator=g) < 0.36
            which = torch.randint(0, 14, (n, 10, 9), generator=g)
            planes[:, grp * 14:(grp + 1) * 14].scatter_(1, which.unsqueeze(1), occ.unsqueeze(1).to(torch.uint8))
        x = F.relu(F.conv2d(planes.double(), w_in.double(), b_in.double(), padding=2)).float()      # [n, c, 10, 9]
        xs = _split(x.permute(0, 2, 3, 1).reshape(n, 90, c).contiguous().cuda(), dtype, 2)
        want = tuple(torch.empty_like(xs[0]) for _ in range(2))
        old = _native.resblock_pipelined(None)
        try:
            _native.resblock(xs, ps[0], bs[0], ps[1], bs[1], out=want)
        finally:
            _native.resblock_pipelined(old)
        got = tuple(torch.full_like(xs[0], 7.0) for _ in range(2))
        _native.input_resblock(planes.cuda(), table, bias, ps[0], bs[0], ps[1], bs[1], out=got)
        y, y_ref = got[0].float() + got[1].float(), want[0].float() + want[1].float()
        assert torch.isfinite(y).all()
        assert (y - y_ref).abs().max().item() <= 2e-5 * y_ref.abs().max().item(), n
        if n == 700:                                              # compact queue: gathered rows, device-side count
            perm = torch.randperm(n, generator=g).to(torch.int32).cuda()
            cnt = torch.tensor([333], dtype=torch.int32, device="cuda")
            got2 = tuple(torch.full_like(xs[0], 7.0) for _ in range(2))
            _native.input_resblock(planes.cuda(), table, bias, ps[0], bs[0], ps[1], bs[1], out=got2, rows=perm, count=cnt)
            for part in range(2):
                assert torch.equal(got2[part][:333], got[part][perm[:333].long()])
                assert torch.all(got2[part][333:] == 7.0)


@pytest.mark.parametrize("arith", ["bf16x3", "f16x3", "c8"])
def test_network_with_fused_input_layer_switch(arith):
    """InferenceNet with the input layer inside the first block's launch (default) vs the separate cz_input_conv kernel."""
    import torch
    from cchess_alphazero.agent.model import CChessNet, InferenceNet
    torch.manual_seed(6)
    raw = CChessNet(cnn_filter_num=128, res_layer_num=3).eval()
    net = InferenceNet(raw, torch.float32, trunk="mfma", arith=arith).cuda()
    planes = torch.zeros((150, 14, 10, 9), dtype=torch.uint8)
    occ = torch.rand((150, 10, 9)) < 0.3
    planes.scatter_(1, torch.randint(0, 14, (150, 1, 10, 9)), occ.unsqueeze(1).to(torch.uint8))
    assert net.fused_input
    p, v = net(planes.cuda())
    net.fused_input = False
    p0, v0 = net(planes.cuda())
    assert (p - p0).abs().max().item() < 1e-6 and (v - v0).abs().max().item() < 1e-5
    with torch.no_grad():
        pr, vr = raw(planes.float())
    assert (p.cpu() - pr).abs().max().item() < 1e-4 and (v.cpu() - vr).abs().max().item() < 1e-4
