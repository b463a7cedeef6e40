#!/usr/bin/env python3
"""CPU emulation of candidate split arithmetics for the residual tower ("cheaper correction terms").

The tower's 3 x 3 convolutions compute  w x  as  w_hi x_hi + w_lo x_hi + w_hi x_lo  on bf16 matrix instructions (three per
product).  tools/mfma_mix.py measured that an fp16 main term + two block-scaled fp8 (e4m3, K = 64) correction MFMAs retires
1.47x faster in the power-capped state.  This script answers the numerical half of the question on the CPU, layer by layer
and end to end: with the operands rounded exactly as those instructions would see them, how far are policy / value from
the float64 network?  (north_star tolerance: 1e-4; today's bf16x3 path: logit 2e-6, value 1e-7.)

Operand models (activation x and folded filter w, both float32; blocks of 32 input channels share a power-of-two scale,
the MX convention of v_mfma_scale_f32_32x32x64_f8f6f4):
    bf16x3      x_hi = bf16(x), x_lo = bf16(x - x_hi), same for w; three exact products                       (today)
    f16+fp8     x_hi = f16(x), x_lo8 = fp8(x - x_hi), x_hi8 = fp8(x);  w likewise;   w_hi x_hi + w_hi8 x_lo8 + w_lo8 x_hi8
    f16+fp6     the same with e2m3 correction operands
    f16 only    w_hi x_hi                                                                              (no corrections)
Products and sums are evaluated in float64 (the instructions' own accumulation error is measured separately:
tools/probes/mfma_mix_probe check -> 6e-5 of the sum of |terms| of a K = 64 fp8 block, i.e. 2^-26 of the main term).

    python tools/emulate_fp8_corrections.py [--filters 128 --blocks 7 --positions 24]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chinesechess-alphazero_amd"))
sys.path.insert(0, ROOT)


UNIFORM = False     # --uniform-scales: one power-of-two scale per tensor (what a constant scale operand gives) instead of per block


def block_scaled(x, dim, fmt):
    """Round x (float64 tensor) to `fmt` elements with one power-of-two scale per block of 32 along `dim`
    (or per tensor with --uniform-scales: then the element format's own exponent range carries the dynamics)."""
    x = x.movedim(dim, -1)
    shp = x.shape
    c = shp[-1]
    assert c % 32 == 0
    xb = x.reshape(*shp[:-1], c // 32, 32)
    emax = {"e4m3": 8, "e2m3": 2}[fmt]                     # largest binade of the element format (448 = 1.75 * 2^8; 7.5 = 1.875 * 2^2)
    if UNIFORM:
        amax = xb.abs().max().clamp_min(1e-300)
    else:
        amax = xb.abs().amax(dim=-1, keepdim=True).clamp_min(1e-300)
    scale = torch.exp2(torch.floor(torch.log2(amax)) - emax)
    y = xb / scale
    if fmt == "e4m3":
        q = y.clamp(-448.0, 448.0).to(torch.float32).to(torch.float8_e4m3fn).to(torch.float64)      # MX: saturating
    else:                                                   # e2m3: sign, 2 exponent bits (bias 1), 3 mantissa bits; max 7.5, subnormal step 0.125
        a = y.abs().clamp_max(7.5)
        e = torch.floor(torch.log2(a.clamp_min(1e-300))).clamp(min=0.0, max=2.0)
        step = torch.exp2(e - 3)
        q = torch.sign(y) * torch.round(a / step) * step    # (round-half-even of torch.round on the grid)
    return (q * scale).reshape(shp).movedim(-1, dim)


FP6 = {"e2m3": (2, 3, 1), "e3m2": (3, 2, 3)}          # exponent bits, mantissa bits, bias (OCP MX: no inf / nan, saturating)


def fp6(y, fmt):
    """Round float64 y to the fp6 grid of `fmt` (round to nearest even, saturating)."""
    eb, mb, bias = FP6[fmt]
    emax = (1 << eb) - 1 - bias
    top = (2.0 - 2.0 ** -mb) * 2.0 ** emax
    a = y.abs().clamp_max(top)
    e = torch.floor(torch.log2(a.clamp_min(1e-300))).clamp(min=1.0 - bias, max=float(emax))
    step = torch.exp2(e - mb)
    return torch.sign(y) * (torch.round(a / step) * step).clamp_max(top)


def c6_operands(x, w, fmt, xscale):
    """The operand model of a c6 kernel: fp16 main operands; 6-bit correction operands with the instruction's E8M0 scale
    per 32 input channels.  Filters: exact per-(output, tap, block) scales for w and for w - f16(w) (host side, free).
    Activations: xscale = "block": per (pixel, block) the binade of the block's largest |x| (the lo image's scale is that
    times 2^-11, no second reduction); "fixed": one scale for the tensor from its maximum (a calibration constant)."""
    eb, mb, bias = FP6[fmt]
    emax = (1 << eb) - 1 - bias

    def blocks(t, dim):
        t = t.movedim(dim, -1)
        return t.reshape(*t.shape[:-1], t.shape[-1] // 32, 32), t.shape

    def back(tb, shp, dim):
        return tb.reshape(shp).movedim(-1, dim)

    xh = x.to(torch.float32).to(torch.float16).to(torch.float64)
    xb, shp = blocks(x, 1)
    xlb, _ = blocks(x - xh, 1)
    if xscale.startswith("block"):
        amax = xb.abs().amax(dim=-1, keepdim=True).clamp_min(2.0 ** -24)
    else:
        amax = xb.abs().max().clamp_min(2.0 ** -24)
    sc = torch.exp2(torch.floor(torch.log2(amax)) - emax)
    xh6 = back(fp6(xb / sc, fmt) * sc, shp, 1)
    scl = sc * 2.0 ** -11
    xl6 = back(fp6(xlb / scl, fmt) * scl, shp, 1)
    wh = w.to(torch.float32).to(torch.float16).to(torch.float64)

    def wq(t):
        tb, ws = blocks(t, 1)
        if xscale.endswith("W"):                              # one scale for the whole filter tensor, like c8
            am = tb.abs().max().clamp_min(1e-300)
        else:
            am = tb.abs().amax(dim=-1, keepdim=True).clamp_min(1e-300)
        s = torch.exp2(torch.floor(torch.log2(am)) - emax)
        return back(fp6(tb / s, fmt) * s, ws, 1)
    return xh, xl6, xh6, wh, wq(w - wh), wq(w)


def conv_model(x, w, b, mode, pad):
    """x [n, c, 10, 9] float64 (exact activations of the previous layer as the kernel would hold them), w [o, c, k, k]."""
    conv = lambda a, ww: F.conv2d(a, ww, None, padding=pad)
    if mode == "f64":
        y = conv(x, w)
    elif mode == "bf16x3":
        xh = x.to(torch.float32).to(torch.bfloat16).to(torch.float64)
        xl = (x - xh).to(torch.float32).to(torch.bfloat16).to(torch.float64)
        wh = w.to(torch.float32).to(torch.bfloat16).to(torch.float64)
        wl = (w - wh).to(torch.float32).to(torch.bfloat16).to(torch.float64)
        y = conv(xh, wh) + conv(xh, wl) + conv(xl, wh)
    elif mode in ("f16x3", "f16x3-ftz"):
        # (hi, lo) pairs of fp16: 22 bits per operand where fp16's range holds them; -ftz: the matrix unit flushing fp16
        # subnormals to zero (worst case for the lo parts)
        def h(t):
            r = t.to(torch.float32).to(torch.float16).to(torch.float64)
            if mode == "f16x3-ftz":
                r = torch.where(r.abs() < 2.0 ** -14, torch.zeros_like(r), r)
            return r
        xh, wh = h(x), h(w)
        xl, wl = h(x - xh), h(w - wh)
        y = conv(xh, wh) + conv(xh, wl) + conv(xl, wh)
    elif mode == "c8-ef":
        # c8-kernel, but the e4m3 rounding of the filters' lo parts is chosen by ERROR DIFFUSION along the input channels,
        # weighted by each channel's mean activation (what a calibration pass knows): the systematic part of the arithmetic's
        # error -- sum_c (w_lo - e4m3(w_lo)) mean(x_c), the same for every position -- is driven to zero per output and tap
        f8 = lambda t: t.clamp(-448.0, 448.0).to(torch.float32).to(torch.float8_e4m3fn).to(torch.float64)
        xh = x.to(torch.float32).to(torch.float16).to(torch.float64)
        wh = w.to(torch.float32).to(torch.float16).to(torch.float64)
        xl8 = f8((x - xh) * 2048.0) / 2048.0
        xh8 = f8(x)
        sh = 2.0 ** (7 - torch.floor(torch.log2(w.abs().max())))
        wl = w - wh
        sl = 2.0 ** (7 - torch.floor(torch.log2(wl.abs().max().clamp_min(1e-300))))
        mu = xh8.mean(dim=(0, 2, 3)).clamp_min(1e-12)                     # [c]
        t = wl * sl                                                        # [o, c, k, k] in e4m3 units
        near = f8(t)
        # the neighbour on the other side of t on the e4m3 grid
        ulp = torch.where(near.abs() >= 2.0 ** -6, torch.exp2(torch.floor(torch.log2(near.abs().clamp_min(2.0 ** -9))) - 3), torch.full_like(t, 2.0 ** -9))
        other = f8(near + torch.sign(t - near + 1e-300) * ulp)
        q = torch.empty_like(t)
        r = torch.zeros_like(t[:, 0])                                      # running error per (o, k, k), weighted
        order = torch.argsort(mu, descending=True)
        for c in order.tolist():
            e_near = r + (t[:, c] - near[:, c]) * mu[c]
            e_other = r + (t[:, c] - other[:, c]) * mu[c]
            pick_other = e_other.abs() < e_near.abs()
            q[:, c] = torch.where(pick_other, other[:, c], near[:, c])
            r = torch.where(pick_other, e_other, e_near)
        y = conv(xh, wh) + conv(xl8, f8(w * sh) / sh) + conv(xh8, q / sl)
    elif mode.startswith("c6k:"):
        # exactly the c6 kernels' operand model (csrc/xq_conv.hip, k_resblock_c8<.., C6>): bf6 (e3m2) correction operands with
        # FIXED scales -- the image's exponent k (2^k * 28 >= the tensor's calibration maximum; saturating), one power of
        # two per output channel of a filter (the row's largest magnitude in [8, 16))
        k = int(mode[4:])
        xh = x.to(torch.float32).to(torch.float16).to(torch.float64)
        wh = w.to(torch.float32).to(torch.float16).to(torch.float64)
        xl6 = fp6((x - xh) * 2.0 ** (11 - k), "e3m2") * 2.0 ** (k - 11)
        xh6 = fp6(x * 2.0 ** -k, "e3m2") * 2.0 ** k
        rowmax = lambda t: t.abs().amax(dim=(1, 2, 3), keepdim=True).clamp_min(1e-300)      # one shift per output channel
        sh = 2.0 ** (3 - torch.floor(torch.log2(rowmax(w))))
        wl = w - wh
        sl = 2.0 ** (3 - torch.floor(torch.log2(rowmax(wl))))
        y = conv(xh, wh) + conv(xl6, fp6(w * sh, "e3m2") / sh) + conv(xh6, fp6(wl * sl, "e3m2") / sl)
    elif mode.startswith("c6-"):                             # c6-<e2m3|e3m2>-<block|fixed>
        _, fmt, xs = mode.split("-")
        xh, xl6, xh6, wh, wl6, wh6 = c6_operands(x, w, fmt, xs)
        y = conv(xh, wh) + conv(xl6, wh6) + conv(xh6, wl6)
    elif mode == "c8-kernel":
        # exactly the kernels' operand model (csrc/xq_conv.hip): FIXED activation scales -- x_lo8 = e4m3(sat(x_lo * 2^11)),
        # x_hi8 = e4m3(sat(x)), saturation at +-448 -- and one power-of-two scale per OUTPUT CHANNEL of a filter (the row's
        # largest magnitude in [128, 256); per tensor until the end of round 4)
        f8 = lambda t: t.clamp(-448.0, 448.0).to(torch.float32).to(torch.float8_e4m3fn).to(torch.float64)
        xh = x.to(torch.float32).to(torch.float16).to(torch.float64)
        wh = w.to(torch.float32).to(torch.float16).to(torch.float64)
        xl8 = f8((x - xh) * 2048.0) / 2048.0
        xh8 = f8(x)
        rowmax = lambda t: t.abs().amax(dim=(1, 2, 3), keepdim=True).clamp_min(1e-300)
        sh = 2.0 ** (7 - torch.floor(torch.log2(rowmax(w))))
        wl = w - wh
        sl = 2.0 ** (7 - torch.floor(torch.log2(rowmax(wl))))
        y = conv(xh, wh) + conv(xl8, f8(w * sh) / sh) + conv(xh8, f8(wl * sl) / sl)
    else:
        xh = x.to(torch.float32).to(torch.float16).to(torch.float64)
        wh = w.to(torch.float32).to(torch.float16).to(torch.float64)
        y = conv(xh, wh)
        if mode != "f16":
            fmt = "e4m3" if mode == "f16+fp8" else "e2m3"
            xl8 = block_scaled(x - xh, 1, fmt)
            xh8 = block_scaled(x, 1, fmt)
            wl8 = block_scaled(w - wh, 1, fmt)
            wh8 = block_scaled(w, 1, fmt)
            y = y + conv(xl8, wh8) + conv(xh8, wl8)
    return y + b.view(1, -1, 1, 1)


def stored(x, mode):
    """The activation as the NEXT layer / the skip connection reads it back: the kernels keep (hi, lo) pairs."""
    if mode == "f64":
        return x
    if mode == "bf16x3":
        xh = x.to(torch.float32).to(torch.bfloat16).to(torch.float64)
        return xh + (x - xh).to(torch.float32).to(torch.bfloat16).to(torch.float64)
    xh = x.to(torch.float32).to(torch.float16).to(torch.float64)
    if mode == "f16":
        return xh
    if mode in ("f16x3", "f16x3-ftz"):
        lo = (x - xh).to(torch.float32).to(torch.float16).to(torch.float64)
        if mode == "f16x3-ftz":
            lo = torch.where(lo.abs() < 2.0 ** -14, torch.zeros_like(lo), lo)
        return xh + lo
    if mode.startswith("c6k:"):
        k = int(mode[4:])
        return xh + fp6((x - xh) * 2.0 ** (11 - k), "e3m2") * 2.0 ** (k - 11)
    if mode.startswith("c6-"):                               # the lo image doubles as the skip connection's low part
        _, fmt, xs = mode.split("-")
        xh_, xl6, _, _, _, _ = c6_operands(x, torch.zeros(1, x.shape[1], 1, 1, dtype=x.dtype), fmt, xs)
        return xh_ + xl6
    if mode in ("c8-kernel", "c8-ef"):
        lo = ((x - xh) * 2048.0).clamp(-448.0, 448.0).to(torch.float32).to(torch.float8_e4m3fn).to(torch.float64)
        return xh + lo / 2048.0
    return xh + block_scaled(x - xh, 1, "e4m3" if mode == "f16+fp8" else "e2m3")


def run(net, planes, mode, exps=None):
    """exps (mode "c6-kernel"): ([k_mid per block], [k_out per block]), the activation images' exponents (default:
    agent/model.py c6_exponents of this batch's own float64 activation maxima)."""
    from cchess_alphazero.agent.model import _fold
    d = torch.float64
    with torch.no_grad():
        ic = _fold(net.input_conv, net.input_bn)
        x = F.relu(F.conv2d(planes.to(d), ic.weight.to(d), ic.bias.to(d), padding=ic.padding))     # input layer: exact fp32 gather in the engine
        if mode == "c6-kernel":
            # the engine's c6 tower: the fused input layer hands block 0 a c8 image (its first convolution is c8), every
            # other convolution reads a bf6 image with the exponent its producer was given
            if exps is None:
                from cchess_alphazero.agent.model import c6_exponents
                amax, t = [float(x.max())], x
                for blk in net.res:
                    c1, c2 = _fold(blk.conv1, blk.bn1), _fold(blk.conv2, blk.bn2)
                    u = F.relu(F.conv2d(t, c1.weight.to(d), c1.bias.to(d), padding=1))
                    t = F.relu(F.conv2d(u, c2.weight.to(d), c2.bias.to(d), padding=1) + t)
                    amax += [float(u.max()), float(t.max())]
                exps = c6_exponents(amax)
            kmid, kout = exps
            x = stored(x.to(torch.float32).to(d), "c8-kernel")
            for bi, blk in enumerate(net.res):
                c1, c2 = _fold(blk.conv1, blk.bn1), _fold(blk.conv2, blk.bn2)
                m1 = "c8-kernel" if bi == 0 else f"c6k:{kout[bi - 1]}"
                y = F.relu(conv_model(x, c1.weight.to(d), c1.bias.to(d), m1, 1)).to(torch.float32).to(d)
                y = stored(y, f"c6k:{kmid[bi]}")
                z = conv_model(y, c2.weight.to(d), c2.bias.to(d), f"c6k:{kmid[bi]}", 1).to(torch.float32).to(d)
                x = stored(F.relu(z + x).to(torch.float32).to(d), f"c6k:{kout[bi]}")
            mode = "done"
        hybrid = None
        if ">" in mode:                      # "c8-kernel>5": blocks 0..4 on the c8 arithmetic, the rest on (hi, lo) fp16 pairs
            mode, hybrid = mode.split(">")[0], int(mode.split(">")[1])
        if mode != "done":
            x = stored(x.to(torch.float32).to(d), mode)
        for bi, blk in enumerate(net.res if mode != "done" else []):
            if hybrid is not None and bi == hybrid:
                mode = "f16x3"
            c1, c2 = _fold(blk.conv1, blk.bn1), _fold(blk.conv2, blk.bn2)
            y = F.relu(conv_model(x, c1.weight.to(d), c1.bias.to(d), mode, 1)).to(torch.float32).to(d)   # fp32 accumulators
            y = stored(y, mode)
            z = conv_model(y, c2.weight.to(d), c2.bias.to(d), mode, 1).to(torch.float32).to(d)
            x = stored(F.relu(z + x).to(torch.float32).to(d), mode)
        pc, vc = _fold(net.policy_conv, net.policy_bn), _fold(net.value_conv, net.value_bn)
        p = F.relu(F.conv2d(x, pc.weight.to(d), pc.bias.to(d)))
        logits = F.linear(p.flatten(1), net.policy_out.weight.to(d), net.policy_out.bias.to(d))
        v = F.relu(F.conv2d(x, vc.weight.to(d), vc.bias.to(d)))
        v = F.relu(F.linear(v.flatten(1), net.value_dense.weight.to(d), net.value_dense.bias.to(d)))
        vpre = F.linear(v, net.value_out.weight.to(d), net.value_out.bias.to(d)).squeeze(1)
        v = torch.tanh(vpre)
    return x, logits, F.softmax(logits, dim=1), v, vpre


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--filters", type=int, default=128)
    ap.add_argument("--blocks", type=int, default=7)
    ap.add_argument("--positions", type=int, default=24)
    ap.add_argument("--seed", type=int, default=11)
    ap.add_argument("--peaked", type=float, default=1.0, help="scale of the policy layer's weights (sharper softmax)")
    ap.add_argument("--uniform-scales", action="store_true")
    ap.add_argument("--act-scale", type=float, default=1.0,
                    help="multiplies the input layer's filters and bias: activations of the whole tower grow by about this factor")
    ap.add_argument("--modes", default="bf16x3,f16x3,c8-kernel,f16+fp8,f16+fp6,f16",
                    help="comma-separated; also f16x3-ftz, c8-kernel>N (first N blocks c8, the rest f16x3)")
    ap.add_argument("--outliers", default="", help="N,G: N channels of the input layer G times larger (heavy-tailed activations)")
    a = ap.parse_args()
    global UNIFORM
    UNIFORM = a.uniform_scales
    from cchess_alphazero.agent.model import CChessNet
    import oracle.xq_oracle as xo
    torch.manual_seed(a.seed)
    net = CChessNet(cnn_filter_num=a.filters, res_layer_num=a.blocks)
    for m in net.modules():                                  # the perturbed BatchNorm statistics of the GPU numerics tests
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.3)
            m.running_var.uniform_(0.5, 2.0)
            m.weight.data.normal_(1, 0.2)
            m.bias.data.normal_(0, 0.2)
    net.policy_out.weight.data.mul_(a.peaked)
    if a.act_scale != 1.0:                                   # (BatchNorm folded: scale its affine output)
        net.input_bn.weight.data.mul_(a.act_scale)
        net.input_bn.bias.data.mul_(a.act_scale)
    if a.outliers:
        n_out, gain = a.outliers.split(",")
        net.input_bn.weight.data[:int(n_out)].mul_(float(gain))
        net.input_bn.bias.data[:int(n_out)].mul_(float(gain))
    net.eval()
    rng = np.random.default_rng(a.seed)
    boards, state = [], xo.INIT_STATE
    while len(boards) < a.positions:                         # a random playout's positions
        mv = xo.get_legal_moves(state)
        if not mv or xo.done(state)[0]:
            state = xo.INIT_STATE
            continue
        boards.append(xo.state_to_board(state))
        state = xo.step(state, mv[rng.integers(len(mv))])
    planes = torch.from_numpy(np.stack([xo.planes_board(b) for b in boards]))
    ref = run(net, planes, "f64")
    out = {"uniform_scales": UNIFORM, "act_scale": a.act_scale, "trunk_max_activation": float(ref[0].abs().max()),
           "filters": a.filters, "blocks": a.blocks, "positions": a.positions, "policy_scale": a.peaked,
           "max_policy_probability": float(ref[2].max()), "value_range": [float(ref[3].min()), float(ref[3].max())],
           "value_preactivation_range": [float(ref[4].min()), float(ref[4].max())], "modes": {}}
    for mode in a.modes.split(","):
        x, lg, p, v, vpre = run(net, planes, mode)
        c = lambda t: t - t.mean(1, keepdim=True)
        out["modes"][mode] = {
            "trunk_rel_err": float((x - ref[0]).norm() / ref[0].norm()),
            "logit_max_abs": float((c(lg) - c(ref[1])).abs().max()),
            "policy_max_abs": float((p - ref[2]).abs().max()),
            "value_max_abs": float((v - ref[3]).abs().max()),
            "value_preactivation_max_abs": float((vpre - ref[4]).abs().max())}
        print(mode, out["modes"][mode], flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
