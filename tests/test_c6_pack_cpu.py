"""The c6 filter pack (csrc/xq_conv.hip cz_conv3x3_c6_pack_weights; host code, no GPU): bf6 (e3m2) rounding, the order of
the 32 channels inside a 24-byte piece (the order the kernels' conversion instruction gives the activations), where heads
and tails sit, the trailing shifts and image exponents."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chinesechess-alphazero_amd"))

W_PAD_STEPS = 3


def bf6_value(code):
    s, e, m = code >> 5, (code >> 2) & 7, code & 3
    v = (1.0 + m / 4.0) * 2.0 ** (e - 3) if e else (m / 4.0) * 2.0 ** -2
    return -v if s else v


GRID = np.array([bf6_value(c) for c in range(32)])


def bf6_round(x):
    """nearest grid value, ties to the even code, saturating at 28 (numpy reference of f32_to_bf6_bits)."""
    a = np.minimum(np.abs(x), 28.0)
    d = np.abs(a[..., None] - GRID)
    best = d.argmin(-1)
    # ties: argmin takes the lower code; the even one may be the upper
    tie = np.isclose(np.take_along_axis(d, best[..., None], -1)[..., 0],
                     np.take_along_axis(d, np.minimum(best + 1, 31)[..., None], -1)[..., 0], rtol=0, atol=0)
    best = np.where(tie & (best % 2 == 1), best + 1, best)
    return np.sign(x) * GRID[best]


def channel_of(e):
    return 8 * (e >> 3) + ((e >> 1) & 3) + 4 * (e & 1)


def test_c6_weight_pack_layout_and_bf6_rounding():
    import torch
    from cchess_alphazero import _native
    torch.manual_seed(5)
    C, KK, CT, NB = 128, 8, 4, 2
    w = torch.randn(C, C, 3, 3) * 0.05
    w[3, 5, 1, 1] = 0.9                                          # the largest magnitude: fixes the shifts
    pk = _native.pack_conv3x3_c6_weights(w, -2, 3).numpy()
    main_u4 = (9 * KK + W_PAD_STEPS) * CT * 64
    c8_u4 = (9 * NB + 1) * 2 * CT * 2 * 64
    assert pk.size == (main_u4 + c8_u4 + 1) * 16 + 2 * C
    tail = pk[(main_u4 + c8_u4) * 16:(main_u4 + c8_u4 + 1) * 16].view(np.int32)
    _, _, x_exp, y_exp = (int(v) for v in tail)
    assert (x_exp, y_exp) == (-2, 3)
    rows = pk[(main_u4 + c8_u4 + 1) * 16:].view(np.int8).astype(np.int64)
    sh, sl = rows[:C], rows[C:]                                  # one shift per OUTPUT CHANNEL and kind
    wn = w.numpy().astype(np.float64)
    wh = w.half().float().numpy().astype(np.float64)
    rmax, lmax = np.abs(wn).reshape(C, -1).max(1), np.abs(wn - wh).reshape(C, -1).max(1)
    assert ((rmax * 2.0 ** sh >= 8) & (rmax * 2.0 ** sh < 16) & (lmax * 2.0 ** sl >= 8) & (lmax * 2.0 ** sl < 16)).all()
    assert len(set(sh.tolist())) > 1                             # (row 3 holds the 0.9: its shift differs from the others')
    # the fp16 fragments are the c8 pack's
    ref8 = _native.pack_conv3x3_c8_weights(w).numpy()
    assert (pk[:main_u4 * 16] == ref8[:main_u4 * 16]).all()
    c6 = pk[main_u4 * 16:(main_u4 + c8_u4) * 16]
    rng = np.random.default_rng(0)
    for _ in range(300):
        tap, ct, lane, b, q = (int(rng.integers(n)) for n in (9, CT, 64, NB, 2))
        grp = ((((tap * NB + b) * 2 + q) * CT + ct) * 2) * 64 * 16
        piece = np.concatenate([c6[grp + lane * 16: grp + lane * 16 + 16], c6[grp + 1024 + lane * 8: grp + 1024 + lane * 8 + 8]])
        bits = int.from_bytes(piece.tobytes(), "little")
        o, ky, kx = ct * 32 + (lane & 31), tap // 3, tap % 3
        for e in range(32):
            code = (bits >> (6 * e)) & 63
            c = b * 64 + (lane >> 5) * 32 + channel_of(e)
            src = wn[o, c, ky, kx] * 2.0 ** sh[o] if q == 0 else (wn[o, c, ky, kx] - wh[o, c, ky, kx]) * 2.0 ** sl[o]
            want = float(bf6_round(np.array([src]))[0])
            assert bf6_value(code) == want, (tap, ct, lane, b, q, e, src, bf6_value(code), want)
    # the tails of a group sit densely behind its heads: bytes [1024 + 512, 2048) of a group stay zero
    g0 = c6[:2048]
    assert not g0[1024 + 512:].any() and g0[:1024].any() and g0[1024:1536].any()


def test_bf6_conversion_on_a_sweep():
    """Every grid point, every midpoint (ties to even), saturation and subnormals, through the packer's converter: a filter
    whose (o = 0, tap 0) row holds the sweep."""
    import torch
    from cchess_alphazero import _native
    vals = sorted(set([float(g) for g in GRID] + [float((GRID[i] + GRID[i + 1]) / 2) for i in range(31)] +
                      [0.01, 0.03, 0.031, 0.0313, 0.09, 27.0, 27.9]))
    vals = vals[:96]                                             # (three 32-blocks of channels 0 .. 127 are enough)
    w = torch.zeros(128, 128, 3, 3)
    w[0, :len(vals), 0, 0] = torch.tensor(vals) / 2.0            # largest 14 -> shift 0 would put 27.9 / 2 < 16: sh = 0
    w[0, 127, 0, 0] = -15.0                                      # fixes row 0's shift at 0 ([8, 16))
    pk = _native.pack_conv3x3_c6_weights(w, 0, 0).numpy()
    main_u4 = (9 * 8 + W_PAD_STEPS) * 4 * 64
    sh = int(pk[(main_u4 + (9 * 2 + 1) * 2 * 4 * 2 * 64 + 1) * 16:].view(np.int8)[0])
    assert sh == 0
    c6 = pk[main_u4 * 16:]
    got = {}
    for b in range(2):
        for half in range(2):
            lane = half * 32                                      # row o = 0
            grp = ((((0 * 2 + b) * 2 + 0) * 4 + 0) * 2) * 64 * 16
            piece = np.concatenate([c6[grp + lane * 16: grp + lane * 16 + 16], c6[grp + 1024 + lane * 8: grp + 1024 + lane * 8 + 8]])
            bits = int.from_bytes(piece.tobytes(), "little")
            for e in range(32):
                got[b * 64 + half * 32 + channel_of(e)] = bf6_value((bits >> (6 * e)) & 63)
    src = w[0, :, 0, 0].numpy().astype(np.float64)
    want = bf6_round(src)
    for c in range(128):
        assert got[c] == want[c], (c, src[c], got[c], want[c])
