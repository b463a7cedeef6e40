"""Host side of the c8 tower arithmetic (csrc/xq_conv.hip, cz_conv3x3_c8_pack_weights): the fragment layout and the
float -> e4m3 conversion, decoded back with PyTorch's own float8_e4m3fn type.  No GPU involved."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chinesechess-alphazero_amd"))


def decode_c8_pack(packed, c=128):
    """-> (w_hi [o, c, 9] float32 from the f16 fragments, w_k0 / w_k1 [o, c, 9] float32 from the e4m3 fragments (still
    scaled), sh [o], sl [o]: the per-output-channel shifts): the inverse of the layout cz_conv3x3_c8_pack_weights documents."""
    import torch
    kk_n, ct_n, nb = c // 16, c // 32, c // 64
    main_u4 = (9 * kk_n + 3) * ct_n * 64
    c8_u4 = (9 * nb + 1) * 2 * ct_n * 2 * 64
    raw = packed.numpy()
    hi = torch.from_numpy(raw[:main_u4 * 16].copy()).view(torch.float16).float().numpy().reshape(-1, ct_n, 64, 8)
    f8 = torch.from_numpy(raw[main_u4 * 16:(main_u4 + c8_u4) * 16].copy()).view(torch.float8_e4m3fn).float().numpy()
    f8 = f8.reshape(9 * nb + 1, 2, ct_n, 2, 64, 16)
    sc = raw[(main_u4 + c8_u4) * 16:(main_u4 + c8_u4 + 1) * 16].view(np.int32)
    rows = raw[(main_u4 + c8_u4 + 1) * 16:].view(np.int8).astype(np.int64)
    assert rows.size == 2 * c
    sh, sl = rows[:c], rows[c:]
    assert int(sc[0]) <= int(sh.max()) and int(sc[1]) <= int(sl.max())        # (ints 0 / 1: the smallest row shifts, informational)
    w_hi = np.zeros((c, c, 9), np.float32)
    w_k = np.zeros((2, c, c, 9), np.float32)
    for tap in range(9):
        for ct in range(ct_n):
            for lane in range(64):
                o = ct * 32 + (lane & 31)
                for kk in range(kk_n):
                    w_hi[o, kk * 16 + (lane >> 5) * 8:kk * 16 + (lane >> 5) * 8 + 8, tap] = hi[tap * kk_n + kk, ct, lane]
                for b in range(nb):
                    for q in range(2):
                        c0 = b * 64 + (lane >> 5) * 32
                        w_k[q, o, c0:c0 + 32, tap] = f8[tap * nb + b, q, ct, :, lane, :].reshape(32)
    assert not f8[9 * nb].any() and not hi[9 * kk_n:].any()            # the prefetch padding is zero
    return w_hi, w_k[0], w_k[1], sh, sl


def test_c8_weight_pack_layout_and_e4m3_rounding():
    import torch
    from cchess_alphazero import _native
    torch.manual_seed(3)
    w = torch.randn(128, 128, 3, 3) * 0.05
    w[0, 0, 0, 0] = 0.31                                   # the largest magnitude: fixes the scale
    w[5, 7, 1, 1] = 0.0
    w[6, 7, 1, 1] = -1e-7                                  # flushes to -0 in the e4m3 image
    packed = _native.pack_conv3x3_c8_weights(w)
    w_hi, k0, k1, sh, sl = decode_c8_pack(packed)
    w3 = w.reshape(128, 128, 9)
    assert np.array_equal(w_hi, w3.half().float().numpy())
    lo = w3 - w3.half().float()
    # one shift per OUTPUT CHANNEL and kind: the row's largest magnitude lands in [128, 256)
    f = lambda a: torch.from_numpy(2.0 ** a.astype(np.float64)).float().view(128, 1, 1)
    rmax, lmax = w3.abs().amax((1, 2)), lo.abs().amax((1, 2))
    assert ((rmax * f(sh).view(-1) >= 128) & (rmax * f(sh).view(-1) < 256)).all()
    assert ((lmax * f(sl).view(-1) >= 128) & (lmax * f(sl).view(-1) < 256)).all()
    want0 = (w3 * f(sh)).to(torch.float8_e4m3fn).float().numpy()
    want1 = (lo * f(sl)).to(torch.float8_e4m3fn).float().numpy()
    assert np.array_equal(k0, want0)
    assert np.array_equal(k1, want1)
    # what the three terms reconstruct: w to 2^-16-ish of the ROW's largest weight
    rec = w_hi + k1 / f(sl).numpy()
    assert (np.abs(rec - w3.numpy()).reshape(128, -1).max(1) <= 2.0 ** -4 * lmax.numpy() + 1e-12).all()


def test_e4m3_conversion_matches_torch_on_a_sweep():
    """Every e4m3 value, the midpoints between neighbours (ties to even), values just off the midpoints, saturation."""
    import torch
    from cchess_alphazero import _native
    grid = torch.arange(256, dtype=torch.uint8).view(torch.float8_e4m3fn).float()
    grid = grid[torch.isfinite(grid)].unique()
    mids = (grid[1:] + grid[:-1]) / 2
    vals = torch.cat([grid, mids, mids * (1 + 1e-6), mids * (1 - 1e-6), torch.tensor([447.9, 448.0, 460.0, 1e-4, 9.7e-4, 1e-3])])
    vals = torch.cat([vals, -vals])[:128 * 128 * 9]
    w = torch.zeros(128 * 128 * 9)
    w[:vals.numel()] = vals / 2.0                          # the pack scales by 2^sh with max |w| * 2^sh in [128, 256): 224 -> sh = 0 ... use /2, sh = 1
    w[-1] = 224.0 / 2.0 * 1.0                              # pins sh = 1 (largest magnitude 230 / 2 -> [128, 256) after * 2)
    w = w.reshape(128, 128, 3, 3)
    packed = _native.pack_conv3x3_c8_weights(w)
    _, k0, _, sh, _ = decode_c8_pack(packed)
    # (per-row shifts since the end of round 4: every row's own largest magnitude lands in [128, 256))
    scaled = (w.reshape(128, 128, 9) * torch.from_numpy(2.0 ** sh.astype(np.float64)).float().view(128, 1, 1)).clamp(-448, 448)
    assert np.array_equal(k0, scaled.to(torch.float8_e4m3fn).float().numpy())
