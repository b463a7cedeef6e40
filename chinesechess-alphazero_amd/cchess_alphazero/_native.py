"""ctypes binding of libczero.so (C-ABI in include/czero.h).

There is NO CPU fallback: if the HIP library is missing or no GPU is visible, every compute
entry point raises.  Device memory, streams and process groups come from PyTorch (plumbing).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CZ_LIB") or os.path.join(os.path.dirname(_HERE), "csrc", "libczero.so")   # CZ_LIB: A/B builds (tools/ab_search.sh)

NSQ, NLABELS, MAXMOVES, NOMOVE = 90, 2086, 128, 0xFFFF
F32, F16, BF16, U8, F16C8, F16C6, F16C86 = 0, 1, 2, 3, 4, 5, 6

_lib = None


class NativeError(RuntimeError):
    pass


def lib():
    """Load libczero.so; raise loudly when it has not been built."""
    global _lib
    if _lib is None:
        # PyTorch ships its own HIP runtime: it must be loaded first so that libczero.so binds to the same
        # one (loading /opt/rocm's runtime first leaves torch without a visible GPU).
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                f"{LIB_PATH} not found: build the HIP engine first "
                f"(python chinesechess-alphazero_amd/build.py); there is no CPU fallback")
        _lib = C.CDLL(LIB_PATH)
        _declare(_lib)
        if _lib.cz_version() != 2:
            raise NativeError("libczero.so version mismatch")
    return _lib


def _declare(L):
    vp, i32 = C.c_void_p, C.c_int
    L.cz_version.restype = i32
    L.cz_last_error.restype = C.c_char_p
    L.cz_device_count.restype = i32
    L.cz_label_tables.argtypes = [vp, vp]
    L.cz_movegen.argtypes = [vp, i32, vp, vp, vp]
    L.cz_done.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp]
    L.cz_step.argtypes = [vp, vp, i32, vp, vp, vp]
    L.cz_encode.argtypes = [vp, i32, vp, i32, vp]
    L.cz_check_or_catch.argtypes = [vp, vp, i32, vp, vp]
    L.cz_be_catched.argtypes = [vp, vp, i32, vp, vp]
    L.cz_has_attack.argtypes = [vp, i32, vp, vp]
    L.cz_rules_fused.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp]
    if hasattr(L, "cz_bias_act"):
        L.cz_bias_act.argtypes = [vp, vp, vp, C.c_size_t, i32, i32, i32, vp]
        L.cz_bias_act.restype = i32
    if hasattr(L, "cz_conv3x3"):
        L.cz_conv3x3.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
        L.cz_conv3x3.restype = i32
        L.cz_conv3x3_packed_elems.argtypes = [i32, i32]
        L.cz_conv3x3_packed_elems.restype = C.c_size_t
        L.cz_conv3x3_pack_weights.argtypes = [vp, i32, i32, i32, vp]
        L.cz_conv3x3_pack_weights.restype = i32
        L.cz_head_convs.argtypes = [vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, vp]
        L.cz_head_convs.restype = i32
        L.cz_input_conv.argtypes = [vp, i32, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
        L.cz_input_conv.restype = i32
        L.cz_input_conv_packed_elems.argtypes = [i32, i32, i32]
        L.cz_input_conv_packed_elems.restype = C.c_size_t
        L.cz_input_conv_pack_weights.argtypes = [vp, i32, i32, i32, i32, vp]
        L.cz_input_conv_pack_weights.restype = i32
        L.cz_resblock_heads.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
        L.cz_resblock_heads.restype = i32
        L.cz_resblock.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
        L.cz_input_conv_q.argtypes = [vp, i32, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp]
        L.cz_resblock_q.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]
        L.cz_resblock_heads_q.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]
        for _n in ("cz_input_conv_q", "cz_resblock_q", "cz_resblock_heads_q"):
            getattr(L, _n).restype = i32
        L.cz_resblock.restype = i32
        L.cz_split_bias_act.argtypes = [vp, vp, vp, vp, C.c_size_t, i32, i32, i32, i32, vp]
        L.cz_split_bias_act.restype = i32
    if hasattr(L, "cz_conv3x3_c8"):
        L.cz_conv3x3_c8_packed_bytes.argtypes = [i32]
        L.cz_conv3x3_c8_packed_bytes.restype = C.c_size_t
        L.cz_conv3x3_c8_pack_weights.argtypes = [vp, i32, vp]
        L.cz_conv3x3_c8_pack_weights.restype = i32
        L.cz_conv3x3_c6_pack_weights.argtypes = [vp, i32, i32, i32, vp]
        L.cz_conv3x3_c6_pack_weights.restype = i32
        L.cz_conv3x3_c8.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]
        L.cz_conv3x3_c8.restype = i32
    if hasattr(L, "cz_input_resblock"):
        L.cz_input_resblock.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp]
        L.cz_input_resblock.restype = i32
    if hasattr(L, "cz_tower"):
        L.cz_input_resblock_m.restype = i32
        L.cz_input_resblock_m.argtypes = [vp, vp, i32] + [vp] * 8 + [i32] * 3 + [vp] * 3
        L.cz_tower.restype = i32
        L.cz_tower.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp]
        L.cz_resblock_chain.restype = i32
        L.cz_resblock_chain.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp]
        L.cz_tower_plain.restype = i32
        L.cz_tower_plain.argtypes = [vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp]
        L.cz_tower_pairs.restype = i32
        L.cz_tower_pairs.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]
        L.cz_tower_c6.restype = i32
        L.cz_tower_c6.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp]
        L.cz_tower_c6_heads.restype = i32
        L.cz_tower_c6_heads.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp]
    if hasattr(L, "cz_heads_tail"):
        L.cz_heads_tail.argtypes = [vp, i32, vp, vp, i32, vp, i32, vp, vp, i32, vp, C.c_float, vp, vp, vp, i32, i32, i32, vp, vp]
        L.cz_heads_tail.restype = i32
        L.cz_fc_packed_elems.argtypes = [i32, i32]
        L.cz_fc_packed_elems.restype = C.c_size_t
        L.cz_fc_pack_weights.argtypes = [vp, i32, i32, i32, vp]
        L.cz_fc_pack_weights.restype = i32
    for name in ("cz_label_tables", "cz_movegen", "cz_done", "cz_step", "cz_encode", "cz_check_or_catch",
                 "cz_be_catched", "cz_has_attack", "cz_rules_fused"):
        getattr(L, name).restype = i32
    if hasattr(L, "cz_search_create"):
        from . import _native_search
        _native_search.declare(L)


def check(rc, what=""):
    if rc != 0:
        raise NativeError(f"{what} failed ({rc}): {lib().cz_last_error().decode()}")


def require_gpu():
    import torch
    if not torch.cuda.is_available() or lib().cz_device_count() < 1:
        raise NativeError("no MI355X visible: the engine has no CPU path")


_tables = None


def label_tables():
    """(label_of[90,90] uint16, from[2086] uint8, to[2086] uint8) -- host copies of the engine tables."""
    global _tables
    if _tables is None:
        lo = np.zeros(NSQ * NSQ, dtype=np.uint16)
        ft = np.zeros(NLABELS, dtype=np.uint16)
        check(lib().cz_label_tables(lo.ctypes.data, ft.ctypes.data), "cz_label_tables")
        _tables = (lo.reshape(NSQ, NSQ), (ft >> 8).astype(np.uint8), (ft & 0xFF).astype(np.uint8))
    return _tables


def _stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, dtype):
    import torch
    assert isinstance(t, torch.Tensor) and t.is_cuda and t.is_contiguous() and t.dtype == dtype, \
        (t.dtype, dtype, t.device)
    return C.c_void_p(t.data_ptr())


_TORCH_DT = None


def torch_dtype(code):
    import torch
    return {F32: torch.float32, F16: torch.float16, BF16: torch.bfloat16, U8: torch.uint8}[code]


# ---- batched rules on device tensors (boards: int8 [n, 90] on cuda) -------------------------
def movegen(boards):
    import torch
    require_gpu()
    n = boards.shape[0]
    moves = torch.empty((n, MAXMOVES), dtype=torch.uint16, device=boards.device)
    counts = torch.empty((n,), dtype=torch.uint8, device=boards.device)
    check(lib().cz_movegen(_dev(boards, torch.int8), n, _dev(moves, torch.uint16), _dev(counts, torch.uint8),
                           _stream()), "cz_movegen")
    return moves, counts


def done(boards, need_check=False):
    import torch
    require_gpu()
    n = boards.shape[0]
    dev = boards.device
    over = torch.empty((n,), dtype=torch.int8, device=dev)
    v = torch.empty((n,), dtype=torch.int8, device=dev)
    fm = torch.empty((n,), dtype=torch.uint16, device=dev)
    ck = torch.zeros((n,), dtype=torch.uint8, device=dev)
    check(lib().cz_done(_dev(boards, torch.int8), n, int(need_check), _dev(over, torch.int8), _dev(v, torch.int8),
                        _dev(fm, torch.uint16), _dev(ck, torch.uint8), _stream()), "cz_done")
    return over, v, fm, ck


def step(boards, moves):
    import torch
    require_gpu()
    n = boards.shape[0]
    out = torch.empty_like(boards)
    ne = torch.empty((n,), dtype=torch.uint8, device=boards.device)
    check(lib().cz_step(_dev(boards, torch.int8), _dev(moves, torch.uint16), n, _dev(out, torch.int8),
                        _dev(ne, torch.uint8), _stream()), "cz_step")
    return out, ne


def encode(boards, dtype=F32):
    import torch
    require_gpu()
    n = boards.shape[0]
    planes = torch.empty((n, 14, 10, 9), dtype=torch_dtype(dtype), device=boards.device)
    check(lib().cz_encode(_dev(boards, torch.int8), n, C.c_void_p(planes.data_ptr()), dtype, _stream()), "cz_encode")
    return planes


def check_or_catch(boards, moves):
    import torch
    require_gpu()
    n = boards.shape[0]
    out = torch.empty((n,), dtype=torch.uint8, device=boards.device)
    check(lib().cz_check_or_catch(_dev(boards, torch.int8), _dev(moves, torch.uint16), n, _dev(out, torch.uint8),
                                  _stream()), "cz_check_or_catch")
    return out


def be_catched(boards, moves):
    import torch
    require_gpu()
    n = boards.shape[0]
    out = torch.empty((n,), dtype=torch.uint8, device=boards.device)
    check(lib().cz_be_catched(_dev(boards, torch.int8), _dev(moves, torch.uint16), n, _dev(out, torch.uint8),
                              _stream()), "cz_be_catched")
    return out


def has_attack(boards):
    import torch
    require_gpu()
    n = boards.shape[0]
    out = torch.empty((n,), dtype=torch.uint8, device=boards.device)
    check(lib().cz_has_attack(_dev(boards, torch.int8), n, _dev(out, torch.uint8), _stream()), "cz_has_attack")
    return out


_DT_CODE = None


def bias_act_(x, bias, residual=None, relu=True):
    """In place: x = relu(x + bias[c] (+ residual)) for a channels-last activation (logical NCHW tensor whose
    memory is NHWC, or any [..., C] contiguous tensor).  One HBM pass (csrc/xq_nn_epilogue.hip)."""
    import torch
    global _DT_CODE
    if _DT_CODE is None:
        _DT_CODE = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}
    c = bias.numel()
    check(lib().cz_bias_act(C.c_void_p(x.data_ptr()), C.c_void_p(bias.data_ptr()),
                            C.c_void_p(residual.data_ptr()) if residual is not None else None,
                            x.numel(), c, _DT_CODE[x.dtype], int(relu), _stream()), "cz_bias_act")
    return x


def _pair_code(x):
    """dtype code of an operand tuple: (f16, uint8 c8 image) is the c8 arithmetic's pair (CZ_F16C8), (f16, int8 image of
    the same size) the c6 arithmetic's (CZ_F16C6: bf6 pieces in the image; the element type is the tag)."""
    import torch
    if len(x) == 2 and x[1].dtype in (torch.uint8, torch.int8):
        assert x[0].dtype == torch.float16 and x[1].shape[-1] == 2 * x[0].shape[-1]
        return F16C8 if x[1].dtype == torch.uint8 else F16C6
    return _dt_code(x[0].dtype)


def _dt_code(dtype):
    import torch
    global _DT_CODE
    if _DT_CODE is None:
        _DT_CODE = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}
    return _DT_CODE[dtype]


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def pack_conv3x3_weights(w_oihw, dtype, parts):
    """fp32 [C, C, 3, 3] filter (any device) -> packed 2-byte tensor in MFMA fragment order, on the CPU
    (cz_conv3x3_pack_weights runs on the host); move it to the GPU with .cuda()."""
    import torch
    w = w_oihw.detach().to("cpu", torch.float32).contiguous()
    c = w.shape[0]
    assert tuple(w.shape) == (c, c, 3, 3)
    n = lib().cz_conv3x3_packed_elems(c, parts)
    if n == 0:
        raise NativeError(f"cz_conv3x3: unsupported channels={c} parts={parts}")
    out = torch.empty((n,), dtype=dtype)
    check(lib().cz_conv3x3_pack_weights(_ptr(w), c, _dt_code(dtype), parts, _ptr(out)), "cz_conv3x3_pack_weights")
    return out


# ---- the c8 tower arithmetic: fp16 main term + two scaled-fp8 correction terms (csrc/xq_conv.hip, k_conv3x3_c8) ----------
C8_X_LO_SHIFT = 11


def pack_conv3x3_c8_weights(w_oihw):
    """fp32 [128, 128, 3, 3] filter -> packed bytes (f16 fragments, e4m3 correction fragments, the two scale exponents)."""
    import torch
    w = w_oihw.detach().to("cpu", torch.float32).contiguous()
    c = w.shape[0]
    n = lib().cz_conv3x3_c8_packed_bytes(c)
    if n == 0:
        raise NativeError(f"cz_conv3x3_c8: unsupported channels={c}")
    out = torch.empty((n,), dtype=torch.uint8)
    check(lib().cz_conv3x3_c8_pack_weights(_ptr(w), c, _ptr(out)), "cz_conv3x3_c8_pack_weights")
    return out


def pack_conv3x3_c6_weights(w_oihw, x_exp, y_exp):
    """fp32 [C, C, 3, 3] filter (C = 128 or 192) -> packed bytes for the c6 arithmetic (f16 fragments, bf6 correction pieces, the two
    filter shifts, and the exponents of the activation images the convolution reads / writes: x_hi6 = bf6(x 2^-exp))."""
    import torch
    w = w_oihw.detach().to("cpu", torch.float32).contiguous()
    c = w.shape[0]
    n = lib().cz_conv3x3_c8_packed_bytes(c)
    if n == 0 or c not in (128, 192):
        raise NativeError(f"cz_conv3x3_c6: unsupported channels={c}")
    out = torch.empty((n,), dtype=torch.uint8)
    check(lib().cz_conv3x3_c6_pack_weights(_ptr(w), c, int(x_exp), int(y_exp), _ptr(out)), "cz_conv3x3_c6_pack_weights")
    return out


def split_c8(x):
    """fp32 [N, 90, C] -> (x_hi f16 [N, 90, C], x_c8 uint8 [N, 90, 2C]): the operand pair of cz_conv3x3_c8
    (x_c8 = e4m3(x_lo * 2^11) for the C channels, then e4m3(x) for them; saturating at 448)."""
    import torch
    hi = x.to(torch.float16)
    lo = (x - hi.float()) * float(2 ** C8_X_LO_SHIFT)
    l8 = lo.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
    h8 = x.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
    return hi.contiguous(), torch.cat([l8, h8], dim=-1).contiguous()


def conv3x3_c8(x, w_packed, bias, skip=None, out=None, out_f32=None, relu=True):
    """x, skip, out: pairs (hi f16 [N, 90, C], c8 uint8 [N, 90, 2C]); out_f32: fp32 [N, 90, C] instead of `out`."""
    require_gpu()
    n, c = x[0].shape[0], x[0].shape[-1]
    assert x[1].shape[-1] == 2 * c
    sh, sc = (skip[0], skip[1]) if skip is not None else (None, None)
    yh, yc = (out[0], out[1]) if out is not None else (None, None)
    check(lib().cz_conv3x3_c8(_ptr(x[0]), _ptr(x[1]), _ptr(w_packed), _ptr(bias), _ptr(sh), _ptr(sc), _ptr(yh), _ptr(yc),
                              _ptr(out_f32), n, c, int(relu), _stream()), "cz_conv3x3_c8")
    return out_f32 if out_f32 is not None else out


def join_c8(pair):
    """The fp32 value an operand pair stands for: hi + e4m3(lo) * 2^-11."""
    import torch
    c = pair[0].shape[-1]
    l8 = pair[1][..., :c].contiguous().view(torch.float8_e4m3fn).float()
    return pair[0].float() + l8 * float(2.0 ** -C8_X_LO_SHIFT)


def conv3x3(x, w_packed, bias, skip=None, out=None, out_f32=None, relu=True):
    """Trunk convolution on the hand-written MFMA kernel (csrc/xq_conv.hip).
    x, skip, out: tuples (hi,) or (hi, lo) of [N, 90, C] bf16/fp16 tensors; bias fp32 [C];
    out_f32: fp32 [N, 90, C] tensor to receive the result instead of `out`."""
    require_gpu()
    parts = len(x)
    xh = x[0]
    n, c = xh.shape[0], xh.shape[-1]
    xl = x[1] if parts == 2 else None
    sh = skip[0] if skip is not None else None
    sl = skip[1] if skip is not None and parts == 2 else None
    yh = out[0] if out is not None else None
    yl = out[1] if out is not None and parts == 2 else None
    check(lib().cz_conv3x3(_ptr(xh), _ptr(xl), _ptr(w_packed), _ptr(bias), _ptr(sh), _ptr(sl), _ptr(yh), _ptr(yl),
                           _ptr(out_f32), n, c, _dt_code(xh.dtype), parts, int(relu), _stream()), "cz_conv3x3")
    return out_f32 if out_f32 is not None else out


def head_convs(x, w, bias, n_policy, policy_feat, value_feat):
    """x [N, 90, C] trunk output -> relu(1x1 convs + bias): policy_feat [N, n_policy*90], value_feat [N, n_value*90]
    (fp32, channels-first Flatten order).  w [n_policy + n_value, C] fp32."""
    require_gpu()
    n, c = x.shape[0], x.shape[-1]
    check(lib().cz_head_convs(_ptr(x), _dt_code(x.dtype), _ptr(w), _ptr(bias), _ptr(policy_feat), _ptr(value_feat),
                              n, c, n_policy, w.shape[0] - n_policy, _stream()), "cz_head_convs")
    return policy_feat, value_feat


def input_table(w_oihw):
    """fp32 [C, in_planes, 5, 5] input filter (BatchNorm folded) -> the gather table of cz_input_resblock:
    table[c][ky * 5 + kx][o] = w[o][c][ky][kx], fp32 [in_planes, 25, C]."""
    import torch
    w = w_oihw.detach().to(torch.float32)
    return w.permute(1, 2, 3, 0).reshape(w.shape[1], 25, w.shape[0]).contiguous()


def input_resblock(planes, table, in_bias, w1, b1, w2, b2, out, rows=None, count=None, masks=None):
    """cz_input_resblock(_m): planes [N, in_planes, 10, 9] uint8 -> input layer + first residual block -> out = (hi, lo)
    [N, 90, 128] operand pair, or the c8 pair (f16 [N, 90, 128], uint8 [N, 90, 256]) with cz_conv3x3_c8_pack_weights
    filters.  rows / count: the compact evaluation queue (int32 device tensors).  masks [N, 96] int32: the positions'
    occupancy boards (Search.leaf_masks): the kernel then skips deriving them from the planes."""
    require_gpu()
    import torch
    if planes.dtype != torch.uint8:
        raise NativeError("cz_input_resblock reads uint8 planes")
    n = planes.shape[0]
    if masks is not None:
        assert masks.dtype == torch.int32 and masks.is_cuda and masks.is_contiguous() and tuple(masks.shape) == (n, 96)
    L = lib()
    check(L.cz_input_resblock_m(_ptr(planes), _ptr(masks), planes.shape[1], _ptr(table), _ptr(in_bias), _ptr(w1), _ptr(b1),
                                _ptr(w2), _ptr(b2), _ptr(out[0]), _ptr(out[1]), n, out[0].shape[-1], _pair_code(out),
                                _ptr(rows) if rows is not None else None, _ptr(count) if count is not None else None,
                                _stream()), "cz_input_resblock")
    return out


def pack_fc_weights(w, dtype=None):
    """fp32 [n_out, n_in] dense-layer matrix -> (hi, lo) pairs of `dtype` (fp16, the default, or bf16) in MFMA fragment
    order (on the CPU)."""
    import torch
    dtype = dtype or torch.float16
    w = w.detach().to("cpu", torch.float32).contiguous()
    n = lib().cz_fc_packed_elems(w.shape[0], w.shape[1])
    if n == 0:
        raise NativeError(f"cz_fc_pack_weights: unsupported shape {tuple(w.shape)}")
    out = torch.empty((n,), dtype=dtype)
    check(lib().cz_fc_pack_weights(_ptr(w), w.shape[0], w.shape[1], _dt_code(dtype), _ptr(out)), "cz_fc_pack_weights")
    return out


def heads_tail(policy_feat, value_feat, wp, bias_p, w1, bias1, w2, b2, policy, value, stats, count=None, normalize=True):
    """softmax(policy_feat @ Wp^T + bp) -> policy [N, n_labels]; tanh(relu(value_feat @ W1^T + b1) @ w2 + b2) -> value [N]
    (cz_heads_tail; wp / w1 from pack_fc_weights -- both of the same pair dtype --, everything else fp32 on the device).  count: optional int32 device
    tensor, only the first min(count, N) rows are computed (compact evaluation queue).  normalize=False: `policy` keeps the
    raw logits (for a search in cz_search_policy_logits mode)."""
    require_gpu()
    n = policy_feat.shape[0]
    check(lib().cz_heads_tail(_ptr(policy_feat), policy_feat.shape[1], _ptr(wp), _ptr(bias_p), policy.shape[1],
                              _ptr(value_feat), value_feat.shape[1], _ptr(w1), _ptr(bias1), bias1.shape[0], _ptr(w2),
                              float(b2), _ptr(policy), _ptr(value), _ptr(stats), n, _dt_code(wp.dtype), int(bool(normalize)),
                              _ptr(count) if count is not None else None, _stream()), "cz_heads_tail")
    return policy, value


def pack_input_conv_weights(w_oihw, dtype, parts):
    """fp32 [C, in_planes, 5, 5] filter -> packed 2-byte tensor in MFMA fragment order (on the CPU)."""
    import torch
    w = w_oihw.detach().to("cpu", torch.float32).contiguous()
    c, ip = w.shape[0], w.shape[1]
    assert tuple(w.shape[2:]) == (5, 5)
    n = lib().cz_input_conv_packed_elems(c, ip, parts)
    if n == 0:
        raise NativeError(f"cz_input_conv: unsupported channels={c} in_planes={ip} parts={parts}")
    out = torch.empty((n,), dtype=dtype)
    check(lib().cz_input_conv_pack_weights(_ptr(w), c, ip, _dt_code(dtype), parts, _ptr(out)),
          "cz_input_conv_pack_weights")
    return out


def input_conv(planes, w_packed, bias, out, relu=True, rows=None, count=None):
    """planes [N, in_planes, 10, 9] (fp32 / fp16 / bf16 / uint8, contiguous, as the search kernel writes them) ->
    relu(conv5x5 + bias) as the (hi,) / (hi, lo) operand tuple `out` of [N, 90, C] tensors.
    Compact queue: rows (int32 cuda [N]) / count (int32 cuda [1]) -> board i = planes[rows[i]], i < min(N, count)."""
    import torch
    require_gpu()
    code = U8 if planes.dtype == torch.uint8 else _dt_code(planes.dtype)
    if not planes.is_contiguous():
        raise NativeError("cz_input_conv: planes must be a contiguous [N, in_planes, 10, 9] tensor")
    parts = len(out)
    check(lib().cz_input_conv_q(_ptr(planes), code, planes.shape[1], _ptr(w_packed), _ptr(bias), _ptr(out[0]),
                                _ptr(out[1]) if parts == 2 else None, planes.shape[0], out[0].shape[-1],
                                _pair_code(out), parts, int(relu), _ptr(rows), _ptr(count), _stream()),
          "cz_input_conv")
    return out


def resblock(x, w1_packed, bias1, w2_packed, bias2, out=None, out_f32=None, count=None, dtype_code=None):
    """One residual block relu(conv(relu(conv(x, w1) + b1), w2) + b2 + x) in a single launch.  x and out are (hi,) or
    (hi, lo) tuples of [N, 90, C] tensors; out_f32 (split operands only) receives fp32 instead of `out`.  dtype_code: overrides
    the code derived from x (F16C86: 192 filters, a c6 block reading the input layer's c8 image)."""
    require_gpu()
    parts = len(x)
    xh = x[0]
    n, c = xh.shape[0], xh.shape[-1]
    xl = x[1] if parts == 2 else None
    yh = out[0] if out is not None else None
    yl = out[1] if out is not None and parts == 2 else None
    check(lib().cz_resblock_q(_ptr(xh), _ptr(xl), _ptr(w1_packed), _ptr(bias1), _ptr(w2_packed), _ptr(bias2),
                              _ptr(yh), _ptr(yl), _ptr(out_f32), n, c, _pair_code(x) if dtype_code is None else int(dtype_code),
                              parts, _ptr(count), _stream()), "cz_resblock")
    return out_f32 if out_f32 is not None else out


IMG_C8, IMG_C6, IMG_PAIR, EXIT_HEADS = 0, 1, 2, 3      # include/czero.h CZ_IMG_* / CZ_EXIT_HEADS


class BlockList:
    """The per-block parameter arrays a chain launch takes (HOST arrays of device pointers + the blocks' image formats), built
    once for a list of blocks [(w1_packed, bias1, w2_packed, bias2), ...] and reused every round (the default path
    runs without a HIP graph, the arrays were rebuilt per call).  Keeps the tensors alive."""

    def __init__(self, blocks, fmt_x=None, fmt_y=None):
        self.blocks = list(blocks)
        nb = self.n = len(self.blocks)
        self.arrays = tuple((C.c_void_p * nb)(*[_ptr(b[k]) for b in self.blocks]) for k in range(4))
        self.fmt_x = (C.c_int * nb)(*[int(v) for v in fmt_x]) if fmt_x is not None else None
        self.fmt_y = (C.c_int * nb)(*[int(v) for v in fmt_y]) if fmt_y is not None else None


def _block_list(blocks, fmt_x=None, fmt_y=None):
    return blocks if isinstance(blocks, BlockList) else BlockList(blocks, fmt_x, fmt_y)


def tower(x, blocks, exit_fmt, out=None, heads=None, count=None, fmt_x=None, fmt_y=None):
    """cz_tower: the consecutive residual blocks `blocks` (a BlockList, or [(w1_packed, bias1, w2_packed, bias2), ...] with fmt_x /
    fmt_y = per-block IMG_C8 / IMG_C6 lists; None = all c6) on the c8 / c6 arithmetics in ONE launch, activations in LDS between
    them.  x: the operand pair the first block reads.  exit_fmt IMG_C8 / IMG_C6: out = that operand pair; IMG_PAIR: out = (hi, lo)
    fp16 tensors [N, 90, 128] (the hand-over of a c8>N tower); EXIT_HEADS: heads = (head_w, head_b, n_policy, policy_feat,
    value_feat).  Bit-identical to len(blocks) resblock() calls."""
    require_gpu()
    L = lib()
    bl = _block_list(blocks, fmt_x, fmt_y)
    a = bl.arrays
    hw = hb = pf = vf = None
    npol = nval = 0
    if exit_fmt == EXIT_HEADS:
        hw, hb, npol, pf, vf = heads
        nval = hw.shape[0] - npol
    check(L.cz_tower(_ptr(x[0]), _ptr(x[1]), bl.n, a[0], a[1], a[2], a[3], bl.fmt_x, bl.fmt_y, int(exit_fmt),
                     _ptr(out[0]) if out is not None else None, _ptr(out[1]) if out is not None else None,
                     _ptr(hw), _ptr(hb), _ptr(pf), _ptr(vf), npol, nval, x[0].shape[0], _ptr(count), _stream()), "cz_tower")
    return out if exit_fmt != EXIT_HEADS else (pf, vf)


def resblock_chain(x, blocks, out=None, out_f32=None, count=None, dtype_code=None):
    """cz_resblock_chain: consecutive 192-filter blocks (a BlockList or [(w1_packed, bias1, w2_packed, bias2), ...], 1 .. 8) of one
    staged arithmetic in one launch; x: the c8 (uint8 image) or c6 (int8 image) operand pair; out: the same kind of pair, or
    out_f32 [N, 90, 192] for the last block.  dtype_code=F16C86: a c6 chain that starts the tower (x = the input layer's c8 image,
    block 0's first filter c8-packed).  Bit-identical to len(blocks) resblock() calls."""
    require_gpu()
    bl = _block_list(blocks)
    a = bl.arrays
    check(lib().cz_resblock_chain(_ptr(x[0]), _ptr(x[1]), bl.n, a[0], a[1], a[2], a[3],
                                  _ptr(out[0]) if out is not None else None, _ptr(out[1]) if out is not None else None,
                                  _ptr(out_f32), x[0].shape[0], x[0].shape[-1], _pair_code(x) if dtype_code is None else dtype_code,
                                  _ptr(count), _stream()),
          "cz_resblock_chain")
    return out_f32 if out_f32 is not None else out


def tower_plain(x, blocks, out, count=None):
    """cz_tower_plain: consecutive 256-filter blocks on plain fp16 / bf16 operands (a BlockList or list, 1 .. 24) in one launch;
    x / out: [N, 90, 256] tensors.  Bit-identical to len(blocks) resblock() calls with one-part operands."""
    require_gpu()
    bl = _block_list(blocks)
    a = bl.arrays
    check(lib().cz_tower_plain(_ptr(x), bl.n, a[0], a[1], a[2], a[3], _ptr(out), x.shape[0], x.shape[-1], _dt_code(x.dtype),
                               _ptr(count), _stream()), "cz_tower_plain")
    return out


def tower_pairs(x, blocks, out=None, heads=None, count=None):
    """cz_tower_pairs: consecutive (hi, lo) pair blocks (f16x3 / bf16x3) in one launch; x = (hi, lo) [N, 90, 128] fp16 / bf16.
    out = (hi, lo), or heads = (head_w, head_b, n_policy, policy_feat, value_feat) when the chain ends the tower."""
    require_gpu()
    L = lib()
    bl = _block_list(blocks)
    a = bl.arrays
    hw = hb = pf = vf = None
    npol = nval = 0
    if heads is not None:
        hw, hb, npol, pf, vf = heads
        nval = hw.shape[0] - npol
    check(L.cz_tower_pairs(_ptr(x[0]), _ptr(x[1]), bl.n, a[0], a[1], a[2], a[3],
                           _ptr(out[0]) if out is not None else None, _ptr(out[1]) if out is not None else None,
                           _ptr(hw), _ptr(hb), _ptr(pf), _ptr(vf), npol, nval, x[0].shape[0], _dt_code(x[0].dtype),
                           _ptr(count), _stream()), "cz_tower_pairs")
    return out if heads is None else (pf, vf)


def tower_c6(x, blocks, out, count=None):
    """cz_tower_c6: the consecutive c6 residual blocks `blocks` = [(w1_packed, bias1, w2_packed, bias2), ...] (2 .. 8) in ONE
    launch, activations in LDS between them; x / out: c6 operand pairs (f16 [N, 90, 128], int8 [N, 90, 256]).  Bit-identical to
    len(blocks) resblock() calls."""
    require_gpu()
    bl = _block_list(blocks)
    a = bl.arrays
    check(lib().cz_tower_c6(_ptr(x[0]), _ptr(x[1]), bl.n, a[0], a[1], a[2], a[3], _ptr(out[0]), _ptr(out[1]), x[0].shape[0],
                            _ptr(count), _stream()), "cz_tower_c6")
    return out


def tower_c6_heads(x, blocks, head_w, head_b, n_policy, policy_feat, value_feat, count=None):
    """cz_tower_c6_heads: tower_c6 ending on the tower's last block, the 1x1 head convolutions as the chain's exit."""
    require_gpu()
    bl = _block_list(blocks)
    a = bl.arrays
    check(lib().cz_tower_c6_heads(_ptr(x[0]), _ptr(x[1]), bl.n, a[0], a[1], a[2], a[3], _ptr(head_w), _ptr(head_b),
                                  _ptr(policy_feat), _ptr(value_feat), x[0].shape[0], n_policy, head_w.shape[0] - n_policy,
                                  _ptr(count), _stream()), "cz_tower_c6_heads")
    return policy_feat, value_feat


def resblock_pipelined(enable=None):
    """Which schedule the 128-filter split residual block runs on (both bit-identical): True = k_resblock_pipe (default),
    False = k_resblock.  None only queries.  Returns the previous setting."""
    L = lib()
    L.cz_resblock_pipelined.argtypes = [C.c_int]
    L.cz_resblock_pipelined.restype = C.c_int
    return L.cz_resblock_pipelined(-1 if enable is None else int(enable))


def resblock_heads(x, w1_packed, bias1, w2_packed, bias2, head_w, head_b, n_policy, policy_feat, value_feat,
                   count=None):
    """The last residual block with the 1x1 head convolutions folded in (split operands, 128 filters):
    x = (hi, lo) -> policy_feat [N, n_policy*90], value_feat [N, (6-n_policy)*90] (fp32)."""
    require_gpu()
    xh, xl = x
    check(lib().cz_resblock_heads_q(_ptr(xh), _ptr(xl), _ptr(w1_packed), _ptr(bias1), _ptr(w2_packed), _ptr(bias2),
                                    _ptr(head_w), _ptr(head_b), _ptr(policy_feat), _ptr(value_feat), xh.shape[0],
                                    xh.shape[-1], _pair_code(x), n_policy, head_w.shape[0] - n_policy,
                                    _ptr(count), _stream()), "cz_resblock_heads")
    return policy_feat, value_feat


def split_bias_act(x, bias, out, relu=True):
    """fp32 [..., C] activation -> relu(x + bias) as a (hi,) or (hi, lo) operand pair (tensors in `out`)."""
    require_gpu()
    parts = len(out)
    c = x.shape[-1] if bias is None else bias.numel()
    check(lib().cz_split_bias_act(_ptr(x), _ptr(bias), _ptr(out[0]), _ptr(out[1]) if parts == 2 else None,
                                  x.numel(), c, _dt_code(out[0].dtype), parts, int(relu), _stream()),
          "cz_split_bias_act")
    return out


def rules_fused(boards, dtype=F32, out=None):
    """move-gen + done(need_check) + planes for a batch; returns a dict of device tensors."""
    import torch
    require_gpu()
    n = boards.shape[0]
    dev = boards.device
    if out is None:
        out = dict(moves=torch.empty((n, MAXMOVES), dtype=torch.uint16, device=dev),
                   counts=torch.empty((n,), dtype=torch.uint8, device=dev),
                   over=torch.empty((n,), dtype=torch.int8, device=dev),
                   v=torch.empty((n,), dtype=torch.int8, device=dev),
                   final_move=torch.empty((n,), dtype=torch.uint16, device=dev),
                   check=torch.empty((n,), dtype=torch.uint8, device=dev),
                   planes=torch.empty((n, 14, 10, 9), dtype=torch_dtype(dtype), device=dev))
    check(lib().cz_rules_fused(_dev(boards, torch.int8), n, _dev(out["moves"], torch.uint16),
                               _dev(out["counts"], torch.uint8), _dev(out["over"], torch.int8),
                               _dev(out["v"], torch.int8), _dev(out["final_move"], torch.uint16),
                               _dev(out["check"], torch.uint8), C.c_void_p(out["planes"].data_ptr()), dtype,
                               _stream()), "cz_rules_fused")
    return out
