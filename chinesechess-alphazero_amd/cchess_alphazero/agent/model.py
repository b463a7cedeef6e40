"""Policy/value ResNet -- the reference's ``CChessModel`` (cchess_alphazero/agent/model.py:32-83)
re-expressed as a ``torch.nn.Module`` for PyTorch-ROCm.

Topology (channels-first, Keras names in brackets): Conv5x5(F) [input_conv] -> BN -> ReLU ->
N x [Conv3x3 -> BN -> ReLU -> Conv3x3 -> BN -> +skip -> ReLU] -> policy: Conv1x1(4) -> BN -> ReLU ->
Flatten(C,H,W) -> Dense(2086) softmax; value: Conv1x1(2) -> BN -> ReLU -> Flatten -> Dense(256) ReLU ->
Dense(1) tanh.  Convs have no bias; BN eps = 1e-3 (Keras default, data/model/*.json).

``InferenceNet`` is the eval-mode form used by self-play: BatchNorm folded into the convolutions and, with
trunk="mfma" (the default), every layer on hand-written HIP kernels (csrc/xq_conv.hip, xq_nn_epilogue.hip,
xq_heads.hip: input layer fused into the first residual block, one launch per block, head convolutions, dense heads
with softmax / tanh); trunk="library" keeps the MIOpen / hipBLASLt path for comparison.
"""
import hashlib
import json
import os
from logging import getLogger

import torch
import torch.nn as nn
import torch.nn.functional as F

logger = getLogger(__name__)

N_LABELS = 2086
BN_EPS = 1e-3


class ResidualBlock(nn.Module):
    def __init__(self, filters, ksize):
        super().__init__()
        self.conv1 = nn.Conv2d(filters, filters, ksize, padding=ksize // 2, bias=False)
        self.bn1 = nn.BatchNorm2d(filters, eps=BN_EPS, momentum=0.01)
        self.conv2 = nn.Conv2d(filters, filters, ksize, padding=ksize // 2, bias=False)
        self.bn2 = nn.BatchNorm2d(filters, eps=BN_EPS, momentum=0.01)

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return F.relu(x + y)


class CChessNet(nn.Module):
    """Trainable form (BatchNorm layers kept); mirrors CChessModel.build(), model.py:32-66."""

    def __init__(self, cnn_filter_num=256, cnn_first_filter_size=5, cnn_filter_size=3, res_layer_num=7,
                 value_fc_size=256, input_depth=14, policy_filters=4, value_filters=2, n_labels=N_LABELS):
        super().__init__()
        self.cfg = dict(cnn_filter_num=cnn_filter_num, cnn_first_filter_size=cnn_first_filter_size,
                        cnn_filter_size=cnn_filter_size, res_layer_num=res_layer_num,
                        value_fc_size=value_fc_size, input_depth=input_depth,
                        policy_filters=policy_filters, value_filters=value_filters, n_labels=n_labels)
        f = cnn_filter_num
        self.input_conv = nn.Conv2d(input_depth, f, cnn_first_filter_size, padding=cnn_first_filter_size // 2,
                                    bias=False)
        self.input_bn = nn.BatchNorm2d(f, eps=BN_EPS, momentum=0.01)
        self.res = nn.ModuleList([ResidualBlock(f, cnn_filter_size) for _ in range(res_layer_num)])
        self.policy_conv = nn.Conv2d(f, policy_filters, 1, bias=False)
        self.policy_bn = nn.BatchNorm2d(policy_filters, eps=BN_EPS, momentum=0.01)
        self.policy_out = nn.Linear(policy_filters * 90, n_labels)
        self.value_conv = nn.Conv2d(f, value_filters, 1, bias=False)
        self.value_bn = nn.BatchNorm2d(value_filters, eps=BN_EPS, momentum=0.01)
        self.value_dense = nn.Linear(value_filters * 90, value_fc_size)
        self.value_out = nn.Linear(value_fc_size, 1)

    def trunk(self, x):
        x = F.relu(self.input_bn(self.input_conv(x)))
        for blk in self.res:
            x = blk(x)
        return x

    def forward(self, x):
        x = self.trunk(x)
        p = F.relu(self.policy_bn(self.policy_conv(x)))
        p = self.policy_out(p.flatten(1))                      # Flatten order C,H,W (channels_first)
        v = F.relu(self.value_bn(self.value_conv(x)))
        v = F.relu(self.value_dense(v.flatten(1)))
        v = torch.tanh(self.value_out(v))
        return F.softmax(p, dim=1), v.squeeze(1)

    @classmethod
    def from_model_config(cls, mc):
        """mc: the reference's ModelConfig object (configs/*.py)."""
        return cls(cnn_filter_num=mc.cnn_filter_num, cnn_first_filter_size=mc.cnn_first_filter_size,
                   cnn_filter_size=mc.cnn_filter_size, res_layer_num=mc.res_layer_num,
                   value_fc_size=mc.value_fc_size, input_depth=getattr(mc, "input_depth", 14))


def _fold(conv, bn):
    """Fold an eval-mode BatchNorm into the preceding bias-free convolution."""
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    w = conv.weight * scale.view(-1, 1, 1, 1)
    b = bn.bias - bn.running_mean * scale
    out = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, padding=conv.padding, bias=True)
    out.weight.data.copy_(w)
    out.bias.data.copy_(b)
    return out


def tower_plan(kinds, heads_exit=True, chain_heads=True, max_chain=8):
    """The launches of a 128-filter tower behind the fused input layer (round 6: every arithmetic chains).  kinds[i] = the
    arithmetic of residual block i: "c6" / "c8" (blocks whose result is staged and converted: cz_tower) or "pair" ((hi, lo)
    operands, f16x3 / bf16x3: cz_tower_pairs), in guard_chain's order: c6 blocks, then c8 blocks, then pair blocks.  A launch
    runs ONE arithmetic; its exit hands over to the next.
    Returns a list of steps:
      ("first", 0)                       cz_input_resblock: input layer + block 0
      ("tower", [blocks], exit)          one cz_tower launch (blocks of one kind); exit = "c6" / "c8" (the image the next launch
                                         reads), "pair" (the hand-over of a c8>N tower to its pair blocks) or "heads"
      ("pairs", [blocks], heads: bool)   one cz_tower_pairs launch
      ("block", i)                       the tower's last block on its own launch (fp32 output / heads not chained)
    heads_exit: the head convolutions can be the last launch's exit (6 head filters); chain_heads=False keeps them on the last
    block's own launch (CZ_TOWER_HEADS=0; the bit-identity tests).  A launch takes at most max_chain blocks."""
    nblk = len(kinds)
    assert nblk >= 2 and all(k in ("c6", "c8", "pair") for k in kinds)
    m = next((i for i, k in enumerate(kinds) if k == "pair"), nblk)          # staged blocks [0, m), pair blocks [m, nblk)
    assert all(k == "pair" for k in kinds[m:]), kinds
    assert m != 1, "a tower whose only staged block is the first hands fp32 over after it (cz_resblock): not a fused-input tower"
    steps = [("first", 0)]
    in_chain_last = heads_exit and chain_heads           # the tower's last block inside a chain (heads as the exit)
    end = nblk if in_chain_last else nblk - 1            # blocks [1, end) are chained, block nblk - 1 maybe on its own

    def chunks(lo, hi):
        out = []
        while lo < hi:
            out.append(list(range(lo, min(hi, lo + max_chain))))
            lo += max_chain
        return out
    # staged chains: one per arithmetic (the last c6 block writes the c8 image the c8 chain reads), at most max_chain blocks each
    m6 = next((i for i, k in enumerate(kinds) if k != "c6"), nblk)
    staged = chunks(1, min(m6, m, end)) + chunks(max(m6, 1), min(m, end))
    for blk in staged:
        nxt = blk[-1] + 1
        steps.append(("tower", blk, "heads" if nxt == nblk else kinds[nxt]))
    if m < nblk:
        pr = chunks(max(m, 1), end)
        for j, blk in enumerate(pr):
            steps.append(("pairs", blk, blk[-1] + 1 == nblk))
    if not in_chain_last:
        steps.append(("block", nblk - 1))
    return steps


def ip_segments(kinds, max_chain=12, first_alone=False):
    """The launches of a 192-filter tower's blocks (round 6): ("chain", [blocks]) = one cz_resblock_chain launch of consecutive
    blocks of one arithmetic (c8 / c6 images: k_resblock_ip4_c8; (hi, lo) pairs: k_tower_pairs4), ("block", [i]) = a block on its
    own launch -- with first_alone (the six-wave kernels, CZ_IP_PAIR=0) a c6 tower's block 0, which reads the input layer's c8
    image (the four-wave kernel takes it as the first block of its chain: dtype CZ_F16C86).  A chain that ends the tower writes
    fp32 (the head convolutions' input)."""
    segs, i, n = [], 0, len(kinds)
    while i < n:
        k = kinds[i]
        if k == "c6" and i == 0 and first_alone:
            segs.append(("block", [i]))
            i += 1
            continue
        j = i
        while j < n and kinds[j] == k and j - i < max_chain:
            j += 1
        segs.append(("chain", list(range(i, j))))
        i = j
    return segs


def events_ms(events):
    """Per-BLOCK times (ms) of the tower launches recorded in InferenceNet.block_events: a (start, end) pair is one residual
    block; (start, end, m) is a launch of m chained blocks (cz_tower / cz_tower_pairs), counted as m blocks of elapsed / m each, so that the
    list keeps one entry per block of the tower in tower order."""
    out = []
    for e in events:
        ms = e[0].elapsed_time(e[1])
        m = e[2] if len(e) > 2 else 1
        out += [ms / m] * m
    return out


class InferenceNet(nn.Module):
    """Eval-mode network for self-play: BN folded, channels_last, optional reduced precision.
    Outputs are always fp32: softmax policy [B, 2086] and value [B].

    trunk="library": the residual tower runs in MIOpen (+ the hand-written bias/skip/ReLU pass).
    trunk="mfma":    the tower's 3x3 convolutions run on the hand-written MFMA kernel (csrc/xq_conv.hip) with the
                     epilogue fused.  With dtype=float32 the operands are (hi, lo) bf16 pairs -- three bf16 MFMAs per
                     product, fp32 accumulate, fp32-class results (policy / value within 1e-4 of the fp32 network) --
                     with bf16 / fp16 they are plain 2-byte operands.
    arith (trunk="mfma", dtype float32; CZ_TOWER_ARITH overrides the default) -- how an fp32 product is formed:
      "bf16x3"  (hi, lo) bf16 pairs, w_hi x_hi + w_lo x_hi + w_hi x_lo: 2^-17 per product whatever the operands' range;
      "f16x3"   the same three MFMAs on (hi, lo) fp16 pairs: 22 bits per operand, ~8x more accurate on networks whose
                activations and folded filters sit in fp16's range (round 4; tools/f16x3_probe.py: the matrix unit honours
                fp16 subnormals, which the filters' lo parts are), less accurate than bf16x3 when they are tiny;
      "c8"      (128 / 192 filters) an fp16 main term plus two block-scaled fp8 correction terms (csrc/xq_conv.hip,
                k_resblock_c8 / k_resblock_ip_c8): one fp16 and two fp8 matrix instructions per 64 input channels, 2^-16
                per product;
      "c8>N"    the first N residual blocks on c8, the rest on f16x3 (error ~ sqrt(N): the guard's middle ground);
      "c6"      (128 / 192 filters, >= 2 blocks, fused kernels) c8 with bf6 (e3m2) correction operands: half the matrix time of the
                e4m3 ones, 2^-15 per product; needs act_exps (the images' exponents, from the calibration);
      "c6>N"    the first N residual blocks on c6, the rest on c8 (round 5: the step down from c6 is not all-or-nothing; block
                N - 1 writes a c8 image, cz_conv3x3_c6_pack_weights y_exp = 127).
    None of the reduced forms is trusted blindly: guarded_inference_net() below measures the candidate against float64 on
    calibration positions when weights are loaded and falls back along c6 -> c6>N -> c8 -> c8>N -> f16x3 -> bf16x3."""

    def __init__(self, net: CChessNet, dtype=torch.float32, trunk="library", arith=None, act_shift=None, act_exps=None):
        """act_shift = (s_x, [s_mid per block]): power-of-two activation scales of the residual stream and of every block's
        intermediate tensor, applied as an EXACT reparametrisation of the folded network (ReLU commutes with a positive
        scale): the input layer is multiplied by 2^s_x, block i's first convolution by 2^(s_mid_i - s_x) (bias 2^s_mid_i), its
        second by 2^(s_x - s_mid_i) (bias 2^s_x), the head convolutions by 2^-s_x.  Outputs are unchanged; the tower's tensors
        move into the range the reduced operand formats resolve (e4m3 saturates at 448, fp16 at 65504).  Chosen by
        guarded_inference_net from the measured activation ranges; None = no scaling.
        act_exps (arith "c6" only) = ([k_mid per block], [k_out per block]): the exponents of the bf6 activation images,
        2^k * 28 >= the tensor's largest value (c6_exponents of the calibration's activation maxima)."""
        super().__init__()
        net = net.eval()
        assert trunk in ("library", "mfma")
        self.dtype = dtype
        self.trunk = trunk
        from_env = arith is None and bool(os.environ.get("CZ_TOWER_ARITH"))    # (an explicit argument is never "from the environment")
        arith = arith or os.environ.get("CZ_TOWER_ARITH") or "bf16x3"
        nblk = net.cfg["res_layer_num"]
        c8_blocks = 0
        self.c6 = False
        self.c6_blocks = 0
        if arith == "c6" or arith.startswith("c6>"):
            # c8 with bf6 correction operands (half the matrix time of the e4m3 ones): 128 filters, >= 2 blocks; "c6>N": the
            # first N blocks only, the rest c8
            n6 = int(arith[3:]) if arith.startswith("c6>") else nblk
            assert 1 <= n6 <= nblk, arith
            self.c6 = (trunk == "mfma" and dtype == torch.float32 and net.cfg["cnn_filter_num"] in (128, 192) and nblk >= 2)
            if self.c6 and act_exps is None:
                if not from_env:
                    raise ValueError("arith='c6' needs the activation images' exponents: build the network through "
                                     "guarded_inference_net (it measures them on the calibration positions)")
                # (CZ_TOWER_ARITH=c6 reaching a direct construction -- tests, tools: c8 is the nearest arithmetic that needs
                #  no calibration)
                logger.warning("CZ_TOWER_ARITH=%s without calibrated exponents: using c8 (guarded_inference_net measures them)", arith)
                self.c6 = False
            self.c6_blocks = n6 if self.c6 else 0
            arith = "c8"
        self.act_exps = ([int(v) for v in act_exps[0]], [int(v) for v in act_exps[1]]) if self.c6 else None
        if arith.startswith("c8"):
            c8_blocks = int(arith[3:]) if arith.startswith("c8>") else nblk
            assert 0 <= c8_blocks <= nblk, arith
            if not (trunk == "mfma" and dtype == torch.float32 and net.cfg["cnn_filter_num"] in (128, 192)):
                c8_blocks = 0               # the c8 arithmetic exists for the 128- and 192-filter split towers
            arith = "c8" if c8_blocks else "f16x3"
        assert arith in ("bf16x3", "f16x3", "c8"), arith
        if arith == "f16x3" and not (trunk == "mfma" and dtype == torch.float32):
            arith = "bf16x3"
        self.arith = arith                  # the family; arith_name says how many blocks run c8
        self.c8_blocks = c8_blocks
        self.fused_epilogue = True          # on the GPU: hand-written bias + skip + ReLU pass after each conv
        self.fused_blocks = True            # trunk="mfma", fp32, 128 filters: one launch per residual block
        self.fused_heads = True             # ... and the 1x1 head convolutions folded into the last block's store pass
        # ... and the input layer computed by the first block's copy waves (uint8 planes; CZ_FUSED_INPUT=0: cz_input_conv)
        self.fused_input = os.environ.get("CZ_FUSED_INPUT", "1") != "0"
        # dense layers + softmax / tanh on the hand-written kernels (csrc/xq_heads.hip); CZ_FUSED_TAIL=0: the hipBLASLt /
        # PyTorch tail they replace (A/B runs)
        self.fused_tail = os.environ.get("CZ_FUSED_TAIL", "1") != "0"
        self.block_events = None            # bench.py: list collecting (start, end[, blocks]) HIP events around tower launches
        self.last_plan = None               # tower_plan's steps of the last chained forward
        # consecutive blocks as one launch (cz_tower / cz_tower_pairs, tower_plan; CZ_TOWER_CHAIN=0: one launch per block)
        self.chain_blocks = os.environ.get("CZ_TOWER_CHAIN", "1") != "0"
        self.chain_heads = os.environ.get("CZ_TOWER_HEADS", "1") != "0"
        self.input_depth = net.cfg["input_depth"]
        self.filters = net.cfg["cnn_filter_num"]
        with torch.no_grad():
            self.input_conv = _fold(net.input_conv, net.input_bn)
            self.res = nn.ModuleList()
            for blk in net.res:
                self.res.append(nn.ModuleList([_fold(blk.conv1, blk.bn1), _fold(blk.conv2, blk.bn2)]))
            self.policy_conv = _fold(net.policy_conv, net.policy_bn)
            self.value_conv = _fold(net.value_conv, net.value_bn)
            self.policy_out = nn.Linear(net.policy_out.in_features, net.policy_out.out_features)
            self.value_dense = nn.Linear(net.value_dense.in_features, net.value_dense.out_features)
            self.value_out = nn.Linear(net.value_out.in_features, 1)
            self.policy_out.load_state_dict(net.policy_out.state_dict())
            self.value_dense.load_state_dict(net.value_dense.state_dict())
            self.value_out.load_state_dict(net.value_out.state_dict())
            self.act_shift = None
            if act_shift is not None and (act_shift[0] != 0 or any(act_shift[1])):
                sx, smid = int(act_shift[0]), [int(v) for v in act_shift[1]]
                assert len(smid) == len(self.res)
                self.act_shift = (sx, smid)
                self.input_conv.weight.mul_(2.0 ** sx)
                self.input_conv.bias.mul_(2.0 ** sx)
                for (c1, c2), sm in zip(self.res, smid):
                    c1.weight.mul_(2.0 ** (sm - sx))
                    c1.bias.mul_(2.0 ** sm)
                    c2.weight.mul_(2.0 ** (sx - sm))
                    c2.bias.mul_(2.0 ** sx)
                self.policy_conv.weight.mul_(2.0 ** -sx)      # (their biases act on the unscaled head features)
                self.value_conv.weight.mul_(2.0 ** -sx)
            packed = self._pack_trunk() if trunk == "mfma" else None
        self.to(dtype)
        self.to(memory_format=torch.channels_last)
        if packed is not None:
            # registered after the dtype conversion: packed filters are raw 2-byte MFMA operands (kept as int16 so
            # that no later .to(dtype) can reinterpret them), biases stay fp32
            for i, (w1, b1, w2, b2) in enumerate(packed):
                self.register_buffer(f"tw{i}a", w1.view(torch.int16))
                self.register_buffer(f"tb{i}a", b1)
                self.register_buffer(f"tw{i}b", w2.view(torch.int16))
                self.register_buffer(f"tb{i}b", b2)
            self.register_buffer("in_bias32", self._in_bias32)
            self.register_buffer("in_w", self._packed_in.view(torch.int16))
            self.register_buffer("in_table32", self._in_table)
            self.register_buffer("head_w32", self._head_w)
            self.register_buffer("head_b32", self._head_b)
            self._tail_dtype = self._tail_pack[0].dtype
            for name, t in zip(("tail_wp", "tail_bp", "tail_w1", "tail_b1", "tail_w2"), self._tail_pack):
                self.register_buffer(name, t.view(torch.int16) if t.dtype in (torch.bfloat16, torch.float16) else t)
            del self._packed_in, self._in_table, self._in_bias32, self._head_w, self._head_b, self._tail_pack
        self._bufs = {}
        self.eval()
        for p in self.parameters():
            p.requires_grad_(False)

    # ---- hand-written trunk ----
    @property
    def parts(self):
        return 2 if self.dtype == torch.float32 else 1

    @property
    def operand_dtype(self):
        if self.arith in ("c8", "f16x3"):
            return torch.float16
        return torch.bfloat16 if self.dtype == torch.float32 else self.dtype

    @property
    def arith_name(self):
        """"bf16x3" / "f16x3" / "c8" / "c8>N" (the first N residual blocks on c8, the rest on f16x3) / "c6" / "c6>N" (the
        first N on c6, the rest on c8)."""
        if self.c6:
            return "c6" if self.c6_blocks == len(self.res) else f"c6>{self.c6_blocks}"
        if self.arith == "c8" and self.c8_blocks < len(self.res):
            return f"c8>{self.c8_blocks}"
        return self.arith

    def _pack_trunk(self):
        """fp32 folded filters -> MFMA fragment order (cz_conv3x3_pack_weights); called before the dtype conversion."""
        from cchess_alphazero import _native
        self._in_bias32 = self.input_conv.bias.detach().float().clone()
        # the dense tail (cz_heads_tail): both matrices as (hi, lo) bf16 pairs in fragment order, the rest fp32
        tail_dt = torch.bfloat16 if self.operand_dtype == torch.bfloat16 else torch.float16
        self._tail_pack = (_native.pack_fc_weights(self.policy_out.weight, tail_dt),
                           self.policy_out.bias.detach().float().clone(),
                           _native.pack_fc_weights(self.value_dense.weight, tail_dt),
                           self.value_dense.bias.detach().float().clone(),
                           self.value_out.weight.detach().float().reshape(-1).clone())
        self._tail_b2 = float(self.value_out.bias.detach().float().item())
        self._head_w = torch.cat([self.policy_conv.weight.detach().float().flatten(1),
                                  self.value_conv.weight.detach().float().flatten(1)]).contiguous()
        self._head_b = torch.cat([self.policy_conv.bias.detach().float(), self.value_conv.bias.detach().float()])
        self._packed_in = _native.pack_input_conv_weights(self.input_conv.weight, self.operand_dtype, self.parts)
        self._in_table = _native.input_table(self.input_conv.weight)      # the gather form of the input layer
        out = []
        pack_c8 = lambda w: _native.pack_conv3x3_c8_weights(w).view(torch.float16)    # raw bytes, like the others
        pack = lambda w: _native.pack_conv3x3_weights(w, self.operand_dtype, self.parts)
        for i, (c1, c2) in enumerate(self.res):
            if self.c6 and i < self.c6_blocks:
                # block i reads the stream image of exponent k_out[i - 1] (block 0: the fused input layer's c8 image, e4m3
                # filters), writes its intermediate image with k_mid[i] and the stream image with k_out[i]; the last c6 block of
                # a hybrid tower writes a c8 image instead (y_exp = 127) for the c8 blocks behind it
                kmid, kout = self.act_exps
                assert len(kmid) == len(kout) == len(self.res)
                pk6 = lambda w, kx, ky: _native.pack_conv3x3_c6_weights(w, kx, ky).view(torch.float16)
                hand_over = i + 1 == self.c6_blocks and self.c6_blocks < len(self.res)
                out.append((pack_c8(c1.weight) if i == 0 else pk6(c1.weight, kout[i - 1], kmid[i]),
                            c1.bias.detach().float().clone(),
                            pk6(c2.weight, kmid[i], 127 if hand_over else kout[i]), c2.bias.detach().float().clone()))
                continue
            pk = pack_c8 if i < self.c8_blocks else pack
            out.append((pk(c1.weight), c1.bias.detach().float().clone(),
                        pk(c2.weight), c2.bias.detach().float().clone()))
        return out

    def _operands(self, n, device):
        """Three rotating activation buffers ([n, 90, C] per part) + the fp32 output of the last layer.  Allocated once
        for the largest batch seen (the evaluation queue) and sliced for smaller ones (the arena / UCI evaluate only
        the rows that carry a leaf, a different count every round)."""
        key = str(device)
        cap = self._bufs.get(key, (0,))[0]
        if n > cap:
            od, c = self.operand_dtype, self.filters
            if self.arith == "c8":          # (f16 operand, c8 correction image: e4m3 lo, e4m3 value)
                # (c6: the same bytes hold bf6 pieces; int8 tags the pair for the entry points)
                bufs = [(torch.empty((n, 90, c), dtype=od, device=device),
                         torch.empty((n, 90, 2 * c), dtype=torch.int8 if self.c6 else torch.uint8, device=device))
                        for _ in range(3)]
            else:
                bufs = [tuple(torch.empty((n, 90, c), dtype=od, device=device) for _ in range(self.parts))
                        for _ in range(3)]
            last = torch.empty((n, 90, c), dtype=torch.float32 if self.parts == 2 else od, device=device)
            self._bufs[key] = (n, bufs, last)
        cap, bufs, last = self._bufs[key]
        if n == cap:
            return bufs, last
        return [tuple(t[:n] for t in b) for b in bufs], last[:n]

    @staticmethod
    def _as_f16_pair(pair):
        """A c8 operand pair's storage seen as an (hi, lo) fp16 pair (same bytes: [n, 90, 2C] u8 = [n, 90, C] f16); an (hi, lo)
        pair of the operand dtype (f16x3 / bf16x3 towers) as it is."""
        return (pair[0], pair[1].view(torch.float16)) if pair[1].dtype in (torch.uint8, torch.int8) else pair

    def _trunk_mfma(self, planes, heads=None, rows=None, count=None, masks=None):
        """planes: the evaluation queue as the search kernel wrote it ([n, in_planes, 10, 9], any supported dtype).
        heads = (n_policy, policy_feat, value_feat): fold the 1x1 head convolutions into the last block where the
        kernel exists for the shape (returns None then), else returns the [n, 90, c] trunk output."""
        from cchess_alphazero import _native
        n, c = planes.shape[0], self.filters
        (cur, tmp, nxt), last = self._operands(n, planes.device)
        # compact queue (rows / count on the device): board i = planes[rows[i]] for i < count; the launch shapes stay
        # those of the whole queue, the kernels read the count themselves
        nblk = len(self.res)
        n8 = self.c8_blocks if self.arith == "c8" else 0        # blocks [0, n8) on the c8 arithmetic
        # whole residual block in one launch where k_resblock exists for the shape
        fused = self.fused_blocks and ((c in (128, 192)) or (c == 256 and self.parts == 1))
        # input layer + first block in one launch: 128 filters, split operands, byte planes, a tower of >= 2 blocks
        # (a hybrid tower whose only c8 block is the first hands fp32 over after it: that block stays on cz_resblock)
        if self.c6 and c == 128 and planes.dtype != torch.uint8:
            # c6 exists on the fused kernels only, whose input layer is a gather over the OCCUPIED squares of byte planes:
            # the feature planes are 0 / 1 by construction (environment/static_env.py state_to_planes), in any dtype
            planes = (planes != 0).to(torch.uint8)
        first_fused = (fused and self.fused_input and c == 128 and self.parts == 2 and nblk >= 2 and
                       planes.dtype == torch.uint8 and n8 != 1)
        if self.c6 and not (fused and (first_fused or c == 192)):
            raise RuntimeError("arith='c6' runs on the fused kernels only (128 filters: uint8 planes, fused input layer and blocks)")
        if not first_fused:
            if self.arith == "c8" and n8 == 0:                  # (cannot happen through the constructor; kept total)
                cur, tmp, nxt = (self._as_f16_pair(t) for t in (cur, tmp, nxt))
            # (a c6 tower's input layer writes a c8 image: its first block's first filter is a c8 filter)
            cur_in = (cur[0], cur[1].view(torch.uint8)) if self.c6 else cur
            _native.input_conv(planes.contiguous(), self.in_w.view(self.operand_dtype), self.in_bias32, cur_in,
                               rows=rows, count=count)
        # (round 5 / 6) the blocks behind the fused input layer run as CHAINS -- one launch for consecutive blocks with the
        # activations staying in LDS (cz_tower for c6 / c8 blocks, cz_tower_pairs for f16x3 / bf16x3 ones; tower_plan): the 7 x 128
        # benchmark tower is FIRST | blocks 1 .. 6 with the head convolutions as the chain's exit, a c8>3 tower FIRST | 1 .. 2 (exit:
        # fp16 pairs) | 3 .. 6 (heads)
        if fused and first_fused and self.chain_blocks:
            return self._tower_chained(planes, cur, nxt, last, heads, rows, count, masks)
        if fused and c == 192 and self.parts == 2 and self.chain_blocks:
            return self._tower_192(cur, nxt, last, count)
        if fused and c == 256 and self.parts == 1 and self.chain_blocks:
            # the deep tower on plain operands: all blocks in one cz_tower_plain launch (24 at most per launch)
            key = ("plan256", self.tb0a.data_ptr())
            if key not in self._bufs:
                self._bufs[key] = [_native.BlockList([self._block_params(i) for i in range(lo, min(nblk, lo + 24))])
                                   for lo in range(0, nblk, 24)]
            self.last_plan = [("chain256", list(range(24 * j, 24 * j + bl.n)), "plain") for j, bl in enumerate(self._bufs[key])]
            x = cur[0]
            for j, bl in enumerate(self._bufs[key]):
                ev = None
                if self.block_events is not None:
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record()
                y = last if j + 1 == len(self._bufs[key]) else nxt[0]
                _native.tower_plain(x, bl, y, count=count)
                x = y
                if ev is not None:
                    ev[1].record()
                    self.block_events.append(ev + (bl.n,) if bl.n > 1 else ev)
            return last
        for i in range(nblk):
            w1 = getattr(self, f"tw{i}a").view(self.operand_dtype)
            w2 = getattr(self, f"tw{i}b").view(self.operand_dtype)
            if self.c6:
                # the image tensors' element type is the tag that selects the kernel: int8 = a c6 block (its input AND output
                # pair, also where the last c6 block of a hybrid tower writes a c8 image), uint8 = a c8 block
                tag = torch.int8 if i < self.c6_blocks else torch.uint8
                cur, tmp, nxt = ((t[0], t[1].view(tag)) for t in (cur, tmp, nxt))
            if fused:
                b1, b2 = getattr(self, f"tb{i}a"), getattr(self, f"tb{i}b")
                ev = None
                if self.block_events is not None:
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record()
                if i == 0 and first_fused:
                    _native.input_resblock(planes.contiguous(), self.in_table32, self.in_bias32, w1, b1, w2, b2, out=nxt,
                                           rows=rows, count=count, masks=masks)
                    cur, nxt = nxt, cur
                elif i + 1 == n8 and n8 < nblk:
                    # the last c8 block of a hybrid tower: fp32 out, re-split into (hi, lo) fp16 pairs for the f16x3 blocks
                    _native.resblock(cur, w1, b1, w2, b2, out_f32=last, count=count)
                    cur, tmp, nxt = (self._as_f16_pair(t) for t in (cur, tmp, nxt))
                    _native.split_bias_act(last, None, cur, relu=False)
                elif i + 1 < nblk:
                    # (192 filters, c6: block 0 reads the input layer's c8 image -- its own dtype code)
                    code = _native.F16C86 if (self.c6 and i == 0 and not first_fused) else None
                    _native.resblock(cur, w1, b1, w2, b2, out=nxt, count=count, dtype_code=code)
                    cur, nxt = nxt, cur
                elif heads is not None and self.parts == 2 and c == 128:
                    _native.resblock_heads(cur, w1, b1, w2, b2, self.head_w32, self.head_b32, heads[0], heads[1],
                                           heads[2], count=count)
                elif self.parts == 2:
                    _native.resblock(cur, w1, b1, w2, b2, out_f32=last, count=count)
                else:
                    _native.resblock(cur, w1, b1, w2, b2, out=(last,), count=count)
                if ev is not None:
                    ev[1].record()
                    self.block_events.append(ev)
                continue
            if i == n8 and 0 < n8:          # (per-convolution launches, hybrid tower: the same re-split)
                cur, tmp, nxt = (self._as_f16_pair(t) for t in (cur, tmp, nxt))
                _native.split_bias_act(last, None, cur, relu=False)
            conv = _native.conv3x3_c8 if i < n8 else _native.conv3x3
            conv(cur, w1, getattr(self, f"tb{i}a"), out=tmp)
            if i + 1 < nblk and i + 1 != n8:
                conv(tmp, w2, getattr(self, f"tb{i}b"), skip=cur, out=nxt)
                cur, nxt = nxt, cur
            elif self.parts == 2:
                conv(tmp, w2, getattr(self, f"tb{i}b"), skip=cur, out_f32=last)
            else:
                conv(tmp, w2, getattr(self, f"tb{i}b"), skip=cur, out=(last,))
        if heads is not None and fused and self.parts == 2 and c == 128:
            return None                                                  # the head features are already written
        return last                                                      # [n, 90, c] channels-last trunk output

    def block_kinds(self):
        """Per residual block, the arithmetic its launch runs (tower_plan's kinds): "c6" / "c8" / "pair"."""
        nblk = len(self.res)
        if self.c6:
            return ["c6"] * self.c6_blocks + ["c8"] * (nblk - self.c6_blocks)
        if self.arith == "c8":
            return ["c8"] * self.c8_blocks + ["pair"] * (nblk - self.c8_blocks)
        return ["pair"] * nblk

    def _block_params(self, i):
        od = self.operand_dtype
        return (getattr(self, f"tw{i}a").view(od), getattr(self, f"tb{i}a"), getattr(self, f"tw{i}b").view(od), getattr(self, f"tb{i}b"))

    def _tower_chained(self, planes, cur, nxt, last, heads, rows, count, masks):
        """The fused 128-filter tower as tower_plan's launches.  cur / nxt: two operand buffers of _operands (storage: (f16, image
        bytes) for the c8 / c6 family, (hi, lo) otherwise); returns like _trunk_mfma."""
        from cchess_alphazero import _native
        kinds = self.block_kinds()
        nblk = len(kinds)
        heads_ok = heads is not None and self.parts == 2 and self.filters == 128
        # the head convolutions as a pair chain's exit read the block's value as hi + lo: fp16 pairs stand for it to 2^-22, bf16
        # pairs only to 2^-17 -- bf16x3 (the guard's last resort before the fp32 library trunk) keeps its HEADS launch, which
        # works on the fp32 value
        chain_heads = self.chain_heads and not (kinds[-1] == "pair" and self.operand_dtype == torch.bfloat16)
        key = ("plan", heads_ok, chain_heads, self.tb0a.data_ptr())     # (device pointers inside: rebuilt if the module moved)
        if key not in self._bufs:
            # launch steps + the per-launch pointer arrays, built once (invalidated with the buffers on a weight repack)
            plan = []
            for step in tower_plan(kinds, heads_exit=heads_ok, chain_heads=chain_heads):
                if step[0] in ("tower", "pairs"):
                    blk = step[1]
                    fmt = [_native.IMG_C6 if kinds[blk[0]] == "c6" else _native.IMG_C8] * len(blk)     # (one arithmetic per launch)
                    bl = _native.BlockList([self._block_params(i) for i in blk], *((fmt, fmt) if step[0] == "tower" else ()))
                    plan.append(step + (bl,))
                else:
                    plan.append(step)
            self._bufs[key] = plan
        self.last_plan = [st[:3] if st[0] != "first" else st for st in self._bufs[key]]      # (bench.py: what a forward launches)
        img_tag = {"c6": torch.int8, "c8": torch.uint8}

        def view(t, kind):                      # an operand buffer seen as the pair of `kind`
            if kind == "pair":
                return (t[0], t[1].view(t[0].dtype)) if t[1].dtype in (torch.uint8, torch.int8) else t
            return (t[0], t[1].view(img_tag[kind]))
        fmt_code = {"c6": _native.IMG_C6, "c8": _native.IMG_C8, "pair": _native.IMG_PAIR}
        hd = (self.head_w32, self.head_b32, heads[0], heads[1], heads[2]) if heads_ok else None
        done_heads = False
        for step in self._bufs[key]:
            ev = None
            if self.block_events is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            if step[0] == "first":
                w1, b1, w2, b2 = self._block_params(0)
                _native.input_resblock(planes.contiguous(), self.in_table32, self.in_bias32, w1, b1, w2, b2,
                                       out=view(nxt, kinds[0]), rows=rows, count=count, masks=masks)
                cur, nxt = nxt, cur
                nb = 1
            elif step[0] == "tower":
                _, blk, ex, bl = step
                nb = len(blk)
                if ex == "heads":
                    _native.tower(view(cur, kinds[blk[0]]), bl, _native.EXIT_HEADS, heads=hd, count=count)
                    done_heads = True
                else:
                    _native.tower(view(cur, kinds[blk[0]]), bl, fmt_code[ex], out=view(nxt, ex), count=count)
                    cur, nxt = nxt, cur
            elif step[0] == "pairs":
                _, blk, with_heads, bl = step
                nb = len(blk)
                if with_heads:
                    _native.tower_pairs(view(cur, "pair"), bl, heads=hd, count=count)
                    done_heads = True
                else:
                    _native.tower_pairs(view(cur, "pair"), bl, out=view(nxt, "pair"), count=count)
                    cur, nxt = nxt, cur
            else:                               # the last block on its own launch
                i = step[1]
                nb = 1
                w1, b1, w2, b2 = self._block_params(i)
                x = view(cur, kinds[i])
                if heads_ok:
                    _native.resblock_heads(x, w1, b1, w2, b2, self.head_w32, self.head_b32, heads[0], heads[1], heads[2],
                                           count=count)
                    done_heads = True
                else:
                    _native.resblock(x, w1, b1, w2, b2, out_f32=last, count=count)
            if ev is not None:
                ev[1].record()
                self.block_events.append(ev + (nb,) if nb > 1 else ev)
        return None if done_heads else last

    def _tower_192(self, cur, nxt, last, count):
        """The 192-filter tower (every split arithmetic) behind cz_input_conv as ip_segments' launches; returns the fp32 trunk output."""
        from cchess_alphazero import _native
        kinds = self.block_kinds()
        nblk = len(kinds)
        legacy = os.environ.get("CZ_IP_PAIR", "1")[:1] == "0"          # (the six-wave kernels: block 0 of a c6 tower on its own launch)
        key = ("plan192", legacy, self.tb0a.data_ptr())
        if key not in self._bufs:
            self._bufs[key] = [(kind, blk, _native.BlockList([self._block_params(i) for i in blk]))
                               for kind, blk in ip_segments(kinds, first_alone=legacy)]
        self.last_plan = [("chain192" if kind == "chain" else "block192", blk, kinds[blk[0]]) for kind, blk, _ in self._bufs[key]]
        tag = {"c6": torch.int8, "c8": torch.uint8}
        for kind, blk, bl in self._bufs[key]:
            ev = None
            if self.block_events is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            k, i1 = kinds[blk[0]], blk[-1]
            tower_end = i1 + 1 == nblk
            to_pairs = k == "c8" and not tower_end and kinds[i1 + 1] == "pair"
            if k == "pair":
                x = self._as_f16_pair(cur)
                w1, b1, w2, b2 = bl.blocks[0]
                if kind == "chain" and tower_end:   # consecutive pair blocks in one launch (k_tower_pairs4<E, 192>)
                    _native.resblock_chain(x, bl, out_f32=last, count=count)
                elif kind == "chain":
                    _native.resblock_chain(x, bl, out=self._as_f16_pair(nxt), count=count)
                    cur, nxt = nxt, cur
                elif tower_end:
                    _native.resblock(x, w1, b1, w2, b2, out_f32=last, count=count)
                else:
                    _native.resblock(x, w1, b1, w2, b2, out=self._as_f16_pair(nxt), count=count)
                    cur, nxt = nxt, cur
            elif kind == "block":                   # a c6 tower's block 0: c8 image in, c6 (or, c6>1, c8) image out
                w1, b1, w2, b2 = bl.blocks[0]
                _native.resblock((cur[0], cur[1].view(torch.uint8)), w1, b1, w2, b2, out=(nxt[0], nxt[1].view(torch.int8)),
                                 count=count, dtype_code=_native.F16C86)
                cur, nxt = nxt, cur
            else:
                # (a c6 chain that starts the tower reads the input layer's c8 image: CZ_F16C86)
                first6 = k == "c6" and blk[0] == 0
                x = (cur[0], cur[1].view(torch.uint8 if first6 else tag[k]))
                code = _native.F16C86 if first6 else None
                # (a c6 chain whose last block hands over to c8 blocks writes a c8 image)
                out_tag = tag[kinds[i1 + 1]] if (k == "c6" and not tower_end and kinds[i1 + 1] == "c8") else tag[k]
                if tower_end or to_pairs:
                    _native.resblock_chain(x, bl, out_f32=last, count=count, dtype_code=code)
                    if to_pairs:                    # the hand-over of a c8>N tower: re-split into (hi, lo) fp16 pairs
                        _native.split_bias_act(last, None, self._as_f16_pair(cur), relu=False)
                else:
                    _native.resblock_chain(x, bl, out=(nxt[0], nxt[1].view(out_tag)), count=count, dtype_code=code)
                    cur, nxt = nxt, cur
            if ev is not None:
                ev[1].record()
                self.block_events.append(ev + (len(blk),) if len(blk) > 1 else ev)
        return last

    def _head_feats(self, n, npol, device):
        key = ("hf", n, str(device))
        if key not in self._bufs:
            self._bufs[key] = (torch.zeros((n, npol * 90), dtype=torch.float32, device=device),
                               torch.zeros((n, (6 - npol) * 90), dtype=torch.float32, device=device))
        return self._bufs[key]

    def _trunk_fused(self, x):
        """Trunk with the hand-written epilogue (csrc/xq_nn_epilogue.hip): every convolution is followed by ONE
        in-place pass  y = relu(y + bias (+ skip))  instead of PyTorch's separate bias / add / ReLU passes."""
        from cchess_alphazero import _native

        def conv(m, t):
            return F.conv2d(t, m.weight, None, m.stride, m.padding)
        cl = torch.channels_last
        x = _native.bias_act_(conv(self.input_conv, x).contiguous(memory_format=cl), self.input_conv.bias)
        for c1, c2 in self.res:
            y = _native.bias_act_(conv(c1, x).contiguous(memory_format=cl), c1.bias)
            x = _native.bias_act_(conv(c2, y).contiguous(memory_format=cl), c2.bias, residual=x)
        return x

    def supports_compact_queue(self):
        """True when the whole convolutional part runs on the hand-written kernels that take the board count from the
        device (cz_*_q): 128- or 192-filter tower on the fused residual-block kernels, 6 head filters."""
        return (self.trunk == "mfma" and self.fused_blocks and self.filters in (128, 192) and
                getattr(self, "head_w32", torch.empty(0)).shape[0] == 6)

    def takes_masks(self, planes_dtype=torch.uint8):
        """True when forward(masks=...) reads the occupancy boards INSTEAD of the planes: the input layer fused into the
        first residual block (_trunk_mfma's `first_fused`) -- the engine then lets the search kernel skip the planes
        (Search.leaf_planes(False))."""
        if self.trunk != "mfma" or planes_dtype != torch.uint8:
            return False
        c, nblk = self.filters, len(self.res)
        n8 = self.c8_blocks if self.arith == "c8" else 0
        fused = self.fused_blocks and ((c in (128, 192)) or (c == 256 and self.parts == 1))
        return bool(fused and self.fused_input and c == 128 and self.parts == 2 and nblk >= 2 and n8 != 1)

    def supports_logits(self):
        """True when forward(logits=True) is available: the hand-written dense tail on 6 head filters."""
        return (self.trunk == "mfma" and self.fused_tail and getattr(self, "head_w32", torch.empty(0)).shape[0] == 6 and
                self.policy_out.in_features in (180, 360) and self.value_dense.in_features in (180, 360))

    @torch.no_grad()
    def forward(self, planes, rows=None, count=None, out=None, logits=False, masks=None):
        """planes: the evaluation queue.  rows / count (int32 cuda tensors, cz_search_round_q): evaluate only the
        boards planes[rows[i]], i < count -- the result rows are indexed by i; rows beyond count are undefined.
        out = (policy [n, 2086] fp32, value [n] fp32): write the results there (the engine's queue tensors) instead of
        into fresh tensors.  logits=True (hand-written tail only; see supports_logits): the policy rows are left as raw
        logits -- for a search object in policy_logits mode, which needs the legal moves' entries only.
        masks (int32 [n, 96], Search.leaf_masks): the same positions as occupancy boards; the fused input layer of the first
        residual block takes them instead of scanning the planes (ignored on every other path)."""
        if logits and not self.supports_logits():
            raise RuntimeError("logits=True needs the hand-written dense tail")
        if rows is not None and not self.supports_compact_queue():
            raise RuntimeError("compact queue: needs the hand-written trunk (128 filters, fused blocks)")
        if self.trunk == "mfma":
            if not planes.is_cuda:
                raise RuntimeError("trunk='mfma' is the hand-written HIP path: it has no CPU implementation")
            n, npol = planes.shape[0], self.policy_conv.out_channels
            if self.head_w32.shape[0] == 6:
                from cchess_alphazero import _native
                if rows is not None:                         # persistent buffers: the tail rows keep old finite values
                    pf, vf = self._head_feats(n, npol, planes.device)
                else:
                    pf = torch.empty((n, npol * 90), dtype=torch.float32, device=planes.device)
                    vf = torch.empty((n, (6 - npol) * 90), dtype=torch.float32, device=planes.device)
                last = self._trunk_mfma(planes, heads=(npol, pf, vf) if self.fused_heads else None,
                                        rows=rows, count=count, masks=masks)
                if last is not None:
                    _native.head_convs(last, self.head_w32, self.head_b32, npol, pf, vf)
                if self.fused_tail and pf.shape[1] in (180, 360) and vf.shape[1] in (180, 360):
                    # dense layers + softmax / tanh: three launches of hand-written kernels, straight into `out`
                    if out is None:
                        out = (torch.empty((n, self.policy_out.out_features), dtype=torch.float32, device=planes.device),
                               torch.empty((n,), dtype=torch.float32, device=planes.device))
                    key = ("tail_stats", str(planes.device))          # scratch for the rows' (max, sum): grown, never per size
                    if key not in self._bufs or self._bufs[key].shape[0] < n:
                        self._bufs[key] = torch.empty((n, 2), dtype=torch.float32, device=planes.device)
                    _native.heads_tail(pf, vf, self.tail_wp.view(self._tail_dtype), self.tail_bp,
                                       self.tail_w1.view(self._tail_dtype), self.tail_b1, self.tail_w2,
                                       self._tail_b2, out[0], out[1], self._bufs[key], count=count, normalize=not logits)
                    return out
                p = self.policy_out(pf.to(self.dtype))
                v = F.relu(self.value_dense(vf.to(self.dtype)))
                if out is not None:                            # straight into the search object's queue tensors
                    torch.softmax(p.float(), dim=1, out=out[0])
                    torch.tanh(self.value_out(v).float(), out=out[1].view(-1, 1))
                    return out
                v = torch.tanh(self.value_out(v).float())
                return F.softmax(p.float(), dim=1), v.squeeze(1)
            # (other head widths -- CChessNet(policy_filters=..., value_filters=...), keras_io: the fused input layer still
            #  reads the occupancy boards where the engine has switched the planes off: takes_masks)
            last = self._trunk_mfma(planes, rows=rows, count=count, masks=masks)
            x = last.view(n, 10, 9, self.filters).permute(0, 3, 1, 2)    # logical NCHW over channels-last memory
        else:
            x = planes.to(self.dtype).contiguous(memory_format=torch.channels_last)
            if x.is_cuda and self.fused_epilogue and self.input_conv.out_channels % 8 == 0:
                x = self._trunk_fused(x)
            else:
                x = F.relu(self.input_conv(x))
                for c1, c2 in self.res:
                    y = F.relu(c1(x))
                    x = F.relu(x + c2(y))
        p = F.relu(self.policy_conv(x))
        p = self.policy_out(p.flatten(1))                # flatten of an NCHW-shaped tensor: C,H,W order
        v = F.relu(self.value_conv(x))
        v = F.relu(self.value_dense(v.flatten(1)))
        v = torch.tanh(self.value_out(v).float())
        return F.softmax(p.float(), dim=1), v.squeeze(1)


# ---- load-time guard of the reduced tower arithmetics -------------------------------------------------------------
# north_star's tolerance (policy / value within 1e-4 of the reference network, agent/model.py:32-83 evaluated as
# api.py:63-74 does) is the only licence the split arithmetics have.  It is a property of a NETWORK, not of a kernel: a
# peaked policy amplifies the trunk's relative error by its logit scale, large activations saturate the c8 image, tiny
# ones underflow fp16's lo parts.  So every time weights enter (engine construction, hot reload, keras_io load) the
# candidate arithmetic is measured against a float64 evaluation of the same network on calibration positions and the
# first one of  c8 -> c8>N -> f16x3 -> bf16x3 -> fp32 library trunk  that stays inside GUARD_TOL is used.
GUARD_TOL = 5e-5            # half of north_star's 1e-4: the calibration set is a sample
# The search never sees the 2086-way softmax: it renormalises the priors over the LEGAL moves (reference player.py:272-283;
# the engine's logit queue does exactly that), so mass on illegal labels -- which can hide a large logit error behind a tiny
# absolute softmax error -- drops out.  With d the logit deviation up to the softmax's free constant
# (logit_max_abs: centred on its mean, so the spread max d - min d is at most 2 * logit_max_abs), a prior renormalised over
# ANY subset of the labels moves by at most  spread * p (1 - p) <= spread / 4 <= logit_max_abs / 2.  The gate below therefore
# bounds every renormalised prior's error by north_star's 1e-4 whatever the position's legal set and however peaked its
# policy is -- a property of the trunk's error and the policy layer's gain, not of the sampled positions' softmax.
LOGIT_TOL = 2e-4
CALIBRATION_POSITIONS = 256


def within_guard(m, tol=GUARD_TOL, logit_tol=LOGIT_TOL):
    """The guard's acceptance test on measure_against_reference's figures: finite, policy (full softmax) and value within
    `tol` on the calibration positions, the logit deviation within `logit_tol` (=> renormalised priors within 1e-4), and --
    where the legal moves of the positions are known -- the legal-renormalised priors themselves within `tol`."""
    return bool(m["finite"] and m["policy_max_abs"] <= tol and m["value_max_abs"] <= tol and
                m["logit_max_abs"] <= logit_tol and m.get("legal_prior_max_abs", 0.0) <= tol)


def calibration_planes(n=CALIBRATION_POSITIONS, input_depth=14, device=None, seed=20260924, with_legal=False):
    """uint8 [n, input_depth, 10, 9]: positions of random playouts from the opening, generated with the engine's own rule
    kernels (cz_movegen / cz_step / cz_done / cz_encode) -- openings, middlegames and thinned-out endgames.  For 28-plane
    (history) networks the second half is the previous position's planes (zero at the start of a game).
    with_legal: also the positions' legal moves as a bool mask [n, 2086] over the policy labels (the side to move is always
    red in the engine's board convention, and cz_movegen's moves are indices into the red label table)."""
    import numpy as np
    from cchess_alphazero import _native
    from cchess_alphazero.environment.static_env import INIT_STATE, state_to_array
    _native.require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    games = 32
    rng = np.random.default_rng(seed)
    init = torch.from_numpy(state_to_array(INIT_STATE)).to(dev)
    boards = init.repeat(games, 1).contiguous()
    prev = torch.zeros((games, 14, 10, 9), dtype=torch.uint8, device=dev)
    out, legal, plies = [], [], 0
    stride = 3                                     # keep every third ply of every game
    while sum(t.shape[0] for t in out) < n:
        planes = _native.encode(boards, _native.U8)
        moves, counts = _native.movegen(boards)
        if plies % stride == 0:
            out.append(planes if input_depth <= 14 else torch.cat([planes, prev], dim=1))
            if with_legal:
                # (a move IS its index into ActionLabelsRed, include/czero.h; the padding 0xFFFF goes to a spare column)
                live = torch.arange(moves.shape[1], device=dev)[None, :] < counts.to(torch.int64)[:, None]
                lab = torch.where(live, moves.to(torch.int64), torch.full((), _native.NLABELS, dtype=torch.int64, device=dev))
                mask = torch.zeros((games, _native.NLABELS + 1), dtype=torch.bool, device=dev)
                mask.scatter_(1, lab, True)
                legal.append(mask[:, :_native.NLABELS])
        over = _native.done(boards)[0].cpu().numpy() != 0
        cnt = counts.cpu().numpy().astype(np.int64)
        pick = (rng.random(games) * np.maximum(cnt, 1)).astype(np.int64)
        mv = moves.to(torch.int32)[torch.arange(games, device=dev), torch.from_numpy(pick).to(dev)]
        nxt, _ = _native.step(boards, mv.to(torch.uint16).contiguous())
        restart = torch.from_numpy(over | (cnt == 0) | (rng.random(games) < 0.004)).to(dev)
        boards = torch.where(restart[:, None], init[None, :], nxt).contiguous()
        prev = torch.where(restart[:, None, None, None], torch.zeros_like(planes), planes)
        plies += 1
    planes = torch.cat(out)[:n].contiguous()
    if input_depth < planes.shape[1]:
        planes = planes[:, :input_depth].contiguous()
    if with_legal:
        return planes, torch.cat(legal)[:n].contiguous()
    return planes


@torch.no_grad()
def reference_forward_f64(net: CChessNet, planes, with_activations=False):
    """The reference network (agent/model.py:32-83; BatchNorm in inference mode as Keras predict_on_batch runs it) in float64
    on the device of `planes`: (policy [n, 2086], value [n], logits) and, on request, two more entries: max |activation|
    of every tower tensor (input layer, each block's intermediate and output) and the [1 %, 50 %] quantiles of its nonzero
    values."""
    import copy
    dev = planes.device
    ref = copy.deepcopy(net).eval().double().to(dev)
    x = planes.double()
    acts, small = [], []

    def note(t):                                    # range of a tower tensor: max |x|, and the nonzero values' quantiles
        acts.append(float(t.abs().max()))
        nz = t[t > 0]
        if with_activations and nz.numel():
            q = torch.quantile(nz.flatten()[:1 << 20].float(), torch.tensor([0.01, 0.5], device=t.device))
            small.append([float(q[0]), float(q[1])])
        else:
            small.append([0.0, 0.0])

    def bn(m, t):
        scale = m.weight / torch.sqrt(m.running_var + m.eps)
        return t * scale.view(1, -1, 1, 1) + (m.bias - m.running_mean * scale).view(1, -1, 1, 1)

    def conv(m, t):                                 # unfold + matmul: float64 has no library convolution on this stack
        k = m.kernel_size[0]
        n, c, h, w = t.shape
        cols = F.unfold(t, k, padding=k // 2)       # [n, c k k, h w]
        return (m.weight.view(m.out_channels, -1) @ cols).view(n, m.out_channels, h, w)

    x = F.relu(bn(ref.input_bn, conv(ref.input_conv, x)))
    note(x)
    for blk in ref.res:
        y = F.relu(bn(blk.bn1, conv(blk.conv1, x)))
        note(y)
        x = F.relu(x + bn(blk.bn2, conv(blk.conv2, y)))
        note(x)

    p = F.relu(bn(ref.policy_bn, conv(ref.policy_conv, x)))
    logits = ref.policy_out(p.flatten(1))
    v = F.relu(bn(ref.value_bn, conv(ref.value_conv, x)))
    v = torch.tanh(ref.value_out(F.relu(ref.value_dense(v.flatten(1))))).squeeze(1)
    out = (F.softmax(logits, dim=1), v, logits)
    return out + (acts, small) if with_activations else out


def legal_priors(p, legal):
    """The priors the search consumes (reference player.py:272-283): p restricted to the legal labels and renormalised over
    them; rows without a legal move (mated positions) come back as zeros."""
    q = torch.where(legal, p, torch.zeros_like(p))
    return q / q.sum(1, keepdim=True).clamp_min(1e-300)


def measure_against_reference(inf, ref_out, planes, legal=None):
    """Max deviations of an InferenceNet from reference_forward_f64's outputs on the same planes.  legal (bool [n, 2086]):
    also legal_prior_max_abs, the deviation of the priors renormalised over each position's legal moves."""
    p, v = inf(planes)
    p, v = p.double(), v.double()
    pr, vr, lr = ref_out[:3]
    # the logit deviation up to the softmax's free constant: log p - log p_ref where both are representable
    ok = (pr > 1e-30) & (p > 1e-30)
    dl = torch.where(ok, torch.log(p.clamp_min(1e-300)) - torch.log(pr.clamp_min(1e-300)), torch.zeros_like(p))
    dl = dl - dl.sum(1, keepdim=True) / ok.sum(1, keepdim=True).clamp_min(1)
    out = dict(policy_max_abs=float((p - pr).abs().max()), value_max_abs=float((v - vr).abs().max()),
               logit_max_abs=float(torch.where(ok, dl, torch.zeros_like(dl)).abs().max()),
               finite=bool(torch.isfinite(p).all() and torch.isfinite(v).all()))
    lg = None
    if getattr(inf, "supports_logits", lambda: False)():
        # the raw logit rows the engine's queue carries: the deviation over ALL labels, centred on its mean
        lg = inf(planes, logits=True)[0].double()
        d = lg - lr.double()
        out["logit_max_abs"] = float((d - d.mean(1, keepdim=True)).abs().max())
    if legal is not None:
        # the engine's queue carries raw logits and the search kernel normalises them over the legal moves: measure exactly
        # that where the network has the logit path (a softmax over all 2086 labels in fp32 would flush the legal moves'
        # probabilities to zero when an illegal label holds the peak)
        if lg is not None:
            neg = torch.full_like(lg, -float("inf"))
            q = torch.softmax(torch.where(legal, lg, neg), dim=1)
            qr = torch.softmax(torch.where(legal, lr.double(), neg), dim=1)
            has = legal.any(1, keepdim=True)
            q, qr = torch.where(has, q, torch.zeros_like(q)), torch.where(has, qr, torch.zeros_like(qr))
        else:
            q, qr = legal_priors(p, legal), legal_priors(pr, legal)
        out["legal_prior_max_abs"] = float((q - qr).abs().max())
    return out


def choose_act_shift(activation_max, n_blocks, target=128.0, max_dev=3):
    """Power-of-two scales (exponents) for the residual stream and each block's intermediate tensor from the measured
    max |activation| of [input layer, block 0 mid, block 0 out, block 1 mid, ...].
    A COMMON shift moves the whole tower (all tensors by the same power of two: the tower's filters are untouched, only
    biases, the input layer and the head convolutions change) when its largest tensor lies outside [2^-3, 224]: e4m3
    resolves 2^-6 .. 448, and a factor of two is kept over the calibration sample.  On top of it a single tensor may
    deviate by at most 2^max_dev -- a deviation moves the factor into the neighbouring filters, and fp16 filter pairs
    lose their lo parts when scaled down far -- and only if it would otherwise still overflow.
    Networks in the usual range get (0, [0, ...]) and are left bit for bit as they were."""
    import math

    def want(m):                                    # exponent that brings a maximum m to ~target
        return int(math.floor(math.log2(target / m))) if m > 0.0 else 0
    top = max(activation_max)
    base = 0 if 2.0 ** -3 <= top <= 224.0 or not top > 0.0 else max(-40, min(40, want(top)))

    def dev(m):                                     # per-tensor deviation: only against overflow, bounded
        return max(-max_dev, min(0, want(m) - base)) if m * 2.0 ** base > 224.0 else 0
    stream = max([activation_max[0]] + [activation_max[2 * i + 2] for i in range(n_blocks)])
    return base + dev(stream), [base + dev(activation_max[2 * i + 1]) for i in range(n_blocks)]


C6_HEADROOM_BITS = 1        # see c6_exponents


def c6_exponents(activation_max, headroom=None):
    """([k_mid per block], [k_out per block]) for InferenceNet(arith="c6") from the (scaled) activation maxima
    [input layer, block 0 mid, block 0 out, ...]: the smallest k with 2^k * 28 >= 2^headroom * max (bf6's largest value is 28).
    headroom (bits kept over the calibration sample's maximum; default C6_HEADROOM_BITS): a live activation above the image's
    range saturates its bf6 value piece and the product falls back to fp16 precision for that element (graceful:
    tests/test_gpu_c6.py::test_c6_saturation_is_graceful), while every bit of headroom moves ALL small activations one binade
    towards bf6's subnormals (e3m2: two mantissa bits).  Measured (tools/c6_headroom_ab.py, profiles/r05_c6_headroom_ab.json):
    fresh positions exceed the calibration sample's maximum in 14 of 15 tower tensors, by up to 8 %; one bit of headroom moves
    the logit error of the benchmark network from 1.9e-6 / 2.2e-6 (calibration / 1024 fresh positions) to 2.2e-6 / 2.1e-6,
    two bits to 2.3e-6 / 2.4e-6 -- inside the noise: one bit is kept."""
    import math
    h = C6_HEADROOM_BITS if headroom is None else int(headroom)
    k = [int(math.ceil(math.log2(a / 28.0))) + h if a > 0.0 else 0 for a in activation_max]
    return k[1::2], k[2::2]


def guard_search(arith, c8_blocks, n_blocks, activation_max, passes):
    """The load-time guard's walk over the tower arithmetics, most reduced first, for a requested arithmetic and the
    tower's measured activation ranges; passes(name) -> bool measures one candidate (called once per name).  Returns
    (chosen or None, names in the order they were measured).

      c6 family   the request itself; if it fails, plain c8 is measured next -- a c6>k hybrid runs its remaining blocks on
                  c8 and its c6 blocks are the coarser ones, so where c8 fails the hybrids are not tried; where c8 passes,
                  c6>k for k = n6-1 .. 1, and c8 itself if none of them does
      c8 family   c8>k for k = c8_blocks .. 2, one block at a time (round 6: the stand-in for a trained network passes at
                  c8>4, which the round-5 steps N, N-2, N-4 never tried).  c8>1 only where it is the request: a tower whose
                  only c8 block is the first has no fused input launch and no chain (_trunk_mfma) -- slower than f16x3
      pairs       f16x3, then bf16x3

    A c8 image saturates above 448 (such a tower is not even tried; bf6 images carry their own exponents, so plain c6 is,
    its hybrids -- a c8 hand-over image -- are not), fp16 pairs overflow at 65504 (kept a factor of two away), bf16 pairs
    have fp32's range."""
    seen, order = {}, []

    def ok(name):
        if name not in seen:
            order.append(name)
            seen[name] = bool(passes(name))
        return seen[name]

    def hyb(fam, k):
        return fam if k >= n_blocks else f"{fam}>{k}"

    top = max(activation_max) if activation_max else 0.0
    c8_fits = top <= 448.0
    fam = arith.split(">")[0]
    if fam == "c6":
        n6 = int(arith[3:]) if ">" in arith else n_blocks
        if (n6 >= n_blocks or c8_fits) and ok(hyb("c6", n6)):
            return hyb("c6", n6), order
        if c8_fits and ok("c8"):
            for k in range(min(n6, n_blocks) - 1, 0, -1):
                if ok(hyb("c6", k)):
                    return hyb("c6", k), order
            return "c8", order
        fam, c8_blocks = "c8", n_blocks
    if fam == "c8" and c8_fits:
        for k in range(c8_blocks, min(c8_blocks, 2) - 1, -1):
            if ok(hyb("c8", k)):
                return hyb("c8", k), order
    if fam in ("c8", "f16x3") and top < 3.0e4 and ok("f16x3"):
        return "f16x3", order
    if ok("bf16x3"):
        return "bf16x3", order
    return None, order


def next_more_exact(name, n_blocks):
    """The request one step below a running arithmetic (engine.demote_arith after a failed live audit): one reduced block
    fewer, then the next family; None below bf16x3."""
    fam, _, k = name.partition(">")
    k = int(k) if k else n_blocks
    if fam == "c6":
        return f"c6>{k - 1}" if k > 1 else "c8"
    if fam == "c8":
        return f"c8>{k - 1}" if k > 2 else "f16x3"              # (c8>1: no fused launches, see guard_search)
    return "bf16x3" if fam == "f16x3" else None


def guarded_inference_net(net: CChessNet, dtype=torch.float32, trunk="mfma", arith=None, device=None, tol=GUARD_TOL,
                          planes=None, guard=None):
    """InferenceNet(net, ...) on `device` with the tower arithmetic CHECKED: the requested arithmetic is measured against
    the float64 network on calibration positions and replaced by the next more exact one while policy or value deviate by
    more than `tol` (c8 -> c8>N with fewer and fewer c8 blocks -> f16x3 -> bf16x3 -> the fp32 library trunk).  The result
    carries  .arith_requested, .arith_effective, .calibration (every candidate's measurements, the tower's activation
    ranges, the c8 image's predicted saturation / underflow counts).  guard=False (or CZ_ARITH_GUARD=0) skips the
    candidate comparison -- tests of a specific kernel path want exactly what they ask for.  (A c6 request still runs the
    float64 calibration pass then: its images' exponents come from the measured activation ranges.)"""
    from cchess_alphazero import _native
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    requested = arith or os.environ.get("CZ_TOWER_ARITH") or "bf16x3"
    if guard is None:
        guard = os.environ.get("CZ_ARITH_GUARD", "1") != "0"
    reduced = trunk == "mfma" and dtype == torch.float32        # (a caller asking for bf16 / fp16 operands asked for them)
    nblk = len(net.res)
    wants_c6 = requested == "c6" or requested.startswith("c6>")
    c6 = wants_c6 and reduced and net.cfg["cnn_filter_num"] in (128, 192) and nblk >= 2 and dev.type == "cuda"
    if wants_c6 and not c6:
        requested = "c8"                                         # (c6 exists for the 128- and 192-filter towers on the fused kernels)
    first = None if c6 else InferenceNet(net, dtype, trunk=trunk, arith=requested).to(dev)
    if first is not None:
        first.arith_requested = requested
        first.arith_effective = first.arith_name
        first.calibration = None
        if not guard or not reduced or dev.type != "cuda":
            return first
    family = "c8" if c6 else first.arith
    with torch.cuda.device(dev):
        legal = None
        if planes is None:
            planes, legal = calibration_planes(CALIBRATION_POSITIONS, net.cfg["input_depth"], dev, with_legal=True)
        ref = reference_forward_f64(net, planes, with_activations=True)
        acts = ref[3]
        report = dict(tol=tol, logit_tol=LOGIT_TOL, positions=int(planes.shape[0]),
                      max_policy_probability=float(ref[0].max()),
                      max_legal_prior=(float(legal_priors(ref[0], legal).max()) if legal is not None else None),
                      activation_max=acts, candidates=[])
        # what the c8 image would do to these activations (value byte saturates above 448; below 2^-9 it is 0; its
        # lo byte e4m3(x_lo * 2^11) saturates when |x| > ~2^9 * 448 / 2^-... i.e. with the value byte): reported, and a
        # saturating tower is not even tried
        report["c8_saturating_layers"] = [i for i, a in enumerate(acts) if a > 448.0]
        # tensors outside the range the operand formats resolve are moved into it by an exact power-of-two reparametrisation
        sx, smid = choose_act_shift(acts, nblk) if family in ("c8", "f16x3") else (0, [0] * nblk)
        shift = (sx, smid) if (sx or any(smid)) else None
        report["act_shift"] = {"stream": sx, "mid": smid}
        # (acts: [input layer, block 0 mid, block 0 out, block 1 mid, ...] -- odd entries are intermediate tensors)
        scaled = [a * 2.0 ** (smid[(i - 1) // 2] if i % 2 else sx) for i, a in enumerate(acts)]
        report["activation_max_scaled"] = scaled
        report["c8_saturating_layers_after_scaling"] = [i for i, a in enumerate(scaled) if a > 448.0]
        # the other end of the c8 image's range: e4m3 is normal down to 2^-6, subnormal (fewer bits) to 2^-9, zero below.  Per
        # tensor, after scaling: the 1 % and 50 % quantiles of the nonzero activations, and whether the MEDIAN sits in the
        # subnormals (then half of the w_lo x corrections of that layer are coarse: reported, the measurement decides)
        qs = ref[4]
        sc = [2.0 ** (smid[(i - 1) // 2] if i % 2 else sx) for i in range(len(acts))]
        report["activation_quantiles_scaled"] = [[q[0] * f, q[1] * f] for q, f in zip(qs, sc)]
        report["c8_median_in_subnormals"] = [i for i, (q, f) in enumerate(zip(qs, sc)) if 0.0 < q[1] * f < 2.0 ** -6]
        exps = c6_exponents(scaled)
        if c6:
            report["c6_exponents"] = {"mid": exps[0], "out": exps[1]}
            first = InferenceNet(net, dtype, trunk=trunk, arith=requested, act_shift=shift, act_exps=exps).to(dev)
            first.arith_requested = requested
            first.arith_effective = first.arith_name
            first.calibration = report
            if not guard:
                return first
        built = {first.arith_name: first} if (shift is None or c6) else {}

        def passes(name):
            # (one candidate alive at a time besides the requested one: packed filters of seven arithmetics add up)
            cand = built.pop(name, None)
            if cand is None:
                cand = InferenceNet(net, dtype, trunk=trunk, arith=name, act_shift=shift,
                                    act_exps=exps if name.startswith("c6") else None).to(dev)
            m = measure_against_reference(cand, ref, planes, legal)
            m["arith"] = name
            report["candidates"].append(m)
            if within_guard(m, tol):
                built[name] = cand
                return True
            return False

        name, tried = guard_search(requested if c6 else first.arith, first.c8_blocks, nblk, scaled, passes)
        if name is None:
            logger.warning("no split arithmetic keeps this network within %g of float64 (%s): using the fp32 library trunk",
                           tol, report["candidates"])
            cand = InferenceNet(net, dtype, trunk="library").to(dev)
            name = "fp32-library"
        else:
            cand = built[name]
        built.clear()
        report["tried"] = tried + (["fp32-library"] if name == "fp32-library" else [])
        report["chosen"] = next((m for m in report["candidates"] if m["arith"] == name), None)
        if name != first.arith_name:
            logger.warning("tower arithmetic %s deviates from the float64 network by more than %g (policy / value / legal "
                           "priors) or %g (logits) on the calibration positions (%s): using %s", requested, tol, LOGIT_TOL,
                           report["candidates"][0], name)
        cand.arith_requested = requested
        cand.arith_effective = name
        cand.calibration = report
    return cand


def flops_per_position(cfg):
    """Multiply-accumulate based FLOPs (2*MAC) of one forward pass."""
    f, k1, k, n = cfg["cnn_filter_num"], cfg["cnn_first_filter_size"], cfg["cnn_filter_size"], cfg["res_layer_num"]
    mac = cfg["input_depth"] * f * k1 * k1 * 90 + 2 * n * f * f * k * k * 90
    mac += f * cfg["policy_filters"] * 90 + cfg["policy_filters"] * 90 * cfg.get("n_labels", N_LABELS)
    mac += f * cfg["value_filters"] * 90 + cfg["value_filters"] * 90 * cfg["value_fc_size"] + cfg["value_fc_size"]
    return 2 * mac


class CChessModel:
    """The reference's model holder (agent/model.py): build / load / save / digest, torch-backed.
    Weight files are torch state-dicts (``.pt``); the Keras ``.h5`` blobs of the reference are not
    shipped (``.MISSING_LARGE_BLOBS``) and h5py is not available here."""

    def __init__(self, config):
        self.config = config
        self.model = None           # CChessNet
        self.digest = None
        self.n_labels = N_LABELS
        self.api = None

    def build(self, seed=None):
        if seed is not None:
            torch.manual_seed(seed)
        self.model = CChessNet.from_model_config(self.config.model)
        return self.model

    @staticmethod
    def fetch_digest(weight_path):
        if os.path.exists(weight_path):
            m = hashlib.sha256()
            with open(weight_path, "rb") as f:
                m.update(f.read())
            return m.hexdigest()
        return None

    @staticmethod
    def _pt(path):
        return os.path.splitext(path)[0] + ".pt"

    @classmethod
    def weight_file(cls, config_path, weight_path):
        """The file load() would read for this pair: the Keras HDF5 for a Keras topology JSON, else the torch .pt"""
        try:
            with open(config_path, "rt") as f:
                cfg = json.load(f)
            if isinstance(cfg, dict) and ("layers" in cfg or "layers" in cfg.get("config", {})):
                return weight_path
        except (OSError, ValueError):
            pass
        return cls._pt(weight_path)

    def load(self, config_path, weight_path):
        """Two on-disk forms: this package's own (JSON of CChessNet keyword arguments + torch state dict ``.pt``) and
        the reference's (Keras ``get_config()`` JSON + ``save_weights`` HDF5, agent/model.py:95-107), read without
        Keras / h5py by lib/keras_io.py."""
        if not os.path.exists(config_path):
            logger.debug(f"model files does not exist at {config_path}")
            return False
        with open(config_path, "rt") as f:
            cfg = json.load(f)
        if isinstance(cfg, dict) and ("layers" in cfg or "layers" in cfg.get("config", {})):
            if not os.path.exists(weight_path):
                logger.debug(f"model files does not exist at {weight_path}")
                return False
            from cchess_alphazero.lib import keras_io
            self.model = CChessNet(**keras_io.config_from_keras(cfg))
            keras_io.load_keras_weights(self.model, weight_path, keras_io.names_from_keras(cfg))
            self.digest = self.fetch_digest(weight_path)
            logger.debug(f"loaded Keras model digest = {self.digest}")
            return True
        wp = self._pt(weight_path)
        if os.path.exists(wp):
            self.model = CChessNet(**cfg)
            self.model.load_state_dict(torch.load(wp, map_location="cpu"))
            self.digest = self.fetch_digest(wp)
            logger.debug(f"loaded model digest = {self.digest}")
            return True
        logger.debug(f"model files does not exist at {config_path} and {wp}")
        return False

    def save(self, config_path, weight_path):
        wp = self._pt(weight_path)
        os.makedirs(os.path.dirname(config_path), exist_ok=True)
        with open(config_path, "wt") as f:
            json.dump(self.model.cfg, f)
        torch.save(self.model.state_dict(), wp)
        self.digest = self.fetch_digest(wp)

    def get_pipes(self, num=1, api=None, need_reload=True):
        from cchess_alphazero.agent.api import CChessModelAPI
        if self.api is None:
            self.api = CChessModelAPI(self.config, self)
            self.api.start(need_reload)
        return self.api.get_pipe(need_reload)

    def close_pipes(self):
        if self.api is not None:
            self.api.close()
            self.api = None
