"""ORACLE (test infrastructure, never on the product path): fp32 CPU restatement of
the reference network forward ``HMDEgoPose.forward`` as one functional pass over a
plain ``state_dict``.

It is a floating-point convolutional network, so the restatement uses
``torch.nn.functional`` fp32 CPU ops (the tier's "torch fp32 reference for a
floating-point kernel"); the integer/index half of the path lives in
``oracle/decode_ref.py`` (numpy).  Written from SURVEY.md section 8a / Appendix C;
each function cites the reference lines it follows.

PINNED: ``tests/golden/make_golden.py`` imports the real reference in the build
container, runs both on identical seeded weights/inputs and commits
slices + sums of every output and of every stage boundary; ``tests/test_oracle_golden.py``
replays them (tolerance 1e-5 abs on O(1) activations).
"""
from __future__ import annotations

import math
from typing import Dict, List, Mapping, Tuple

import torch
import torch.nn.functional as F

BN_EPS = 1e-3
FUSION_EPS = 1e-4
NUM_ANCHORS = 9

_SCALING = [(1.0, 1.0), (1.0, 1.1), (1.1, 1.2), (1.2, 1.4), (1.4, 1.8), (1.6, 2.2), (1.8, 2.6), (2.0, 3.1)]
_BACKBONE_OF_PHI = [0, 1, 2, 3, 4, 5, 6, 6, 7]
_FPN_REPEATS = [3, 4, 5, 6, 7, 7, 8, 8, 8]
_HEAD_DEPTH = [3, 3, 3, 4, 4, 4, 5, 5, 5]
_STAGES = [(1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80),
           (3, 5, 1, 6, 80, 112), (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320)]


def swish(x: torch.Tensor) -> torch.Tensor:
    """efficientnet/utils.py:57-59 (onnx_export=True everywhere => plain x*sigmoid(x))."""
    return x * torch.sigmoid(x)


def _same_pad(x: torch.Tensor, k: int, s: int) -> torch.Tensor:
    """Zero padding of efficientnet/utils_extra.py:33-44 (conv) and :72-83 (max-pool):
    extra=(ceil(n/s)-1)*s-n+k, before=extra//2, after=extra-before; width then height."""
    h, w = x.shape[-2:]
    eh = (math.ceil(w / s) - 1) * s - w + k
    ev = (math.ceil(h / s) - 1) * s - h + k
    l, t = eh // 2, ev // 2
    return F.pad(x, [l, eh - l, t, ev - t])


def conv_same(x, w, b=None, stride=1, groups=1):
    """Conv2dStaticSamePadding.forward, efficientnet/utils_extra.py:33-47."""
    return F.conv2d(_same_pad(x, w.shape[-1], stride), w, b, stride=stride, groups=groups)


def maxpool_same(x):
    """MaxPool2dStaticSamePadding(3, 2).forward, utils_extra.py:72-86: padded with
    ZEROS (F.pad default), not -inf, so edge windows take max(..., 0)."""
    return F.max_pool2d(_same_pad(x, 3, 2), 3, 2)


def bn(sd: Mapping[str, torch.Tensor], p: str, x: torch.Tensor) -> torch.Tensor:
    """nn.BatchNorm2d in eval mode, eps 1e-3 (efficientnet/model.py:32-33; efficientdet/model.py:36)."""
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        training=False, eps=BN_EPS)


def _round_width(c, mult):
    c2 = c * mult
    r = max(8, int(c2 + 4) // 8 * 8)
    return int(r + 8 if r < 0.9 * c2 else r)


def block_table(phi: int) -> List[dict]:
    """MBConv block list: efficientnet/utils.py:231-257 strings scaled by utils.py:62-82 and
    expanded by efficientnet/model.py:144-160."""
    wm, dm = _SCALING[_BACKBONE_OF_PHI[phi]]
    out = []
    for (r, k, s, e, i, o) in _STAGES:
        ci, co = _round_width(i, wm), _round_width(o, wm)
        for j in range(int(math.ceil(dm * r))):
            out.append(dict(k=k, s=s if j == 0 else 1, e=e, cin=ci if j == 0 else co, cout=co,
                            skip=j > 0))   # first block of a stage: stride is a list -> never adds (model.py:100)
    return out


def mbconv(sd, p: str, blk: dict, x: torch.Tensor) -> torch.Tensor:
    """MBConvBlock.forward, efficientnet/model.py:69-104."""
    inp = x
    if blk["e"] != 1:
        x = swish(bn(sd, p + "._bn0", conv_same(x, sd[p + "._expand_conv.conv.weight"])))
    wdw = sd[p + "._depthwise_conv.conv.weight"]
    x = swish(bn(sd, p + "._bn1", conv_same(x, wdw, stride=blk["s"], groups=wdw.shape[0])))
    sq = F.adaptive_avg_pool2d(x, 1)
    sq = swish(F.conv2d(sq, sd[p + "._se_reduce.conv.weight"], sd[p + "._se_reduce.conv.bias"]))
    sq = F.conv2d(sq, sd[p + "._se_expand.conv.weight"], sd[p + "._se_expand.conv.bias"])
    x = torch.sigmoid(sq) * x
    x = bn(sd, p + "._bn2", conv_same(x, sd[p + "._project_conv.conv.weight"]))
    if blk["skip"]:
        x = x + inp
    return x


def backbone(sd, phi: int, x: torch.Tensor, trace: dict | None = None):
    """EfficientNet wrapper forward, efficientdet/model.py:436-458, + the extra drop at
    backbone.py:107: P3/P4/P5 are the last three taps."""
    bb = "backbone_net.model"
    x = swish(bn(sd, bb + "._bn0", conv_same(x, sd[bb + "._conv_stem.conv.weight"], stride=2)))
    if trace is not None:
        trace["stem"] = x
    taps, last = [], None
    blocks = block_table(phi)
    for i, blk in enumerate(blocks):
        x = mbconv(sd, f"{bb}._blocks.{i}", blk, x)
        if trace is not None:
            trace[f"block{i}"] = x
        if blk["s"] == 2:
            taps.append(last)
        elif i == len(blocks) - 1:
            taps.append(x)
        last = x
    return taps[-3:]


def sepconv(sd, p: str, x: torch.Tensor, norm: bool) -> torch.Tensor:
    """SeparableConvBlock.forward, efficientdet/model.py:42-52 (dw no bias, pw with bias)."""
    wdw = sd[p + ".depthwise_conv.conv.weight"]
    x = conv_same(x, wdw, groups=wdw.shape[0])
    x = F.conv2d(x, sd[p + ".pointwise_conv.conv.weight"], sd[p + ".pointwise_conv.conv.bias"])
    return bn(sd, p + ".bn", x) if norm else x


def _lateral(sd, p: str, x):
    return bn(sd, p + ".1", F.conv2d(x, sd[p + ".0.conv.weight"], sd[p + ".0.conv.bias"]))


def _fw(sd, key: str, attention: bool, n: int):
    """relu(w)/(sum(relu(w))+1e-4), efficientdet/model.py:212-213; plain sum for phi>=6 (:268-341)."""
    if not attention:
        return [1.0] * n
    w = F.relu(sd[key])
    w = w / (w.sum() + FUSION_EPS)
    return [w[i] for i in range(n)]


def bifpn_cell(sd, p: str, feats, first: bool, attention: bool):
    """BiFPN._forward_fast_attention, efficientdet/model.py:194-266."""
    up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")
    if first:
        p3, p4, p5 = feats
        p6_in = maxpool_same(_lateral(sd, p + ".p5_to_p6", p5))
        p7_in = maxpool_same(p6_in)
        p3_in = _lateral(sd, p + ".p3_down_channel", p3)
        p4_in = _lateral(sd, p + ".p4_down_channel", p4)
        p5_in = _lateral(sd, p + ".p5_down_channel", p5)
    else:
        p3_in, p4_in, p5_in, p6_in, p7_in = feats
    w = _fw(sd, p + ".p6_w1", attention, 2)
    p6_up = sepconv(sd, p + ".conv6_up", swish(w[0] * p6_in + w[1] * up(p7_in)), True)
    w = _fw(sd, p + ".p5_w1", attention, 2)
    p5_up = sepconv(sd, p + ".conv5_up", swish(w[0] * p5_in + w[1] * up(p6_up)), True)
    w = _fw(sd, p + ".p4_w1", attention, 2)
    p4_up = sepconv(sd, p + ".conv4_up", swish(w[0] * p4_in + w[1] * up(p5_up)), True)
    w = _fw(sd, p + ".p3_w1", attention, 2)
    p3_out = sepconv(sd, p + ".conv3_up", swish(w[0] * p3_in + w[1] * up(p4_up)), True)
    if first:   # a SECOND lateral projection feeds the bottom-up path (model.py:236-237)
        p4_in = _lateral(sd, p + ".p4_down_channel_2", p4)
        p5_in = _lateral(sd, p + ".p5_down_channel_2", p5)
    w = _fw(sd, p + ".p4_w2", attention, 3)
    p4_out = sepconv(sd, p + ".conv4_down", swish(w[0] * p4_in + w[1] * p4_up + w[2] * maxpool_same(p3_out)), True)
    w = _fw(sd, p + ".p5_w2", attention, 3)
    p5_out = sepconv(sd, p + ".conv5_down", swish(w[0] * p5_in + w[1] * p5_up + w[2] * maxpool_same(p4_out)), True)
    w = _fw(sd, p + ".p6_w2", attention, 3)
    p6_out = sepconv(sd, p + ".conv6_down", swish(w[0] * p6_in + w[1] * p6_up + w[2] * maxpool_same(p5_out)), True)
    w = _fw(sd, p + ".p7_w2", attention, 2)
    p7_out = sepconv(sd, p + ".conv7_down", swish(w[0] * p7_in + w[1] * maxpool_same(p6_out)), True)
    return p3_out, p4_out, p5_out, p6_out, p7_out


def num_classes(sd) -> int:
    """Classes of the classifier: its header has num_anchors * num_classes channels (efficientdet/model.py:393)."""
    return int(sd["classifier.header.pointwise_conv.conv.weight"].shape[0]) // NUM_ANCHORS


def head(sd, name: str, depth: int, feats, headers: List[Tuple[str, int]], sigmoid=False) -> torch.Tensor:
    """Regressor/Classifier.forward efficientdet/model.py:361-417 and RotationNet/TranslationNet/
    HandNet.forward hmdegopose/model.py:55-90,127-156,191-228 with num_iteration_steps == 0:
    shared conv_list, per-level bn_list, header(s); NHWC flatten -> [B, HW*9, K]; levels concatenated;
    two headers (translation xy|z) are concatenated per anchor (model.py:214-220)."""
    outs = []
    for lvl, f in enumerate(feats):
        for i in range(depth):
            f = swish(bn(sd, f"{name}.bn_list.{lvl}.{i}", sepconv(sd, f"{name}.conv_list.{i}", f, False)))
        parts = []
        for hname, k in headers:
            y = sepconv(sd, f"{name}.{hname}", f, False).permute(0, 2, 3, 1).contiguous()
            parts.append(y.view(y.shape[0], -1, k))
        outs.append(parts[0] if len(parts) == 1 else torch.cat(parts, dim=2))
    y = torch.cat(outs, dim=1)
    return y.sigmoid() if sigmoid else y


@torch.no_grad()
def forward(sd: Mapping[str, torch.Tensor], x: torch.Tensor, phi: int, trace: Dict[str, torch.Tensor] | None = None):
    """HMDEgoPose.forward, backbone.py:104-125.  x: fp32 [B,3,S,S] (any strides).
    Returns (features(5), regression[B,N,4], classification[B,N,num_classes] (post-sigmoid),
    rotation[B,N,3], translation_raw[B,N,3], hand[B,N,63]).  num_classes is what the classifier's header holds
    (efficientdet/model.py:393: num_anchors * num_classes output channels, reshaped per anchor at :406-408)."""
    attention = phi < 6
    feats = backbone(sd, phi, x.float(), trace)
    if trace is not None:
        trace["p3"], trace["p4"], trace["p5"] = feats
    for r in range(_FPN_REPEATS[phi]):
        feats = bifpn_cell(sd, f"bifpn.{r}", feats, first=(r == 0), attention=attention)
        if trace is not None:
            for l, f in enumerate(feats):
                trace[f"bifpn{r}_p{l + 3}"] = f
    d = _HEAD_DEPTH[phi]
    A = NUM_ANCHORS
    regression = head(sd, "regressor", d, feats, [("header", 4)])
    classification = head(sd, "classifier", d, feats, [("header", num_classes(sd))], sigmoid=True)
    rotation = head(sd, "rotation_net", d, feats, [("initial_rotation", 3)])
    translation = head(sd, "translation_net", d, feats, [("initial_translation_xy", 2), ("initial_translation_z", 1)])
    hand = head(sd, "hand_net", d, feats, [("initial_hand_coords", 63)])
    return feats, regression, classification, rotation, translation, hand


# ======================================================================================
# Storage-emulating variant (still TEST INFRASTRUCTURE): the same network with BatchNorm
# folded into the preceding convolution and explicit quantisation points, so that a
# reduced-precision device path (bf16 storage / fp8 pointwise operands, fp32 accumulate)
# can be gated against an oracle that rounds where the device rounds instead of against
# the fp32 oracle at a loose bound.  With every hook = identity it is the SAME function
# as ``forward`` up to fp32 summation order (tests/test_oracle_golden.py pins that at
# 2e-5), so the golden vectors of the real reference pin this restatement as well.
#
# Rounding points (what a bf16 session of libhep.so stores, hmd_ego_pose_amd/csrc):
#   q_act  every activation tensor that is stored (HBM or LDS): stem out, expand out,
#          depthwise out (the SE mean is taken BEFORE that rounding, in fp32), the
#          SE-scaled project operand, block out (after the residual add), lateral out,
#          the fused+swished BiFPN node input, every separable conv's depthwise result,
#          BiFPN node / tower layer outputs.  Head outputs stay fp32.
#   q_w    pointwise (1x1) weights AFTER the BN scale is folded in, and the squeeze-excite
#          expand FC weight (read by every project workgroup).  Depthwise, stem, SE reduce
#          weights, biases and fusion weights stay fp32.
#   q_pw   (fp8 sessions) replaces the backbone pointwise convs (expand / project): called as
#          q_pw(x, w_folded, bias, tag) with tag "b{i}.expand" / "b{i}.project"; see
#          ``make_q_pw_fp8`` (e4m3 operands, per-output-channel weight scale, per-tensor
#          power-of-two activation scale taken from the device's calibration).
# ======================================================================================
def q_bf16(t: torch.Tensor) -> torch.Tensor:
    """Round to nearest-even bf16 and back (what v_cvt_pk_bf16_f32 does)."""
    return t.to(torch.bfloat16).to(torch.float32)


def _fold(sd, p: str):
    """scale = gamma / sqrt(var + eps), shift = beta - mean * scale, in fp32 like the plan builder."""
    s = sd[p + ".weight"] / torch.sqrt(sd[p + ".running_var"] + BN_EPS)
    return s, sd[p + ".bias"] - sd[p + ".running_mean"] * s


class _Emu:
    def __init__(self, q_act=None, q_w=None, q_pw=None):
        ident = lambda t: t
        self.qa = q_act or ident
        self.qw = q_w or ident
        self.q_pw = q_pw

    def pw(self, x, w, bias, tag=None):
        """1x1 conv with an already folded weight [N,K,1,1]; tag names a backbone conv (quantised in fp8 sessions)."""
        if tag is not None and self.q_pw is not None:
            return self.q_pw(x, w, bias, tag)
        return F.conv2d(x, self.qw(w), bias)


E4M3_MAX = 448.0


def q_e4m3(t: torch.Tensor) -> torch.Tensor:
    """Round to OCP e4m3fn (nearest even, saturating at +-448 like the device conversion) and back."""
    return torch.clamp(t, -E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).to(torch.float32)


def make_q_pw_fp8(act_scales: Mapping[str, float]):
    """The e4m3 pointwise conv of an fp8 device session (hmd_ego_pose_amd/csrc/k_pw_impl.h, k_mbf.hip): weights
    w / sw[n] with sw[n] = max_k |w[n, k]| / 448 per output channel, activations x / sa with the power-of-two
    per-tensor scale ``act_scales[tag]`` the device calibrated (Session.fp8_scales(): a fused front launch
    "b{i}.front" carries the scale of its expand conv), exact products, fp32 accumulation, then * sa * sw[n] + bias."""
    def q_pw(x, w, bias, tag):
        sa = float(act_scales.get(tag, act_scales.get(tag.replace(".expand", ".front"), 0.0)))
        if sa <= 0:
            raise KeyError(f"no fp8 activation scale for {tag}")
        amax = w.abs().amax(dim=(1, 2, 3), keepdim=True)
        sw = torch.where(amax > 0, amax / E4M3_MAX, torch.ones_like(amax))
        y = F.conv2d(q_e4m3(x / sa), q_e4m3(w / sw)) * (sa * sw.view(1, -1, 1, 1))
        return y + bias.view(1, -1, 1, 1)
    return q_pw


def _mbconv_emu(E: _Emu, sd, p: str, blk: dict, x):
    inp = x
    tag = "b" + p.rsplit(".", 1)[-1]
    if blk["e"] != 1:
        s0, b0 = _fold(sd, p + "._bn0")
        x = E.qa(swish(E.pw(x, sd[p + "._expand_conv.conv.weight"] * s0[:, None, None, None], b0, tag + ".expand")))
    s1, b1 = _fold(sd, p + "._bn1")
    wdw = sd[p + "._depthwise_conv.conv.weight"] * s1[:, None, None, None]
    v = swish(conv_same(x, wdw, b1, stride=blk["s"], groups=wdw.shape[0]))
    sq = F.adaptive_avg_pool2d(v, 1)                                   # fp32 mean of the un-rounded values
    sq = swish(F.conv2d(sq, sd[p + "._se_reduce.conv.weight"], sd[p + "._se_reduce.conv.bias"]))
    sq = torch.sigmoid(F.conv2d(sq, E.qw(sd[p + "._se_expand.conv.weight"]), sd[p + "._se_expand.conv.bias"]))
    a = E.qa(v) * sq                                                   # stored depthwise output x scale -> GEMM operand
    if E.q_pw is None:
        a = E.qa(a)                                                    # (fp8 sessions convert the fp32 product straight to e4m3)
    s2, b2 = _fold(sd, p + "._bn2")
    y = E.pw(a, sd[p + "._project_conv.conv.weight"] * s2[:, None, None, None], b2, tag + ".project")
    if blk["skip"]:
        y = y + inp
    return E.qa(y)


def _sepconv_emu(E: _Emu, sd, p: str, x, bnkey: str | None):
    wdw = sd[p + ".depthwise_conv.conv.weight"]
    d = E.qa(conv_same(x, wdw, groups=wdw.shape[0]))
    w, b = sd[p + ".pointwise_conv.conv.weight"], sd[p + ".pointwise_conv.conv.bias"]
    if bnkey is not None:
        s, sh = _fold(sd, bnkey)
        w, b = w * s[:, None, None, None], b * s + sh
    return E.pw(d, w, b)


def _lateral_emu(E: _Emu, sd, p: str, x):
    s, sh = _fold(sd, p + ".1")
    return E.qa(E.pw(x, sd[p + ".0.conv.weight"] * s[:, None, None, None], sd[p + ".0.conv.bias"] * s + sh))


def _bifpn_cell_emu(E: _Emu, sd, p: str, feats, first: bool, attention: bool):
    up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")

    def node(conv, wkey, srcs):
        w = _fw(sd, f"{p}.{wkey}", attention, len(srcs))
        acc = w[0] * srcs[0]
        for wi, si in zip(w[1:], srcs[1:]):
            acc = acc + wi * si
        return E.qa(_sepconv_emu(E, sd, f"{p}.{conv}", E.qa(swish(acc)), f"{p}.{conv}.bn"))

    if first:
        p3, p4, p5 = feats
        p6_in = maxpool_same(_lateral_emu(E, sd, p + ".p5_to_p6", p5))
        p7_in = maxpool_same(p6_in)
        p3_in, p4_in, p5_in = (_lateral_emu(E, sd, f"{p}.p{l}_down_channel", t) for l, t in ((3, p3), (4, p4), (5, p5)))
    else:
        p3_in, p4_in, p5_in, p6_in, p7_in = feats
    p6_up = node("conv6_up", "p6_w1", [p6_in, up(p7_in)])
    p5_up = node("conv5_up", "p5_w1", [p5_in, up(p6_up)])
    p4_up = node("conv4_up", "p4_w1", [p4_in, up(p5_up)])
    p3_out = node("conv3_up", "p3_w1", [p3_in, up(p4_up)])
    if first:
        p4_in = _lateral_emu(E, sd, p + ".p4_down_channel_2", p4)
        p5_in = _lateral_emu(E, sd, p + ".p5_down_channel_2", p5)
    p4_out = node("conv4_down", "p4_w2", [p4_in, p4_up, maxpool_same(p3_out)])
    p5_out = node("conv5_down", "p5_w2", [p5_in, p5_up, maxpool_same(p4_out)])
    p6_out = node("conv6_down", "p6_w2", [p6_in, p6_up, maxpool_same(p5_out)])
    p7_out = node("conv7_down", "p7_w2", [p7_in, maxpool_same(p6_out)])
    return p3_out, p4_out, p5_out, p6_out, p7_out


def _head_emu(E: _Emu, sd, name: str, depth: int, feats, headers, sigmoid=False):
    outs = []
    for lvl, f in enumerate(feats):
        for i in range(depth):
            f = E.qa(swish(_sepconv_emu(E, sd, f"{name}.conv_list.{i}", f, f"{name}.bn_list.{lvl}.{i}")))
        parts = []
        for hname, k in headers:
            y = _sepconv_emu(E, sd, f"{name}.{hname}", f, None).permute(0, 2, 3, 1).contiguous()
            parts.append(y.view(y.shape[0], -1, k))
        outs.append(parts[0] if len(parts) == 1 else torch.cat(parts, dim=2))
    y = torch.cat(outs, dim=1)
    return y.sigmoid() if sigmoid else y


def emulated_stages(sd: Mapping[str, torch.Tensor], phi: int, q_act=q_bf16, q_w=q_bf16, q_pw=None):
    """The emulated network cut into its stages, for TEACHER-FORCED parity: a reduced-precision network with
    generic weights is chaotic with respect to its own rounding (a 1e-6 relative perturbation in front of the
    bf16 rounding of the stem grows to the full bf16 drift of ~3 % within five blocks: two correct bf16
    realisations differ as much from each other as from fp32), so end-to-end comparison cannot be tight.  Feeding
    every stage the DEVICE's own input and comparing that stage's output can: what remains is fp32 summation
    order flipping an occasional rounding inside one stage.  Returns a dict of callables:
      stem(x) -> y;  block(i, y) -> y;  cell(r, feats) -> feats;  heads(feats) -> (reg, cls, rot, trn, hand);
      taps = indices of the blocks whose outputs are P3, P4, P5."""
    E = _Emu(q_act, q_w, q_pw)
    attention = phi < 6
    bb = "backbone_net.model"
    blocks = block_table(phi)
    taps, last = [], None
    for i, blk in enumerate(blocks):
        if blk["s"] == 2:
            taps.append(last)
        elif i == len(blocks) - 1:
            taps.append(i)
        last = i
    d = _HEAD_DEPTH[phi]

    @torch.no_grad()
    def stem(x):
        s, sh = _fold(sd, bb + "._bn0")
        return E.qa(swish(conv_same(x.float(), sd[bb + "._conv_stem.conv.weight"] * s[:, None, None, None], sh, stride=2)))

    @torch.no_grad()
    def block(i, y):
        return _mbconv_emu(E, sd, f"{bb}._blocks.{i}", blocks[i], y)

    @torch.no_grad()
    def cell(r, feats):
        return _bifpn_cell_emu(E, sd, f"bifpn.{r}", feats, r == 0, attention)

    @torch.no_grad()
    def heads(feats):
        return (_head_emu(E, sd, "regressor", d, feats, [("header", 4)]),
                _head_emu(E, sd, "classifier", d, feats, [("header", num_classes(sd))], sigmoid=True),
                _head_emu(E, sd, "rotation_net", d, feats, [("initial_rotation", 3)]),
                _head_emu(E, sd, "translation_net", d, feats, [("initial_translation_xy", 2), ("initial_translation_z", 1)]),
                _head_emu(E, sd, "hand_net", d, feats, [("initial_hand_coords", 63)]))

    return dict(stem=stem, block=block, cell=cell, heads=heads, taps=taps[-3:], n_blocks=len(blocks), n_cells=_FPN_REPEATS[phi])


@torch.no_grad()
def forward_emulated(sd: Mapping[str, torch.Tensor], x: torch.Tensor, phi: int, trace: Dict[str, torch.Tensor] | None = None,
                     q_act=q_bf16, q_w=q_bf16, q_pw=None):
    """``forward`` with BN folded and the storage rounding of a reduced-precision device session
    (see the block comment above).  Defaults emulate a bf16 session; q_act=q_w=None is fp32."""
    st = emulated_stages(sd, phi, q_act, q_w, q_pw)
    y = st["stem"](x)
    if trace is not None:
        trace["stem"] = y
    outs = []
    for i in range(st["n_blocks"]):
        y = st["block"](i, y)
        outs.append(y)
        if trace is not None:
            trace[f"block{i}"] = y
    feats = [outs[t] for t in st["taps"]]
    for r in range(st["n_cells"]):
        feats = st["cell"](r, feats)
        if trace is not None:
            for l, f in enumerate(feats):
                trace[f"bifpn{r}_p{l + 3}"] = f
    return (feats, *st["heads"](feats))
