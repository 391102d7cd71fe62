"""Architecture tables for the EfficientPose (HMD-EgoPose) inference path.

Host-side description of *what* the network is for a compound coefficient
``phi``: the MBConv block list, BiFPN width/repeats, head depth, pyramid
geometry and the ordered parameter inventory (names + shapes) that the
reference's ``state_dict`` uses.  ``csrc/hep_model.cpp`` holds the same tables
in C++ (the C-ABI library is self-contained); ``tests/test_host_cpu.py`` checks
this table against the golden key list captured from the reference, and the GPU
parity tests exercise the C++ copy.

Reference behaviour restated here (not copied):
  * per-phi tables ............ pytorch-sandbox/backbone.py:22-43
  * block strings / scaling ... pytorch-sandbox/efficientnet/utils.py:62-82,138-153,231-257
  * block expansion ........... pytorch-sandbox/efficientnet/model.py:29-67,144-160
  * backbone taps ............. pytorch-sandbox/efficientdet/model.py:436-458
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Iterator, List, Tuple

# (width, depth) multipliers of EfficientNet-B0..B7 (efficientnet/utils.py:138-153)
_EFFNET_SCALING = [(1.0, 1.0), (1.0, 1.1), (1.1, 1.2), (1.2, 1.4),
                   (1.4, 1.8), (1.6, 2.2), (1.8, 2.6), (2.0, 3.1)]
# B0 stage table: repeats, kernel, stride, expand, in, out (efficientnet/utils.py:235-240)
_B0_STAGES = [(1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40),
              (3, 3, 2, 6, 40, 80), (3, 5, 1, 6, 80, 112), (4, 5, 2, 6, 112, 192),
              (1, 3, 1, 6, 192, 320)]
_SE_RATIO = 0.25
BN_EPS = 1e-3            # every BatchNorm on the path (efficientnet/utils.py:245; efficientdet/model.py:36)
FUSION_EPS = 1e-4        # BiFPN fast-attention epsilon (efficientdet/model.py:60)

# backbone.py:22-29
_BACKBONE_OF_PHI = [0, 1, 2, 3, 4, 5, 6, 6, 7]
_FPN_WIDTH = [64, 88, 112, 160, 224, 288, 384, 384, 384]
_FPN_REPEATS = [3, 4, 5, 6, 7, 7, 8, 8, 8]
_HEAD_DEPTH = [3, 3, 3, 4, 4, 4, 5, 5, 5]

NUM_ANCHORS = 9          # 3 ratios x 3 scales (backbone.py:30-31,45)
MAX_CLASSES = 63         # the classifier header then is as wide as the hand header (9 * 63 channels): the widest the head kernels are built for
NUM_LEVELS = 5           # P3..P7 (phi 8 adds P8 and is not supported here)
HEAD_NAMES = ("regressor", "classifier", "rotation_net", "translation_net", "hand_net")
OUT_NAMES = ("regression", "classification", "rotation", "translation_raw", "hand")
OUT_WIDTH = (4, 1, 3, 3, 63)   # values per anchor of the five [B, N, K] outputs


def _round_width(c: int, mult: float) -> int:
    """Channel rounding to a multiple of 8, never shrinking by more than 10 %."""
    c2 = c * mult
    r = max(8, int(c2 + 4) // 8 * 8)
    if r < 0.9 * c2:
        r += 8
    return int(r)


def _round_depth(r: int, mult: float) -> int:
    return int(math.ceil(mult * r))


@dataclass(frozen=True)
class MBConv:
    cin: int
    cexp: int
    k: int
    stride: int
    se: int          # squeeze width
    cout: int
    expand: bool     # has the 1x1 expand conv (+BN+swish)
    skip: bool       # residual add executed


@dataclass(frozen=True)
class Arch:
    phi: int
    stem: int
    blocks: Tuple[MBConv, ...]
    taps: Tuple[int, int, int]        # block indices whose outputs are P3, P4, P5
    tap_channels: Tuple[int, int, int]
    fpn_w: int
    fpn_cells: int
    head_depth: int
    attention: bool                   # fast-attention BiFPN (phi < 6)


def get_arch(phi: int) -> Arch:
    if not 0 <= phi <= 7:
        raise ValueError(f"compound_coef {phi} is not supported by the MI355X path (0..7; phi 8 needs P8)")
    wmul, dmul = _EFFNET_SCALING[_BACKBONE_OF_PHI[phi]]
    blocks: List[MBConv] = []
    stride2_at: List[int] = []
    for (r, k, s, e, i, o) in _B0_STAGES:
        cin, cout, reps = _round_width(i, wmul), _round_width(o, wmul), _round_depth(r, dmul)
        for j in range(reps):
            bi = cin if j == 0 else cout
            bs = s if j == 0 else 1
            # the first block of a stage never adds its input (its stride is a *list* in
            # the reference, efficientnet/model.py:100 + utils.py:184); later ones do.
            blocks.append(MBConv(cin=bi, cexp=bi * e, k=k, stride=bs,
                                 se=max(1, int(bi * _SE_RATIO)), cout=cout,
                                 expand=(e != 1), skip=(j > 0)))
            if bs == 2:
                stride2_at.append(len(blocks) - 1)
    # taps: the tensor *entering* each stride-2 block, plus the last output; the wrapper
    # drops the first and HMDEgoPose.forward drops one more (efficientdet/model.py:452-458,
    # backbone.py:107) -> the last three.
    tapped = [b - 1 for b in stride2_at] + [len(blocks) - 1]
    taps = tuple(tapped[-3:])
    return Arch(phi=phi, stem=_round_width(32, wmul), blocks=tuple(blocks), taps=taps,
                tap_channels=tuple(blocks[t].cout for t in taps),
                fpn_w=_FPN_WIDTH[phi], fpn_cells=_FPN_REPEATS[phi],
                head_depth=_HEAD_DEPTH[phi], attention=phi < 6)


def level_sizes(size: int) -> List[int]:
    """Feature-map side of P3..P7 for a square input (anchors.py:257-270)."""
    return [(size + 2 ** p - 1) // 2 ** p for p in range(3, 8)]


def num_anchors_total(size: int) -> int:
    return NUM_ANCHORS * sum(s * s for s in level_sizes(size))


def same_pad(n: int, k: int, s: int) -> Tuple[int, int]:
    """TF 'SAME' padding (before, after) exactly as efficientnet/utils_extra.py:33-44."""
    extra = (math.ceil(n / s) - 1) * s - n + k
    before = extra // 2
    return before, extra - before


# --------------------------------------------------------------------------------------
# parameter inventory in the reference's state_dict order
# --------------------------------------------------------------------------------------
_BN = ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")


def _bn(prefix: str, c: int) -> Iterator[Tuple[str, tuple]]:
    for n in _BN:
        yield f"{prefix}.{n}", (() if n == "num_batches_tracked" else (c,))


def _sepconv(prefix: str, cin: int, cout: int, norm: bool) -> Iterator[Tuple[str, tuple]]:
    yield f"{prefix}.depthwise_conv.conv.weight", (cin, 1, 3, 3)
    yield f"{prefix}.pointwise_conv.conv.weight", (cout, cin, 1, 1)
    yield f"{prefix}.pointwise_conv.conv.bias", (cout,)
    if norm:
        yield from _bn(f"{prefix}.bn", cout)


def _lateral(prefix: str, cin: int, cout: int) -> Iterator[Tuple[str, tuple]]:
    yield f"{prefix}.0.conv.weight", (cout, cin, 1, 1)
    yield f"{prefix}.0.conv.bias", (cout,)
    yield from _bn(f"{prefix}.1", cout)


def _head(name: str, w: int, depth: int, headers: List[Tuple[str, int]]) -> Iterator[Tuple[str, tuple]]:
    for i in range(depth):
        yield from _sepconv(f"{name}.conv_list.{i}", w, w, norm=False)
    for lvl in range(NUM_LEVELS):
        for i in range(depth):
            yield from _bn(f"{name}.bn_list.{lvl}.{i}", w)
    for hname, n in headers:
        yield from _sepconv(f"{name}.{hname}", w, n, norm=False)


HEADERS = {
    "regressor": [("header", NUM_ANCHORS * 4)],
    "classifier": [("header", NUM_ANCHORS * 1)],
    "rotation_net": [("initial_rotation", NUM_ANCHORS * 3)],
    "translation_net": [("initial_translation_xy", NUM_ANCHORS * 2), ("initial_translation_z", NUM_ANCHORS)],
    "hand_net": [("initial_hand_coords", NUM_ANCHORS * 63)],
}

BIFPN_NODES = ("conv6_up", "conv5_up", "conv4_up", "conv3_up",
               "conv4_down", "conv5_down", "conv6_down", "conv7_down")
BIFPN_FUSION = (("p6_w1", 2), ("p5_w1", 2), ("p4_w1", 2), ("p3_w1", 2),
                ("p4_w2", 3), ("p5_w2", 3), ("p6_w2", 3), ("p7_w2", 2))


def param_spec(phi: int, num_classes: int = 1) -> List[Tuple[str, tuple]]:
    """Ordered (key, shape) list of the reference ``HMDEgoPose(...).state_dict()``
    with ``params['iter'] == 0`` (backbone.py:47-97 registration order)."""
    if not 1 <= int(num_classes) <= MAX_CLASSES:
        raise ValueError(f"num_classes must be in 1..{MAX_CLASSES}")
    a = get_arch(phi)
    w = a.fpn_w
    out: List[Tuple[str, tuple]] = []
    for r in range(a.fpn_cells):
        p = f"bifpn.{r}"
        for n, k in BIFPN_FUSION:
            out.append((f"{p}.{n}", (k,)))
        for n in BIFPN_NODES:
            out.extend(_sepconv(f"{p}.{n}", w, w, norm=True))
        if r == 0:
            c3, c4, c5 = a.tap_channels
            for n, c in (("p5_down_channel", c5), ("p4_down_channel", c4), ("p3_down_channel", c3),
                         ("p5_to_p6", c5), ("p4_down_channel_2", c4), ("p5_down_channel_2", c5)):
                out.extend(_lateral(f"{p}.{n}", c, w))
    out.extend(_head("regressor", w, a.head_depth, HEADERS["regressor"]))
    out.extend(_head("classifier", w, a.head_depth, [("header", NUM_ANCHORS * int(num_classes))]))   # efficientdet/model.py:393
    bb = "backbone_net.model"
    out.append((f"{bb}._conv_stem.conv.weight", (a.stem, 3, 3, 3)))
    out.extend(_bn(f"{bb}._bn0", a.stem))
    for i, b in enumerate(a.blocks):
        p = f"{bb}._blocks.{i}"
        if b.expand:
            out.append((f"{p}._expand_conv.conv.weight", (b.cexp, b.cin, 1, 1)))
            out.extend(_bn(f"{p}._bn0", b.cexp))
        out.append((f"{p}._depthwise_conv.conv.weight", (b.cexp, 1, b.k, b.k)))
        out.extend(_bn(f"{p}._bn1", b.cexp))
        out.append((f"{p}._se_reduce.conv.weight", (b.se, b.cexp, 1, 1)))
        out.append((f"{p}._se_reduce.conv.bias", (b.se,)))
        out.append((f"{p}._se_expand.conv.weight", (b.cexp, b.se, 1, 1)))
        out.append((f"{p}._se_expand.conv.bias", (b.cexp,)))
        out.append((f"{p}._project_conv.conv.weight", (b.cout, b.cexp, 1, 1)))
        out.extend(_bn(f"{p}._bn2", b.cout))
    for n in ("rotation_net", "translation_net", "hand_net"):
        out.extend(_head(n, w, a.head_depth, HEADERS[n]))
    return out
