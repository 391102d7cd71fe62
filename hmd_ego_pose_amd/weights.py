"""Weights for the MI355X EfficientPose path: checkpoint key handling, a
version-stable seeded initialiser, and the ``HEPW`` container the C-ABI reads.

The container is deliberately dumb: the reference's own ``state_dict`` tensors
by their own names, fp32, nothing folded.  BatchNorm folding, fusion-weight
normalisation, NHWC/GEMM layouts and bf16 conversion happen inside
``libhep.so`` at ``hep_create`` time (csrc/hep_model.cpp), so a checkpoint
converted once serves every dtype and the C# host never needs Python.

Reference behaviour restated:
  * checkpoint prefixes ``model.`` / ``model.module.`` .. pytorch-sandbox/evaluate.py:102-119,
    pytorch-sandbox/hmdegopose/misc_utils.py:45-47
"""
from __future__ import annotations

import struct
import zlib
from collections import OrderedDict
from typing import Dict, Mapping

import numpy as np
import torch

from .arch import param_spec

MAGIC = b"HEPW"
VERSION = 1
_ALIGN = 64
# amplitude multipliers on top of N(0, 1/fan_in), per conv role; fixed constants tuned once so
# that seeded activations stay O(1) AND input-dependent through every stage (see tests/golden)
GAINS = {"stem": 1.0, "expand": 1.5, "dw": None, "project": 1.0, "se": 1.0, "sep_dw": 1.35, "sep_pw": 1.2, "lateral": 1.0}
_DW_GAIN_OF_PHI = [1.85, 1.8, 1.75, 1.7, 1.65, 1.45, 1.35, 1.35]   # deeper backbones need less (0..6 tuned; golden vectors exist for 0 and 3)
_SEP_GAIN_OF_PHI = [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.8, 0.8]       # 8 BiFPN cells + 5-layer heads of phi 6 need less in the separable convs


def strip_checkpoint_prefix(state: Mapping[str, torch.Tensor]) -> "OrderedDict[str, torch.Tensor]":
    """Accept un-prefixed keys, ``model.`` (TrainModelWithLoss wrapper) and
    ``model.module.`` (wrapper around DataParallel) exactly as evaluate.py does by
    slicing ``k[6:]`` / ``k[13:]``."""
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, v in state.items():
        if k.startswith("model.module."):
            k = k[len("model.module."):]
        elif k.startswith("model."):
            k = k[len("model."):]
        out[k] = v
    return out


def _kind(key: str) -> str:
    if key.endswith("num_batches_tracked"):
        return "count"
    if key.endswith("running_var"):
        return "var"
    if key.endswith("running_mean"):
        return "mean"
    leaf = key.rsplit(".", 1)[-1]
    parent = key.rsplit(".", 2)[-2] if key.count(".") >= 1 else ""
    if leaf in ("p6_w1", "p5_w1", "p4_w1", "p3_w1", "p4_w2", "p5_w2", "p6_w2", "p7_w2"):
        return "fusion"
    if parent == "conv":
        return "conv_w" if leaf == "weight" else "conv_b"
    return "bn_w" if leaf == "weight" else "bn_b"


def _conv_role(key: str) -> str:
    for pat, role in (("_conv_stem", "stem"), ("_expand_conv", "expand"), ("_depthwise_conv", "dw"),
                      ("_project_conv", "project"), ("_se_", "se"), ("depthwise_conv", "sep_dw"),
                      ("pointwise_conv", "sep_pw")):
        if pat in key:
            return role
    return "lateral"


def seeded_state_dict(phi: int, seed: int = 0, gain: float = 1.0, num_classes: int = 1) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic synthetic weights (no checkpoint ships with the reference).

    Every tensor gets its own ``numpy`` PCG64 stream keyed by (seed, crc32(key)), so
    the values do not depend on iteration order or on torch's RNG: the golden-vector
    script (run against the reference) and the GPU box regenerate identical weights.
    Distributions keep activations O(1) through ~100 layers and exercise every quirk:
    conv ~ N(0, gain/fan_in), biases/BN-beta/mean ~ N(0, 0.1), BN-gamma/var ~ U(0.5, 1.5),
    BiFPN fusion weights ~ U(-0.5, 2) (negative ones hit the ReLU).
    """
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for key, shape in param_spec(phi, num_classes):   # (num_classes only changes the classifier header's two tensors)
        rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(key.encode())]))
        kind = _kind(key)
        if kind == "count":
            out[key] = torch.zeros((), dtype=torch.int64)
            continue
        if kind == "conv_w":
            fan_in = shape[1] * shape[2] * shape[3]
            a = rng.standard_normal(shape) * np.sqrt(gain / fan_in)
            role = _conv_role(key)
            a = a * (GAINS[role] if GAINS[role] is not None else _DW_GAIN_OF_PHI[phi])
            if role in ("sep_dw", "sep_pw"):
                a = a * _SEP_GAIN_OF_PHI[phi]
        elif kind in ("conv_b", "bn_b", "mean"):
            a = rng.standard_normal(shape) * 0.1
        elif kind in ("bn_w", "var"):
            a = rng.uniform(0.5, 1.5, shape)
        elif kind == "fusion":
            a = rng.uniform(-0.5, 2.0, shape)
        else:  # pragma: no cover
            raise AssertionError(kind)
        out[key] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return out


def pack_bytes(state: Mapping[str, torch.Tensor]) -> bytes:
    """Serialise a state_dict to the HEPW container (little-endian):

        "HEPW" u32 version u32 count
        count x { u16 name_len, name, u8 ndim, ndim x u32 dims, u64 offset, u64 nbytes }
        payload (each tensor fp32, 64-byte aligned, offsets relative to file start)

    ``num_batches_tracked`` and any non-float tensor are skipped.
    """
    state = strip_checkpoint_prefix(state)
    entries = []
    for k, v in state.items():
        if not torch.is_floating_point(v):
            continue
        a = v.detach().to("cpu", torch.float32).contiguous().numpy()
        entries.append((k.encode(), a))
    head = 12
    for name, a in entries:
        head += 2 + len(name) + 1 + 4 * a.ndim + 16
    off = (head + _ALIGN - 1) // _ALIGN * _ALIGN
    table = [MAGIC, struct.pack("<II", VERSION, len(entries))]
    blobs = []
    for name, a in entries:
        nb = a.nbytes
        table.append(struct.pack("<H", len(name)) + name + struct.pack("<B", a.ndim)
                     + struct.pack(f"<{a.ndim}I", *a.shape) + struct.pack("<QQ", off, nb))
        pad = (-nb) % _ALIGN
        blobs.append(a.tobytes() + b"\0" * pad)
        off += nb + pad
    headb = b"".join(table)
    headb += b"\0" * ((-len(headb)) % _ALIGN)
    return headb + b"".join(blobs)


def save_pack(state: Mapping[str, torch.Tensor], path: str) -> None:
    with open(path, "wb") as f:
        f.write(pack_bytes(state))


def load_pack(path_or_bytes) -> Dict[str, np.ndarray]:
    """Read a HEPW container back (used by tests and tools)."""
    raw = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
    if raw[:4] != MAGIC:
        raise ValueError("not a HEPW weight pack")
    ver, count = struct.unpack_from("<II", raw, 4)
    if ver != VERSION:
        raise ValueError(f"HEPW version {ver} not supported")
    p = 12
    out: Dict[str, np.ndarray] = {}
    for _ in range(count):
        (nl,) = struct.unpack_from("<H", raw, p); p += 2
        name = raw[p:p + nl].decode(); p += nl
        nd = raw[p]; p += 1
        dims = struct.unpack_from(f"<{nd}I", raw, p); p += 4 * nd
        off, nb = struct.unpack_from("<QQ", raw, p); p += 16
        out[name] = np.frombuffer(raw, dtype="<f4", count=nb // 4, offset=off).reshape(dims)
    return out
