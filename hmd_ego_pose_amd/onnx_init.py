"""Minimal ONNX initialiser reader: pulls the named weight tensors out of an ``.onnx`` file
without onnx / onnxruntime / protobuf, so that an exported model (reference ``pytorch-sandbox/main.py`` ``--export-onnx``
path, ``hmdegopose/misc_utils.py:36-56,77-83``) can feed ``tools/pack_weights.py``.

Only the protobuf wire format and four message layouts of the published ``onnx.proto`` (onnx 1.x, IR version >= 3) are
needed - field numbers below are from that file:

    ModelProto   7: graph (GraphProto)
    GraphProto   5: initializer (repeated TensorProto)
    TensorProto  1: dims (repeated int64, packed or not)   2: data_type (int32)   4: float_data (packed float)
                 7: int64_data (packed)   8: name   9: raw_data (little-endian bytes)   13: data_location (0 = inline)
                 3: name   4: op_type
    GraphProto   1: node (repeated NodeProto)
    NodeProto    1: input (repeated string)   2: output   3: name   4: op_type

Pinned against a REAL exporter (round 4): ``tests/golden/make_golden_onnx.py`` runs ``torch.onnx.export`` on the imported
reference model in the build container exactly as ``export_to_onnx`` does (eval mode, opset 9; the exporter's C++ serialiser
needs no ``onnx`` package - only a post-pass that splices onnx-script functions in does, and there are none) and commits the
file's structure (every byte of it except the tensor payloads) as a fixture; ``tests/test_host_cpu.py`` rebuilds the payloads
from the seeded weights and reads the file back.  The authors' own ``model.onnx`` (``.MISSING_LARGE_BLOBS``) is still absent.

An eval-mode export - what ``export_to_onnx`` writes - folds every BatchNorm into the convolution in front of it: the folded
weights are anonymous initialisers (``onnx::Conv_3046``; older exporters: bare numbers) and 153 of the 1048 ``state_dict``
entries no longer exist by name.  ``state_dict_from_onnx`` maps them back through the ORDER of the graph's Conv nodes - the
trace order of ``HMDEgoPose.forward``, ``conv_exec_order`` below - checked at every convolution that kept its name (squeeze-
excite convs, every depthwise conv of a SeparableConvBlock, the header convs: 191 of the 344 Conv nodes at phi 0) and by the
shape of every folded one; the head towers' shared pointwise convs, folded once per pyramid level, are split back into one
weight and five per-level scales (refused unless the five folded copies are exact per-channel multiples of each other).
The result is an equivalent ``state_dict`` (conv weight = folded weight, BatchNorm = identity scale + folded bias), not the
original tensors.  A training-mode export keeps every key as an initialiser name and is read directly.
"""
from __future__ import annotations

import struct
from collections import OrderedDict
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np

_DTYPES = {1: np.float32, 6: np.int32, 7: np.int64, 10: np.float16, 11: np.float64}


def _varint(b: bytes, i: int) -> Tuple[int, int]:
    v = 0; s = 0
    while True:
        if i >= len(b):
            raise ValueError("truncated varint")
        c = b[i]; i += 1
        v |= (c & 0x7F) << s
        if not c & 0x80:
            return v, i
        s += 7
        if s > 63:
            raise ValueError("varint too long")


def _fields(b: bytes) -> Iterator[Tuple[int, int, object]]:
    """(field number, wire type, value) of one message: varint -> int, 64-bit / 32-bit -> bytes, length-delimited -> bytes."""
    i = 0
    while i < len(b):
        key, i = _varint(b, i)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(b, i)
        elif wt == 1:
            v = b[i:i + 8]; i += 8
        elif wt == 5:
            v = b[i:i + 4]; i += 4
        elif wt == 2:
            n, i = _varint(b, i)
            if i + n > len(b):
                raise ValueError("truncated length-delimited field")
            v = b[i:i + n]; i += n
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield f, wt, v


def _packed_varints(b: bytes):
    i = 0
    while i < len(b):
        v, i = _varint(b, i)
        yield v - (1 << 64) if v >= 1 << 63 else v


def _tensor(b: bytes) -> Tuple[str, np.ndarray]:
    dims, dtype, name, raw, fdata, idata, external = [], 1, "", None, None, None, False
    for f, wt, v in _fields(b):
        if f == 1:
            dims.extend(_packed_varints(v) if wt == 2 else [v])
        elif f == 2:
            dtype = v
        elif f == 4:
            fdata = np.frombuffer(v, "<f4") if wt == 2 else np.concatenate([fdata if fdata is not None else np.zeros(0, "<f4"), np.frombuffer(v, "<f4")])
        elif f == 7 and wt == 2:
            idata = np.array(list(_packed_varints(v)), np.int64)
        elif f == 8:
            name = v.decode("utf-8")
        elif f == 9:
            raw = v
        elif f == 13 and v != 0:
            external = True
    if external:
        raise ValueError(f"initializer {name!r} keeps its data in an external file: not supported")
    if dtype not in _DTYPES:
        raise ValueError(f"initializer {name!r}: unsupported ONNX data type {dtype}")
    if raw is not None:
        a = np.frombuffer(raw, np.dtype(_DTYPES[dtype]).newbyteorder("<"))
    elif fdata is not None and dtype == 1:
        a = fdata
    elif idata is not None and dtype in (6, 7):
        a = idata.astype(_DTYPES[dtype])
    else:
        a = np.zeros(0, _DTYPES[dtype])
    n = int(np.prod(dims)) if dims else a.size
    if a.size != n:
        raise ValueError(f"initializer {name!r}: {a.size} elements for dims {dims}")
    return name, a.astype(_DTYPES[dtype]).reshape(dims)


def read_initializers(path_or_bytes) -> "OrderedDict[str, np.ndarray]":
    """name -> array for every inline initialiser of the model's graph, in file order."""
    data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    graphs = [v for f, wt, v in _fields(bytes(data)) if f == 7 and wt == 2]
    if len(graphs) != 1:
        raise ValueError("not an ONNX ModelProto (expected exactly one graph)")
    for f, wt, v in _fields(graphs[0]):
        if f == 5 and wt == 2:
            name, a = _tensor(v)
            out[name] = a
    return out


def _fields_at(b: bytes, lo: int, hi: int) -> Iterator[Tuple[int, int, int, int]]:
    """(field number, wire type, value start, value end) of the message b[lo:hi], offsets absolute in b (no copies)."""
    i = lo
    while i < hi:
        key, i = _varint(b, i)
        f, wt = key >> 3, key & 7
        if wt == 0:
            s0 = i; _v, i = _varint(b, i); yield f, wt, s0, i
        elif wt == 1:
            yield f, wt, i, i + 8; i += 8
        elif wt == 5:
            yield f, wt, i, i + 4; i += 4
        elif wt == 2:
            n, i = _varint(b, i)
            if i + n > hi:
                raise ValueError("truncated length-delimited field")
            yield f, wt, i, i + n; i += n
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")


def initializer_spans(data: bytes) -> List[Tuple[str, int, int]]:
    """(name, payload start, payload end) of every initialiser that keeps its values as ``raw_data``, offsets into ``data`` -
    what a tool needs to blank or refill the tensor payloads of a file in place (tests/golden/make_golden_onnx.py)."""
    data = bytes(data)
    out = []
    for f, wt, s0, e0 in _fields_at(data, 0, len(data)):
        if f == 7 and wt == 2:
            for f1, wt1, s1, e1 in _fields_at(data, s0, e0):
                if f1 == 5 and wt1 == 2:
                    name, span = "", None
                    for f2, wt2, s2, e2 in _fields_at(data, s1, e1):
                        if f2 == 8:
                            name = data[s2:e2].decode("utf-8")
                        elif f2 == 9:
                            span = (s2, e2)
                    if span is not None:
                        out.append((name, span[0], span[1]))
    return out


def _graph(data: bytes) -> bytes:
    graphs = [v for f, wt, v in _fields(bytes(data)) if f == 7 and wt == 2]
    if len(graphs) != 1:
        raise ValueError("not an ONNX ModelProto (expected exactly one graph)")
    return graphs[0]


def read_nodes(path_or_bytes) -> List[Tuple[str, List[str], List[str], str]]:
    """(op_type, inputs, outputs, name) of every node of the model's graph, in graph (= trace) order."""
    data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
    out = []
    for f, wt, v in _fields(_graph(data)):
        if f == 1 and wt == 2:
            ins, outs, name, op = [], [], "", ""
            for f2, wt2, v2 in _fields(v):
                if f2 == 1:
                    ins.append(v2.decode("utf-8"))
                elif f2 == 2:
                    outs.append(v2.decode("utf-8"))
                elif f2 == 3:
                    name = v2.decode("utf-8")
                elif f2 == 4:
                    op = v2.decode("utf-8")
            out.append((op, ins, outs, name))
    return out


def conv_exec_order(phi: int) -> List[Tuple[str, Optional[str], Optional[str]]]:
    """(weight key, bias key | None, BatchNorm prefix | None) of every convolution CALL of ``HMDEgoPose.forward`` in execution
    order - the order of the Conv nodes of a traced export.  Reference: backbone ``efficientdet/model.py:437-456`` +
    ``efficientnet/model.py:69-104`` (expand, depthwise, squeeze-excite reduce / expand, project); BiFPN
    ``efficientdet/model.py:194-264`` (cell 0: p5_to_p6 and the three lateral convs, the four top-down nodes, the two ``_2``
    laterals, the four bottom-up nodes; a node = depthwise then pointwise + bn); heads ``efficientdet/model.py:361-417``,
    ``hmdegopose/model.py:55-228`` (per net, per level: the tower layers then the header convs; shared convs are called once per
    level)."""
    from .arch import BIFPN_NODES, HEAD_NAMES, HEADERS, NUM_LEVELS, get_arch
    a = get_arch(phi)
    bb = "backbone_net.model"
    seq: List[Tuple[str, Optional[str], Optional[str]]] = [(f"{bb}._conv_stem.conv.weight", None, f"{bb}._bn0")]
    for i, b in enumerate(a.blocks):
        p = f"{bb}._blocks.{i}"
        if b.expand:
            seq.append((f"{p}._expand_conv.conv.weight", None, f"{p}._bn0"))
        seq.append((f"{p}._depthwise_conv.conv.weight", None, f"{p}._bn1"))
        seq.append((f"{p}._se_reduce.conv.weight", f"{p}._se_reduce.conv.bias", None))
        seq.append((f"{p}._se_expand.conv.weight", f"{p}._se_expand.conv.bias", None))
        seq.append((f"{p}._project_conv.conv.weight", None, f"{p}._bn2"))
    lat = lambda p: (f"{p}.0.conv.weight", f"{p}.0.conv.bias", f"{p}.1")
    def sep(p, bn):
        return [(f"{p}.depthwise_conv.conv.weight", None, None), (f"{p}.pointwise_conv.conv.weight", f"{p}.pointwise_conv.conv.bias", bn)]
    for r in range(a.fpn_cells):
        p = f"bifpn.{r}"
        if r == 0:
            seq += [lat(f"{p}.p5_to_p6"), lat(f"{p}.p3_down_channel"), lat(f"{p}.p4_down_channel"), lat(f"{p}.p5_down_channel")]
        for n in BIFPN_NODES[:4]:
            seq += sep(f"{p}.{n}", f"{p}.{n}.bn")
        if r == 0:
            seq += [lat(f"{p}.p4_down_channel_2"), lat(f"{p}.p5_down_channel_2")]
        for n in BIFPN_NODES[4:]:
            seq += sep(f"{p}.{n}", f"{p}.{n}.bn")
    for net in HEAD_NAMES:
        for lvl in range(NUM_LEVELS):
            for i in range(a.head_depth):
                seq += sep(f"{net}.conv_list.{i}", f"{net}.bn_list.{lvl}.{i}")
            for hname, _n in HEADERS[net]:
                seq += sep(f"{net}.{hname}", None)
    return seq


def state_dict_from_onnx(path_or_bytes, phi: int) -> Dict[str, "np.ndarray"]:
    """A ``state_dict`` of an EfficientPose of this phi from an exported model.

    * training-mode export (every key an initialiser name, with or without the ``model.`` / ``model.module.`` wrappers): the
      tensors themselves, shapes checked;
    * eval-mode export with BatchNorm folded into the convolutions (what the reference's ``export_to_onnx`` writes): an
      EQUIVALENT state_dict - see the module docstring.  Raises ``ValueError`` with the first inconsistency when the graph is
      not the trace of this architecture."""
    import torch
    from .arch import BN_EPS
    from .weights import param_spec, strip_checkpoint_prefix
    data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
    init = read_initializers(data)
    named = strip_checkpoint_prefix(OrderedDict((k, torch.from_numpy(np.array(v))) for k, v in init.items()))
    # num_classes is what the classifier header holds (9 * num_classes channels; it has no BatchNorm behind it, so it keeps its name in both kinds of export)
    hdr = named.get("classifier.header.pointwise_conv.conv.weight")
    classes = int(hdr.shape[0]) // 9 if hdr is not None and hdr.dim() == 4 and hdr.shape[0] % 9 == 0 and 9 <= hdr.shape[0] <= 9 * 63 else 1
    want = OrderedDict(param_spec(phi, classes))
    missing = [k for k in want if k not in named and not k.endswith("num_batches_tracked")]
    wrong = [k for k in want if k in named and tuple(named[k].shape) != tuple(want[k])]
    if wrong:
        raise ValueError(f"{len(wrong)} initialisers have another shape than the phi {phi} architecture (e.g. {wrong[:2]})")
    if not missing:
        return {k: named[k] for k in want if k in named}
    # ---- BatchNorm-folded export: walk the Conv nodes in trace order ----
    strip = lambda n: n[13:] if n.startswith("model.module.") else (n[6:] if n.startswith("model.") else n)
    convs = [n for n in read_nodes(data) if n[0] == "Conv"]
    seq = conv_exec_order(phi)
    if len(convs) != len(seq):
        raise ValueError(f"the graph has {len(convs)} Conv nodes, the trace of a phi {phi} EfficientPose has {len(seq)}: {len(missing)} state_dict "
                         f"entries are not initialisers of this file (e.g. {missing[:2]}) and it cannot be mapped back by node order either")
    state: Dict[str, np.ndarray] = {}
    folded: Dict[str, List[Tuple[str, np.ndarray, np.ndarray]]] = OrderedDict()      # weight key -> [(bn prefix, folded W, folded B)]
    for idx, ((_op, ins, _outs, nname), (wk, bk, bn)) in enumerate(zip(convs, seq)):
        if len(ins) < 2 or ins[1] not in init:
            raise ValueError(f"Conv node {idx} ({nname or '?'}): its weight is not an initialiser")
        W = np.asarray(init[ins[1]], np.float32)
        B = np.asarray(init[ins[2]], np.float32) if len(ins) > 2 and ins[2] in init else None
        if tuple(W.shape) != tuple(want[wk]):
            raise ValueError(f"Conv node {idx} ({nname or ins[1]}): weight shape {tuple(W.shape)}, the trace of a phi {phi} EfficientPose expects "
                             f"{wk} {tuple(want[wk])} here - not an export of this architecture (or another trace order)")
        wname = strip(ins[1])
        if wname in want and wname != wk:
            raise ValueError(f"Conv node {idx}: weight initialiser {wname!r} where the trace order expects {wk!r}")
        if wname == wk:                               # kept its name: not folded (no BatchNorm behind it, or an unfolded export)
            if wk in state and not np.array_equal(state[wk], W):
                raise ValueError(f"{wk}: two different tensors under one name")
            state[wk] = W
            if bk is not None:
                if B is None:
                    raise ValueError(f"Conv node {idx}: {bk} is missing")
                state[bk] = B
            continue
        if bn is None:
            raise ValueError(f"Conv node {idx}: anonymous weight {ins[1]!r} for {wk}, which has no BatchNorm to fold")
        if B is None or B.shape != (W.shape[0],):
            raise ValueError(f"Conv node {idx}: a BatchNorm-folded convolution carries a bias of its output width")
        folded.setdefault(wk, []).append((bn, W, B))
        if bk is not None:
            state[bk] = np.zeros((W.shape[0],), np.float32)          # the conv bias went into the folded bias
    ident = np.float32(np.sqrt(np.float32(1.0) + np.float32(BN_EPS)))          # gamma / sqrt(1 + eps) == 1 exactly in float32
    def put_bn(prefix, scale, shift):
        c = shift.shape[0]
        state[prefix + ".weight"] = (scale * ident).astype(np.float32) if scale is not None else np.full((c,), ident, np.float32)
        state[prefix + ".bias"] = shift.astype(np.float32)
        state[prefix + ".running_mean"] = np.zeros((c,), np.float32)
        state[prefix + ".running_var"] = np.ones((c,), np.float32)
        state[prefix + ".num_batches_tracked"] = np.zeros((), np.int64)
    for wk, items in folded.items():
        W0 = items[0][1]
        state[wk] = W0
        put_bn(items[0][0], None, items[0][2])
        f0 = W0.reshape(W0.shape[0], -1).astype(np.float64)
        n0 = (f0 * f0).sum(1)
        for bn, W, B in items[1:]:                    # a shared convolution folded once per BatchNorm (head towers, one per level)
            fl = W.reshape(W.shape[0], -1).astype(np.float64)
            if np.any(n0 == 0):
                raise ValueError(f"{wk}: an all-zero output channel - the per-level scales cannot be separated from the shared weight")
            ratio = (fl * f0).sum(1) / n0
            resid = np.abs(fl - ratio[:, None] * f0).max()
            if resid > 1e-5 * max(1.0, float(np.abs(fl).max())):
                raise ValueError(f"{wk}: the folded copies are not per-channel multiples of one weight (residual {resid:.2e}): not a shared convolution")
            put_bn(bn, ratio.astype(np.float32), B)
    out = {}
    for k, shape in want.items():
        if k in state:
            v = state[k]
        elif k in named:
            v = named[k].numpy()
        elif k.endswith("num_batches_tracked"):
            v = np.zeros((), np.int64)
        else:
            raise ValueError(f"{k} could not be recovered from the file")
        if tuple(np.shape(v)) != tuple(shape):
            raise ValueError(f"{k}: shape {tuple(np.shape(v))} != {tuple(shape)}")
        out[k] = torch.from_numpy(np.ascontiguousarray(v))
    return out
