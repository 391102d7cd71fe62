"""Minimal ONNX initialiser reader: pulls the named weight tensors out of an ``.onnx`` file
without onnx / onnxruntime / protobuf, so that an exported model (reference ``pytorch-sandbox/main.py`` ``--export-onnx``
path, ``hmdegopose/misc_utils.py:36-56,77-83``) can feed ``tools/pack_weights.py``.

Only the protobuf wire format and four message layouts of the published ``onnx.proto`` (onnx 1.x, IR version >= 3) are
needed - field numbers below are from that file:

    ModelProto   7: graph (GraphProto)
    GraphProto   5: initializer (repeated TensorProto)
    TensorProto  1: dims (repeated int64, packed or not)   2: data_type (int32)   4: float_data (packed float)
                 7: int64_data (packed)   8: name   9: raw_data (little-endian bytes)   13: data_location (0 = inline)

PARITY UNPINNED: no ``.onnx`` file exists in this build (the authors' ``efficientpose-0.onnx`` is listed in
``.MISSING_LARGE_BLOBS``; ``torch.onnx.export`` needs the absent ``onnx`` package), so the reader is tested against files
written by the encoder in ``tests/test_host_cpu.py`` from the same published layout - not against a real exporter.
An eval-mode export folds BatchNorm into anonymous conv initialisers (``onnx::Conv_123``): those files carry no
``state_dict`` names and ``state_dict_from_onnx`` refuses them; a training-mode export (``training=TrainingMode.TRAINING``,
``do_constant_folding=False``) keeps every ``state_dict`` key as an initialiser name and is what this reader is for.
"""
from __future__ import annotations

import struct
from collections import OrderedDict
from typing import Dict, Iterator, Tuple

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


def state_dict_from_onnx(path_or_bytes, phi: int) -> Dict[str, "np.ndarray"]:
    """The initialisers that are ``state_dict`` entries of an EfficientPose of this phi (names with or without the
    ``model.`` / ``model.module.`` wrappers, shapes checked).  Raises when the file does not carry them - an eval-mode
    export with BatchNorm folded into anonymous initialisers cannot be mapped back."""
    import torch
    from .weights import param_spec, strip_checkpoint_prefix
    init = read_initializers(path_or_bytes)
    state = strip_checkpoint_prefix(OrderedDict((k, torch.from_numpy(np.array(v))) for k, v in init.items()))
    want = dict(param_spec(phi))
    missing = [k for k in want if k not in state and not k.endswith("num_batches_tracked")]
    wrong = [k for k in want if k in state and tuple(state[k].shape) != tuple(want[k])]
    if missing or wrong:
        raise ValueError(f"{len(missing)} of {len(want)} state_dict entries are not initialisers of this file (e.g. {missing[:2]}), "
                         f"{len(wrong)} have another shape (e.g. {wrong[:2]}): an eval-mode export folds BatchNorm into anonymous "
                         f"initialisers and cannot be mapped back - export in training mode without constant folding")
    return {k: state[k] for k in want if k in state}
