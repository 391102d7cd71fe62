"""CPU (no GPU): host logic of the product - architecture tables, weight pack, the C-ABI
library's exports and its host-only entry points, the nn.Module drop-in's state_dict
contract and its loud refusal to run without a ROCm device."""
import ctypes
import hashlib
import math
import os
import re

import numpy as np
import pytest
import torch

from hmd_ego_pose_amd import _capi
from hmd_ego_pose_amd.arch import get_arch, level_sizes, num_anchors_total, param_spec, same_pad
from hmd_ego_pose_amd.weights import load_pack, pack_bytes, seeded_state_dict, strip_checkpoint_prefix
from tests._util import golden_meta

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("phi", [0, 3])
def test_param_spec_matches_reference_state_dict(phi):
    spec = param_spec(phi)
    assert hashlib.sha256(repr(spec).encode()).hexdigest() == golden_meta()[f"keys_phi{phi}_sha256"]
    assert len(spec) == {0: 1048, 3: 1618}[phi]


def test_arch_tables_phi0_phi3():
    a = get_arch(0)
    assert (a.stem, len(a.blocks), a.taps, a.tap_channels) == (32, 16, (4, 10, 15), (40, 112, 320))
    assert [(b.cexp, b.k, b.stride) for b in a.blocks[:4]] == [(32, 3, 1), (96, 3, 2), (144, 3, 1), (144, 5, 2)]
    assert [b.skip for b in a.blocks] == [False, False, True, False, True, False, True, True, False, True, True, False, True, True, True, False]
    b = get_arch(3)
    assert (b.stem, len(b.blocks), b.taps, b.tap_channels, b.fpn_w, b.fpn_cells, b.head_depth) == (40, 26, (7, 17, 25), (48, 136, 384), 160, 6, 4)
    assert level_sizes(256) == [32, 16, 8, 4, 2] and num_anchors_total(256) == 12276 and num_anchors_total(512) == 49104
    # TF-SAME asymmetry
    assert same_pad(256, 3, 2) == (0, 1) and same_pad(64, 5, 2) == (1, 2) and same_pad(32, 3, 1) == (1, 1) and same_pad(16, 5, 1) == (2, 2)
    with pytest.raises(ValueError):
        get_arch(8)


def test_weight_pack_roundtrip_and_prefixes():
    sd = seeded_state_dict(0, 3)
    blob = pack_bytes({("model.module." + k): v for k, v in sd.items()})
    back = load_pack(blob)
    assert "bifpn.0.p6_w1" in back and not any(k.endswith("num_batches_tracked") for k in back)
    for k in ("backbone_net.model._conv_stem.conv.weight", "hand_net.initial_hand_coords.pointwise_conv.conv.bias"):
        assert np.array_equal(back[k], sd[k].numpy())
    assert list(strip_checkpoint_prefix({"model.bifpn.0.p6_w1": 1, "backbone_net.model._bn0.weight": 2})) == ["bifpn.0.p6_w1", "backbone_net.model._bn0.weight"]
    # version-stable: same seed -> same bytes, different seed -> different
    assert torch.equal(seeded_state_dict(0, 3)["regressor.header.pointwise_conv.conv.weight"], sd["regressor.header.pointwise_conv.conv.weight"])
    assert not torch.equal(seeded_state_dict(0, 4)["bifpn.1.p4_w2"], sd["bifpn.1.p4_w2"])


def test_library_exports_every_symbol_in_header():
    """The C-ABI library loads and exports exactly what include/hep.h declares."""
    header = open(os.path.join(REPO, "include", "hep.h")).read()
    declared = set(re.findall(r"\b(hep_[a-z_0-9]+)\s*\(", header)) - {"hep_handle"}
    assert declared == set(_capi.SYMBOLS), declared ^ set(_capi.SYMBOLS)
    l = _capi.lib()
    for name in declared:
        assert hasattr(l, name), name
    assert l.hep_abi_version() == 1


@pytest.mark.parametrize("size", [256, 512])
def test_hep_anchors_bit_exact_against_reference_fixtures(size):
    """hep_anchors is host code (float64 math, one cast): bit-identical to the reference's
    anchors_for_shape and to its onnx-models/*.txt fixtures."""
    l = _capi.lib()
    n = num_anchors_total(size)
    a = np.empty((n, 4), np.float32); t = np.empty((n, 3), np.float32)
    assert l.hep_anchors(size, a.ctypes.data, t.ctypes.data) == n
    meta = golden_meta()
    assert hashlib.sha256(a.tobytes()).hexdigest() == meta[f"anchors_{size}_sha256"]
    assert hashlib.sha256(t.tobytes()).hexdigest() == meta[f"translation_anchors_{size}_sha256"]
    assert l.hep_anchors(100, None, None) < 0 and b"multiple of 128" in l.hep_last_error()


def test_create_fails_loudly_without_gpu_or_with_bad_pack():
    l = _capi.lib()
    h = ctypes.c_void_p()
    rc = l.hep_create(b"/nonexistent/pack.hepw", 0, 256, 1, 0, 0, 0, ctypes.byref(h))
    assert rc == -2 and b"cannot open" in l.hep_last_error() and not h.value
    blob = pack_bytes(seeded_state_dict(0, 0))
    if l.hep_device_count() == 0:     # this container: no silent CPU path
        rc = l.hep_create_from_memory(blob, len(blob), 0, 256, 1, 0, 0, 0, ctypes.byref(h))
        assert rc == -3 and b"no CPU fallback" in l.hep_last_error()
    assert l.hep_create_from_memory(b"XXXX" + bytes(60), 64, 0, 256, 1, 0, 0, 0, ctypes.byref(h)) in (-2, -3)
    assert l.hep_create_from_memory(blob, len(blob), 9, 256, 1, 0, 0, 0, ctypes.byref(h)) == -4
    assert l.hep_create_from_memory(blob, len(blob), 0, 200, 1, 0, 0, 0, ctypes.byref(h)) == -4
    # the fp8 dtype is an opt-in build (make fp8 -> libhep_fp8.so): the default library says so before it looks for a device
    if not os.environ.get("HEP_LIB"):
        assert l.hep_create_from_memory(blob, len(blob), 0, 256, 1, _capi.HEP_FP8, 0, 0, ctypes.byref(h)) == -4
        assert b"FP8=1" in l.hep_last_error() and not h.value


def test_module_state_dict_contract_and_cpu_refusal():
    from hmd_ego_pose_amd import HMDEgoPose
    m = HMDEgoPose({"iter": 0}, num_classes=1, compound_coef=0, onnx_export=True, input_sizes=[256] * 9)
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == param_spec(0)
    sd = seeded_state_dict(0, 5)
    res = m.load_state_dict({"model." + k: v for k, v in sd.items()}, strict=False)   # prefixed keys are NOT auto-stripped by torch
    assert len(res.unexpected_keys) == len(sd)
    assert m.load_state_dict(strip_checkpoint_prefix({"model." + k: v for k, v in sd.items()}), strict=True).missing_keys == []
    assert torch.equal(m.state_dict()["rotation_net.initial_rotation.pointwise_conv.conv.bias"], sd["rotation_net.initial_rotation.pointwise_conv.conv.bias"])
    m.eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 256, 256))
    m.train()
    with pytest.raises(RuntimeError, match="inference path"):
        m(torch.zeros(1, 3, 256, 256))
    with pytest.raises(ValueError, match="iter"):
        HMDEgoPose({"iter": 1}, compound_coef=0)


def test_pack_weights_tool_roundtrip(tmp_path):
    """tools/pack_weights.py: a `model.module.`-prefixed .pth checkpoint (what evaluate.py:102-116 loads) -> HEPW pack
    holding the reference's tensors by their own names; a wrong phi is refused."""
    import subprocess
    import sys
    sd = seeded_state_dict(0, 5)
    ckpt = tmp_path / "ckpt.pth"
    torch.save({("model.module." + k): v for k, v in sd.items()}, ckpt)
    out = tmp_path / "m.hepw"
    tool = os.path.join(REPO, "tools", "pack_weights.py")
    r = subprocess.run([sys.executable, tool, str(ckpt), str(out), "--phi", "0"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    back = load_pack(out.read_bytes())
    assert len(back) == len([k for k in sd if not k.endswith("num_batches_tracked")])
    for k in ("backbone_net.model._blocks.7._se_reduce.conv.weight", "bifpn.2.p4_w2", "translation_net.initial_translation_z.pointwise_conv.conv.bias"):
        assert np.array_equal(back[k], sd[k].numpy())
    r = subprocess.run([sys.executable, tool, str(ckpt), str(out), "--phi", "3"], capture_output=True, text=True)
    assert r.returncode != 0 and "does not match phi=3" in (r.stderr + r.stdout)


def test_corrupt_pack_is_refused_without_reading_out_of_bounds():
    """HEPW parser: offsets / sizes that wrap around 2^64, dims whose product overflows and tensors overlapping the
    header are rejected (hep_create_from_memory parses the pack before it looks for a device)."""
    import struct
    l = _capi.lib()

    def pack(dims, off, nb, total=4096):
        name = b"t"
        blob = b"HEPW" + struct.pack("<II", 1, 1) + struct.pack("<H", len(name)) + name + bytes([len(dims)])
        blob += b"".join(struct.pack("<I", d) for d in dims) + struct.pack("<QQ", off, nb)
        return blob + b"\0" * (total - len(blob))

    for dims, off, nb in (([2], 2 ** 64 - 4, 8), ([2 ** 31, 2 ** 31, 4], 64, 0), ([2], 4, 8), ([1024], 64, 4096 * 4), ([2], 66, 8)):
        blob = pack(dims, off, nb)
        h = ctypes.c_void_p()
        rc = l.hep_create_from_memory(blob, len(blob), 0, 256, 1, _capi.HEP_F32, 0, 0, ctypes.byref(h))
        assert rc == -2 and b"weight pack" in l.hep_last_error(), (dims, off, nb, rc, l.hep_last_error())


def test_no_cpp_exception_crosses_the_c_abi():
    """include/hep.h: "never throws".  Every extern "C" entry point is a function-try-block (csrc/hep_api.cpp): a std::bad_alloc /
    length_error raised while a hostile pack is copied, or anything else a C++ body could throw, comes back as HEP_ERR_INTERNAL
    with a message instead of unwinding into a ctypes / P/Invoke caller (which terminates the process).  Truncated tables and
    oversized headers are plain HEP_ERR_PACK."""
    import struct
    l = _capi.lib()
    h = ctypes.c_void_p()
    good = pack_bytes(seeded_state_dict(0, 0))
    # (1) a size far beyond what the host can allocate: the parser's copy of the pack throws before a byte is read
    buf = ctypes.create_string_buffer(bytes(good[:4096]), 4096)
    for huge in (1 << 62, (1 << 63) + 5):
        rc = l.hep_create_from_memory(buf, huge, 0, 256, 1, _capi.HEP_F32, 0, 0, ctypes.byref(h))
        assert rc == -5 and not h.value, (huge, rc, l.hep_last_error())
        assert b"memory" in l.hep_last_error() or b"internal error" in l.hep_last_error()
    # (2) truncated packs: every prefix that cuts the table or the data
    for cut in (12, 13, 20, 100, len(good) // 2, len(good) - 100):
        rc = l.hep_create_from_memory(good[:cut], cut, 0, 256, 1, _capi.HEP_F32, 0, 0, ctypes.byref(h))
        assert rc == -2 and b"weight pack" in l.hep_last_error(), (cut, rc, l.hep_last_error())
    # (3) oversized headers: a tensor count, a name length and a rank no file of this size can hold
    hdr = lambda count: b"HEPW" + struct.pack("<II", 1, count)
    for blob in (hdr(0xFFFFFFFF) + bytes(64),
                 hdr(1) + struct.pack("<H", 0xFFFF) + b"t" * 32,
                 hdr(1) + struct.pack("<H", 1) + b"t" + bytes([200]) + bytes(64),
                 hdr(3) + struct.pack("<H", 1) + b"t" + bytes([1]) + struct.pack("<I", 2) + struct.pack("<QQ", 64, 8) + bytes(64)):
        rc = l.hep_create_from_memory(blob, len(blob), 0, 256, 1, _capi.HEP_F32, 0, 0, ctypes.byref(h))
        assert rc == -2 and b"weight pack" in l.hep_last_error() and not h.value, (blob[:24], rc, l.hep_last_error())
    # the library still works afterwards
    assert l.hep_abi_version() == 1 and l.hep_anchors(256, None, None) == 12276


def test_evaluator_host_logic(tmp_path):
    """hmd_ego_pose_amd.evaluate without a GPU: the Linemod-folder reader (binary and ASCII PLY, yml, split file, mask
    boxes), Rodrigues both ways against scipy, IoU with the +1 convention, AP, and the post-filter (a15) against the
    oracle's restatement of eval/common.py:419-447."""
    from scipy.spatial.transform import Rotation
    from hmd_ego_pose_amd import evaluate as E
    from oracle import decode_ref as D
    from tests._util import make_linemod_folder
    pts, truth = make_linemod_folder(str(tmp_path / "ds"), n=4)
    ds = E.LinemodFolder(str(tmp_path / "ds"))
    assert len(ds) == 4 and ds.diameter == 180.0 and np.array_equal(ds.points, pts)
    for i, t in enumerate(truth):
        assert np.array_equal(ds.load_image(i), t["image"]) and np.array_equal(ds.annotations[i]["bbox"], t["bbox"])
        assert np.allclose(E.axis_angle_to_matrix(ds.annotations[i]["rotation"]), t["R"], atol=1e-12)
        assert np.allclose(ds.annotations[i]["translation"], t["t"])
        assert ds.camera_input(i, 1.0).tolist() == [480.0, 480.0, 128.0, 128.0, 1000.0, 1.0]
        assert np.array_equal(ds.annotations[i]["coords_3d"], t["joints"])          # hands/<frame>_coords_3d.npy (generators/colibri.py:430-436)
    pts2, _ = make_linemod_folder(str(tmp_path / "ds2"), n=1, binary_ply=False)
    assert np.array_equal(E.LinemodFolder(str(tmp_path / "ds2")).points, pts2)
    rng = np.random.Generator(np.random.PCG64(3))
    for rv in list(rng.standard_normal((20, 3))) + [np.array([math.pi - 1e-8, 0, 0]), np.array([0.0, 0.0, 0.0]), np.array([1e-11, 0, 0]), np.array([0, 2.0, 0])]:
        R = Rotation.from_rotvec(rv).as_matrix()
        assert np.allclose(E.axis_angle_to_matrix(rv), R, atol=1e-12) and np.allclose(E.axis_angle_to_matrix(rv), D.rodrigues(rv), atol=1e-15)
        back = E.matrix_to_axis_angle(R)
        assert np.allclose(E.axis_angle_to_matrix(back), R, atol=1e-9), rv
    # 2D reprojection (eval/common.py:646-679): a pure sideways shift of dx at depth Z moves every projected point by fx * dx / Z pixels
    P3 = rng.standard_normal((50, 3)) * 10
    K = np.array([[480.0, 0, 128.0], [0, 470.0, 120.0], [0, 0, 1.0]])
    assert abs(E.reprojection_distance(P3, np.eye(3), np.array([0, 0, 500.0]), np.eye(3), np.array([0, 0, 500.0]), K)) == 0.0
    d2 = E.reprojection_distance(np.zeros((3, 3)), np.eye(3), np.array([0, 0, 400.0]), np.eye(3), np.array([2.0, 0, 400.0]), K)
    assert abs(d2 - 480.0 * 2.0 / 400.0) < 1e-12
    # IoU, +1 convention: identical boxes 1.0; [0,0,9,9] vs [5,5,14,14]: 25 / (100 + 100 - 25)
    ov = E.compute_overlap(np.array([[0, 0, 9, 9], [20, 20, 30, 30]], float), np.array([[0, 0, 9, 9], [5, 5, 14, 14]], float))
    assert ov[0, 0] == 1.0 and abs(ov[0, 1] - 25 / 175) < 1e-15 and ov[1, 0] == 0.0
    assert abs(E.compute_ap(np.array([0.5, 0.5, 1.0]), np.array([1.0, 0.5, 2 / 3])) - (0.5 * 1.0 + 0.5 * 2 / 3)) < 1e-15
    # post-filter
    M = 8
    det = {"boxes": torch.arange(M * 4, dtype=torch.float32).reshape(M, 4), "scores": torch.tensor([0.9, 0.3, 0.7, 0.7, 0.06, 0.04, -1.0, -1.0]),
           "labels": torch.tensor([0, 0, 0, 0, 0, 0, -1, -1], dtype=torch.int32), "rotation": torch.linspace(-1, 1, M * 3).reshape(M, 3),
           "translation": torch.randn(M, 3), "hand": torch.randn(M, 63)}
    b, s_, _l, r, t, _h = E.post_filter(det, 0.8, 0.05, 4)
    ob, os_, orr, ot = D.post_filter(det["boxes"].numpy(), det["scores"].numpy(), det["rotation"].numpy(), det["translation"].numpy(), 0.8, 0.05, 4)
    assert s_.tolist() == [np.float32(0.9), np.float32(0.7), np.float32(0.7), np.float32(0.3)]
    assert np.array_equal(b, ob) and np.array_equal(s_, os_) and np.array_equal(r, orr) and np.array_equal(t, ot)
    assert np.array_equal(b[1], det["boxes"][2].numpy() / np.float32(0.8))         # equal scores: lower row first


# ---- ONNX initialiser reader ----
def _pb_varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        c = v & 0x7F; v >>= 7
        out.append(c | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _pb_field(num, wt, payload):
    return _pb_varint((num << 3) | wt) + (payload if wt == 0 else _pb_varint(len(payload)) + payload)


def _onnx_tensor(name, arr, how):
    """TensorProto per the published onnx.proto: dims 1, data_type 2, float_data 4, int64_data 7, name 8, raw_data 9."""
    dt = {np.dtype("float32"): 1, np.dtype("int64"): 7, np.dtype("float16"): 10}[arr.dtype]
    if how == "packed_dims":
        body = _pb_field(1, 2, b"".join(_pb_varint(int(d)) for d in arr.shape))
    else:
        body = b"".join(_pb_field(1, 0, _pb_varint(int(d))) for d in arr.shape)
    body += _pb_field(2, 0, _pb_varint(dt))
    if how == "float_data" and dt == 1:
        body += _pb_field(4, 2, arr.astype("<f4").tobytes())
    elif how == "int64_data" and dt == 7:
        body += _pb_field(7, 2, b"".join(_pb_varint(int(x)) for x in arr.reshape(-1)))
    else:
        body += _pb_field(9, 2, arr.astype(arr.dtype.newbyteorder("<")).tobytes())
    return body + _pb_field(8, 2, name.encode())


def _onnx_model(tensors):
    graph = _pb_field(1, 2, _pb_field(4, 2, b"Conv"))                       # a node in front of the initialisers (skipped)
    graph += _pb_field(2, 2, b"main_graph")
    for name, arr, how in tensors:
        graph += _pb_field(5, 2, _onnx_tensor(name, arr, how))
    return _pb_field(1, 0, _pb_varint(7)) + _pb_field(2, 2, b"pytorch") + _pb_field(7, 2, graph) + _pb_field(8, 2, _pb_field(2, 0, _pb_varint(11)))


def test_onnx_initializer_reader_roundtrip_and_pack(tmp_path):
    """hmd_ego_pose_amd/onnx_init.py against files written here from the published onnx.proto layout (the encodings a real
    exporter does not happen to use; the real export is test_onnx_reader_against_the_real_eval_mode_export below): raw_data /
    float_data / int64_data encodings, packed and unpacked dims, negative int64, a scalar; then a whole phi-0 state_dict with
    `model.` prefixes through tools/pack_weights.py's path, and the refusal of a file whose initialisers are anonymous AND
    whose graph is not the trace of the architecture (no Conv nodes to map them back by)."""
    import torch
    from hmd_ego_pose_amd import param_spec, seeded_state_dict
    from hmd_ego_pose_amd.onnx_init import read_initializers, state_dict_from_onnx
    from hmd_ego_pose_amd.weights import load_pack, pack_bytes
    rng = np.random.default_rng(0)
    ts = [("a.weight", rng.standard_normal((4, 3, 3, 3)).astype(np.float32), "raw"),
          ("b.bias", rng.standard_normal((7,)).astype(np.float32), "float_data"),
          ("c.num_batches_tracked", np.array(-5, dtype=np.int64), "int64_data"),
          ("d.idx", np.array([[1, -2], [3, 1 << 40]], dtype=np.int64), "packed_dims"),
          ("e.half", rng.standard_normal((2, 5)).astype(np.float16), "raw")]
    got = read_initializers(_onnx_model(ts))
    assert list(got) == [t[0] for t in ts]
    for name, arr, _ in ts:
        assert got[name].dtype == arr.dtype and got[name].shape == arr.shape and np.array_equal(got[name], arr), name
    with pytest.raises(ValueError):
        read_initializers(_onnx_model(ts)[:-40] + b"\xff")                  # truncated / corrupt
    with pytest.raises(ValueError):
        read_initializers(b"\x0a\x03abc")                                   # not a ModelProto with a graph
    # a whole network: every state_dict entry as an initialiser, `model.` prefixed like TrainModelWithLoss saves them
    sd = seeded_state_dict(0, 3)
    blob = _onnx_model([("model." + k, v.numpy(), "raw" if i % 2 else ("float_data" if v.dtype == torch.float32 else "int64_data")) for i, (k, v) in enumerate(sd.items())])
    path = tmp_path / "m.onnx"; path.write_bytes(blob)
    state = state_dict_from_onnx(str(path), 0)
    assert set(state) == {k for k, _ in param_spec(0)} and all(torch.equal(state[k], sd[k]) for k in state)
    a, b = load_pack(pack_bytes(state)), load_pack(pack_bytes(sd))
    assert a.keys() == b.keys() and all(np.array_equal(a[k], b[k]) for k in a)
    anon = _onnx_model([(f"onnx::Conv_{i}", v.numpy(), "raw") for i, (k, v) in enumerate(sd.items()) if v.dtype == torch.float32])
    with pytest.raises(ValueError, match="Conv nodes"):
        state_dict_from_onnx(anon, 0)


def test_onnx_reader_against_the_real_eval_mode_export():
    """The ONNX reader pinned to a REAL exporter output.  tests/golden/onnx_eval_phi0.structure.gz is what torch.onnx.export wrote for
    the reference's HMDEgoPose (eval mode, opset 9, exactly as the reference's export_to_onnx: BatchNorm folded into 153
    anonymous convolutions), every byte of it except the tensor payloads (tests/golden/make_golden_onnx.py).  The payloads are
    rebuilt here from the seeded weights - the tensors that kept their names bit for bit (sha256 recorded from the real file),
    the folded ones with the numpy restatement of the fold that the generating script measured at 2.4e-7 from the real
    payloads - and the file is read back: state_dict_from_onnx maps the anonymous initialisers through the order of the 344
    Conv nodes, splits the head towers' per-level folds into one shared weight + five scales, and the recovered state_dict must
    drive the oracle to the same outputs as the original one."""
    import torch
    from hmd_ego_pose_amd import seeded_state_dict
    from hmd_ego_pose_amd.onnx_init import conv_exec_order, read_nodes, state_dict_from_onnx
    from oracle import efficientpose_ref as R
    from tests._util import rebuilt_real_onnx_export
    blob, meta, written = rebuilt_real_onnx_export()
    assert len(blob) == meta["file_bytes"]
    phi = meta["phi"]
    sd = seeded_state_dict(phi, 0)
    convs = [n for n in read_nodes(blob) if n[0] == "Conv"]
    seq = conv_exec_order(phi)
    assert len(convs) == len(seq) == meta["conv_nodes"] == 344
    for (_op, ins, _o, _n), (wk, _bk, _bn) in zip(convs, seq):
        assert ins[1] not in sd or ins[1] == wk                   # a convolution that kept its name sits where the trace order says
    want = {k: v for k, v in written.items() if k not in sd}
    assert len(want) == meta["folded_initializers"] == 306 and len(written) == len(meta["tensors"]) >= 440
    for name, a in written.items():
        e = meta["tensors"][name]
        assert list(a.shape) == e["shape"], name
        if "sha256" in e:                                         # the exporter stored the state_dict tensor verbatim
            assert hashlib.sha256(np.ascontiguousarray(a, "<f4").tobytes()).hexdigest() == e["sha256"], name
        else:                                                     # folded: float64 sums of the REAL payload
            assert abs(float(a.astype(np.float64).sum()) - e["sum"]) <= 1e-6 * max(1.0, e["abssum"]), name
    assert meta["max_abs_diff_real_fold_vs_numpy_fold"] <= 1e-6 and max(meta["max_abs_diff_recovered_state_dict_forward_vs_reference"]) <= 1e-5
    rec = state_dict_from_onnx(blob, phi)
    assert list(rec) == list(sd)
    # tensors that were never folded come back bit for bit; a folded conv comes back as (folded weight, identity BatchNorm + folded bias)
    for k in ("backbone_net.model._blocks.7._se_reduce.conv.weight", "bifpn.1.p4_w2", "bifpn.2.conv4_up.depthwise_conv.conv.weight", "hand_net.initial_hand_coords.pointwise_conv.conv.bias"):
        assert torch.equal(rec[k], sd[k]), k
    p = "backbone_net.model._blocks.3"
    assert np.array_equal(rec[p + "._project_conv.conv.weight"].numpy(), want[convs[[q[0] for q in seq].index(p + "._project_conv.conv.weight")][1][1]])
    assert float(rec[p + "._bn2.running_var"].min()) == 1.0 and float(rec[p + "._bn2.running_mean"].abs().max()) == 0.0
    x = torch.from_numpy(np.random.Generator(np.random.PCG64(3)).standard_normal((1, 3, 128, 128)).astype(np.float32))
    a, b = R.forward(sd, x, phi), R.forward(rec, x, phi)
    for u, v in zip(a[1:], b[1:]):
        assert float((u - v).abs().max()) <= 2e-4 * max(1.0, float(u.abs().max()))
    # a graph that is not the trace of this architecture is refused: one Conv node fewer
    data = bytes(blob)
    from hmd_ego_pose_amd.onnx_init import _fields_at
    g0 = [(s0, e0) for f, wt, s0, e0 in _fields_at(data, 0, len(data)) if f == 7 and wt == 2][0]
    with pytest.raises(ValueError, match="Conv nodes|trace order|shape"):
        nodes = [(s1, e1) for f1, wt1, s1, e1 in _fields_at(data, g0[0], g0[1]) if f1 == 1 and wt1 == 2]
        lo, hi = nodes[[i for i, (s1, e1) in enumerate(nodes) if b'"\x04Conv' in data[s1:e1]][40]]      # (op_type field: tag 0x22, length 4)
        cut = bytearray(data)
        cut[lo:hi] = data[lo:hi].replace(b'"\x04Conv', b'"\x04Conx')          # (same length: the enclosing length prefixes stay valid)
        state_dict_from_onnx(bytes(cut), phi)
