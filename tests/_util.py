"""Shared helpers for the tests: seeded inputs (same recipe as tests/golden/make_golden.py)
and golden-digest comparison."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = {  # tag -> (phi, size, batch, seed, input kind)
    "phi0_s256_b2_seed0": (0, 256, 2, 0, "normal"),
    "phi0_s256_b1_seed1": (0, 256, 1, 1, "uniform"),
    "phi3_s512_b1_seed0": (3, 512, 1, 0, "normal"),
    # the BENCHMARKED shapes (BASELINE configs[1] and configs[3]): the launch plan depends on the batch (multi-pass fronts by
    # rounds of workgroups, two tiles per workgroup in the boundary kernels, the split-K tile of the project GEMMs)
    "phi0_s256_b16_seed0": (0, 256, 16, 0, "normal"),
    "phi3_s512_b8_seed0": (3, 512, 8, 0, "normal"),
    # the regime of the reference's only published latency ("effnet_b0_512", FP32, batch 1: unity-sandbox/WebRTCNetCoreSandbox/Program.cs:24-33),
    # what bench.py's latency_b1 block times
    "phi0_s512_b1_seed0": (0, 512, 1, 0, "normal"),
}
CLASS_CASES = {  # tag -> (phi, size, batch, seed, input kind, num_classes): the classifier header with more than one class
    "phi0_s256_b2_seed0_k3": (0, 256, 2, 0, "normal", 3),
}
CAMS = np.array([[480, 480, 128, 128, 1000, 1.0],
                 [572.4114, 573.57043, 325.2611, 242.04899, 1000, 0.8]], dtype=np.float32)


def seeded_input(shape, seed, kind="normal"):
    rng = np.random.Generator(np.random.PCG64([seed, 0x1234]))
    a = rng.standard_normal(shape) if kind == "normal" else rng.random(shape)
    return a.astype(np.float32)


def golden_meta():
    with open(os.path.join(GOLDEN, "golden_meta.json")) as f:
        return json.load(f)


def golden_case(tag):
    return golden_meta()[tag], np.load(os.path.join(GOLDEN, f"net_{tag}.npz"))


def strides_for(size, key, batch=1):
    """Stride of the committed slice of a flattened tensor (primes; the batch-8 / batch-16 cases keep every ~16th sample of
    the small-batch ones so that their fixtures stay small)."""
    big = batch >= 8
    if key.startswith("trace_"):
        return (16139 if big else 1009) if size == 256 else (131071 if big else 8191)
    return (1543 if big else 97) if size == 256 else (6353 if big else 397)


def check_digest(name, arr, info, ref_slice, stride, atol, rtol=0.0):
    """Compare a full tensor against the committed strided slice (elementwise) and the
    float64 sum / abs-sum (global)."""
    a = np.ascontiguousarray(arr, dtype=np.float32).reshape(-1)
    assert list(np.shape(arr)) == info["shape"], (name, np.shape(arr), info["shape"])
    got = a[::stride]
    err = np.abs(got.astype(np.float64) - ref_slice.astype(np.float64))
    tol = atol + rtol * np.abs(ref_slice.astype(np.float64))
    assert np.all(err <= tol), f"{name}: max err {err.max():.3e} (tol {atol:g}+{rtol:g}*|ref|) at {int(err.argmax())}"
    n = a.size
    s = float(a.astype(np.float64).sum())
    sa = float(np.abs(a.astype(np.float64)).sum())
    assert abs(sa - info["abssum"]) <= n * atol + rtol * info["abssum"] + 1e-6, (name, sa, info["abssum"])
    assert abs(s - info["sum"]) <= n * atol + rtol * info["abssum"] + 1e-6, (name, s, info["sum"])
    return float(err.max())


def make_linemod_folder(root, n=5, size=256, seed=0, binary_ply=True):
    """A tiny synthetic dataset in the Linemod layout the reference's ColibriGenerator reads (data/01/{rgb,mask}/*.png,
    gt_0.yml, info_0.yml, test_0.txt, models/obj_01.ply, models_info.yml).  Returns the ground truth it wrote."""
    import os
    import yaml
    from PIL import Image
    from scipy.spatial.transform import Rotation
    rng = np.random.Generator(np.random.PCG64(seed))
    obj = os.path.join(root, "data", "01")
    for d in ("rgb", "mask", "hands"):
        os.makedirs(os.path.join(obj, d), exist_ok=True)
    os.makedirs(os.path.join(root, "models"), exist_ok=True)
    pts = (rng.standard_normal((1500, 3)) * np.array([40.0, 25.0, 60.0])).astype(np.float32)
    with open(os.path.join(root, "models", "obj_01.ply"), "wb") as f:
        if binary_ply:
            f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\nelement face 0\nproperty list uchar int vertex_indices\nend_header\n" % len(pts))
            rec = np.zeros(len(pts), dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1")])
            rec["x"], rec["y"], rec["z"] = pts[:, 0], pts[:, 1], pts[:, 2]
            f.write(rec.tobytes())
        else:
            f.write(b"ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nend_header\n" % len(pts))
            for p in pts:
                f.write(("%r %r %r\n" % (float(p[0]), float(p[1]), float(p[2]))).encode())
    with open(os.path.join(root, "models", "models_info.yml"), "w") as f:
        yaml.safe_dump({1: {"diameter": 180.0, "min_x": -100.0, "min_y": -80.0, "min_z": -150.0, "size_x": 200.0, "size_y": 160.0, "size_z": 300.0}}, f)
    gt, info, names, truth = {}, {}, [], []
    for i in range(n + 1):                        # one extra frame that is NOT in the split
        img = rng.integers(0, 256, (size, size, 3), dtype=np.uint8)
        Image.fromarray(img).save(os.path.join(obj, "rgb", f"{i:06d}.png"))
        m = np.zeros((size, size), np.uint8)
        x0, y0 = int(rng.integers(5, 100)), int(rng.integers(5, 100))
        x1, y1 = x0 + int(rng.integers(30, 120)), y0 + int(rng.integers(30, 120))
        m[y0:y1 + 1, x0:x1 + 1] = 255
        Image.fromarray(m).save(os.path.join(obj, "mask", f"{i:06d}.png"))
        R = Rotation.from_rotvec(rng.standard_normal(3)).as_matrix()
        t = np.array([rng.normal(0, 40), rng.normal(0, 40), 500 + rng.normal(0, 50)])
        joints = (rng.standard_normal((21, 3)) * 0.05).astype(np.float64)          # hand joints in metres (generators/colibri.py:430-436)
        np.save(os.path.join(obj, "hands", f"{i:06d}_coords_3d.npy"), joints)
        gt[i] = [{"cam_R_m2c": [float(v) for v in R.reshape(-1)], "cam_t_m2c": [float(v) for v in t], "obj_bb": [x0, y0, x1 - x0, y1 - y0],
                  "obj_id": 1, "drill_tip_transform": [10.0, -20.0, 30.0, 1.0]}]
        info[i] = {"cam_K": [480.0, 0.0, 128.0, 0.0, 480.0, 128.0, 0.0, 0.0, 1.0], "depth_scale": 1.0}
        if i < n:
            names.append(f"{i:06d}")
            truth.append(dict(image=img, bbox=np.array([x0, y0, x1, y1], np.float32), R=R, t=t, joints=joints))
    with open(os.path.join(obj, "gt_0.yml"), "w") as f:
        yaml.safe_dump(gt, f)
    with open(os.path.join(obj, "info_0.yml"), "w") as f:
        yaml.safe_dump(info, f)
    with open(os.path.join(obj, "test_0.txt"), "w") as f:
        f.write("\n".join(names) + "\n")
    return pts, truth


def loss_cases():
    """Seeded predictions / targets for the training-side losses (hmdegopose/loss.py:54-428): name -> dict of float32
    arrays.  Shared by tests/golden/make_golden_losses.py (the real reference), the oracle test and the GPU test."""
    out = {}
    for name, B, N, K, P, npos, seed in (("typical", 3, 2000, 2, 60, 24, 1), ("empty", 2, 500, 1, 20, 0, 2),
                                         ("full", 2, 12276, 1, 500, 40, 3), ("one_positive", 1, 777, 1, 33, 1, 4)):
        rng = np.random.Generator(np.random.PCG64([seed, 0x10555]))
        state = np.zeros((B, N), np.float32)
        for b in range(B):
            state[b, rng.choice(N, size=N // 10, replace=False)] = -1.0          # ignore
            if npos:
                state[b, rng.choice(N, size=npos + b, replace=False)] = 1.0       # object (a different count per image)
        cls = rng.integers(0, K, size=(B, N))
        labels = np.zeros((B, N, K), np.float32)
        pos = state == 1
        labels[pos, cls[pos]] = 1.0
        gt_classification = np.concatenate([labels, state[..., None]], axis=2)
        gt_regression = np.concatenate([rng.standard_normal((B, N, 4)).astype(np.float32) * 0.3, state[..., None]], axis=2)
        rot_t = rng.uniform(-1, 1, (B, N, 3)).astype(np.float32)
        tr_t = (rng.standard_normal((B, N, 3)) * 100).astype(np.float32)
        sym = rng.integers(0, 2, size=(B, N, 1)).astype(np.float32)
        gt_transformation = np.concatenate([rot_t, tr_t, sym, cls[..., None].astype(np.float32), state[..., None]], axis=2)
        gt_hand = np.concatenate([(rng.standard_normal((B, N, 63)) * 0.2).astype(np.float32), state[..., None]], axis=2)
        out[name] = dict(
            gt_classification=gt_classification.astype(np.float32),
            classification=(1.0 / (1.0 + np.exp(-rng.standard_normal((B, N, K)) * 3))).astype(np.float32),
            gt_regression=gt_regression.astype(np.float32),
            regression=(gt_regression[..., :4] + rng.standard_normal((B, N, 4)) * 0.15).astype(np.float32),
            gt_transformation=gt_transformation.astype(np.float32),
            transformation=np.concatenate([rot_t + rng.standard_normal((B, N, 3)).astype(np.float32) * 0.1,
                                           tr_t + rng.standard_normal((B, N, 3)).astype(np.float32) * 5], axis=2).astype(np.float32),
            gt_hand=gt_hand.astype(np.float32),
            hand=(gt_hand[..., :63] + rng.standard_normal((B, N, 63)) * 0.1).astype(np.float32),
            model_points=(rng.standard_normal((K, P, 3)) * np.array([40, 25, 60])).astype(np.float32))
    return out


def rebuilt_real_onnx_export():
    """tests/golden/onnx_eval_phi0.structure.gz (the reference model as torch.onnx.export wrote it, eval mode, opset 9, payloads blanked:
    tests/golden/make_golden_onnx.py) with its tensor payloads rebuilt from ``seeded_state_dict(0, 0)``: tensors that kept their names
    verbatim, BatchNorm-folded convolutions through the numpy restatement of the fold (2.4e-7 from the real payloads).
    Returns (file bytes, meta dict, {initialiser name: array written})."""
    import gzip
    import json
    import os
    from hmd_ego_pose_amd import seeded_state_dict
    from hmd_ego_pose_amd.arch import BN_EPS
    from hmd_ego_pose_amd.onnx_init import conv_exec_order, initializer_spans, read_nodes
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    meta = json.load(open(os.path.join(gdir, "onnx_eval_phi0.json")))
    blob = bytearray(gzip.decompress(open(os.path.join(gdir, "onnx_eval_phi0.structure.gz"), "rb").read()))
    sd = seeded_state_dict(meta["phi"], 0)
    convs = [n for n in read_nodes(bytes(blob)) if n[0] == "Conv"]
    want = {}
    for (_op, ins, _o, _n), (wk, bk, bn) in zip(convs, conv_exec_order(meta["phi"])):
        if ins[1] in sd:
            continue
        g, b, m, v = (sd[bn + k].numpy() for k in (".weight", ".bias", ".running_mean", ".running_var"))
        sc = (g / np.sqrt(v + np.float32(BN_EPS))).astype(np.float32)
        want[ins[1]] = (sd[wk].numpy() * sc[:, None, None, None]).astype(np.float32)
        want[ins[2]] = (((sd[bk].numpy() if bk else np.zeros_like(m)) - m) * sc + b).astype(np.float32)
    written = {}
    for name, lo, hi in initializer_spans(bytes(blob)):
        if name not in meta["tensors"]:
            continue
        a = sd[name].numpy() if name in sd else want[name]
        raw = np.ascontiguousarray(a, "<f4").tobytes()
        assert len(raw) == hi - lo, name
        blob[lo:hi] = raw
        written[name] = a
    return bytes(blob), meta, written
