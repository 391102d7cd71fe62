"""CPU: the numpy decode oracle on hand-built known-answer cases (NMS, padding, post-filter,
ADD / ADD-S), i.e. the parts of the path whose reference implementation (TensorFlow, OpenCV,
Cython) cannot run here.  The expected answers are derived by hand from the rule stated in
oracle/decode_ref.py (greedy, IoU strictly greater than the threshold suppresses, ties by
lower index)."""
import math

import numpy as np
import pytest

from oracle import decode_ref as D


def _boxes():
    # 3 clusters; scores chosen so the expected survivors are obvious
    b = np.array([
        [10, 10, 50, 50],      # 0  s=.90  keep (best of cluster A)
        [12, 12, 52, 52],      # 1  s=.80  IoU with 0 = 0.82 -> suppressed
        [10, 10, 50, 90],      # 2  s=.70  IoU with 0 = 0.5 exactly -> NOT suppressed (strict >)
        [100, 100, 140, 140],  # 3  s=.95  keep (global best)
        [101, 101, 141, 141],  # 4  s=.95  tie with 3 -> index 3 first, 4 suppressed (IoU .905)
        [200, 200, 220, 220],  # 5  s=.40  below threshold
        [300, 300, 300, 340],  # 6  s=.99  degenerate (zero area): IoU 0 with everything -> kept
        [300, 300, 300, 340],  # 7  s=.60  identical degenerate box: IoU 0 -> kept too
    ], dtype=np.float32)
    s = np.array([.90, .80, .70, .95, .95, .40, .99, .60], dtype=np.float32)
    return b, s


def test_iou_known_values():
    b, _ = _boxes()
    assert abs(float(D.iou_xyxy(b[0], b[2])) - 0.5) < 1e-7
    assert abs(float(D.iou_xyxy(b[0], b[1])) - (38 * 38) / (2 * 1600 - 38 * 38)) < 1e-6
    assert float(D.iou_xyxy(b[6], b[7])) == 0.0
    # coordinate-order invariance: the reference passes (x1,y1,x2,y2) to a (y1,x1,y2,x2) API
    sw = lambda v: v[[1, 0, 3, 2]]
    assert D.iou_xyxy(sw(b[0]), sw(b[1])) == D.iou_xyxy(b[0], b[1])


def test_nms_known_answer_and_padding():
    b, s = _boxes()
    n = len(s)
    rot = np.arange(n * 3, dtype=np.float32).reshape(n, 3)
    tr = -np.arange(n * 3, dtype=np.float32).reshape(n, 3)
    hand = np.tile(np.arange(n, dtype=np.float32)[:, None], (1, 63))
    out = D.filter_detections(b, s[:, None], rot, tr, hand, score_threshold=0.5, max_detections=6, nms_threshold=0.5)
    boxes, scores, labels, r, t, h, idx = out
    assert idx.tolist() == [6, 3, 0, 2, 7, -1]
    assert scores[:5].tolist() == [s[6], s[3], s[0], s[2], s[7]] and scores[5] == -1
    assert labels.tolist() == [0, 0, 0, 0, 0, -1] and labels.dtype == np.int32
    assert np.array_equal(boxes[1], b[3]) and np.all(boxes[5] == -1) and np.all(h[5] == -1) and np.all(r[5] == -1)
    assert np.array_equal(r[2], rot[0]) and np.array_equal(t[3], tr[2]) and h[4, 0] == 7
    # max_detections caps the greedy pass itself (max_output_size)
    out2 = D.filter_detections(b, s[:, None], rot, tr, hand, score_threshold=0.5, max_detections=2, nms_threshold=0.5)
    assert out2[6].tolist() == [6, 3]
    # nothing above threshold -> all padding
    out3 = D.filter_detections(b, s[:, None] * 0, rot, tr, hand, max_detections=3)
    assert out3[6].tolist() == [-1, -1, -1] and np.all(out3[0] == -1)


def test_filter_with_several_classes_known_answer():
    """layers.py:347-380 worked by hand.  Boxes A, B overlap (IoU 81 / 119 = 0.68), C is far away.
    class 0: A .9, B .8, C .7 -> A kept, B suppressed by A, C kept;  class 1: B .95, A .6, C .2 -> B kept, A suppressed by B.
    Pairs class by class: (A,0) (C,0) (B,1) with scores .9 .7 .95 -> top_k: (B,1) (A,0) (C,0).
    Not class-specific (:359-362): best class per anchor: A .9 / 0, B .95 / 1, C .7 / 0 -> B kept, A suppressed, C kept."""
    b = np.array([[0, 0, 10, 10], [1, 1, 11, 11], [50, 50, 60, 60]], np.float32)
    cls = np.array([[.9, .6], [.8, .95], [.7, .2]], np.float32)
    rot = np.arange(9, dtype=np.float32).reshape(3, 3)
    hand = np.tile(np.arange(3, dtype=np.float32)[:, None], (1, 63))
    boxes, scores, labels, r, t, h, idx = D.filter_detections(b, cls, rot, -rot, hand, score_threshold=0.5, max_detections=5)
    assert idx.tolist() == [1, 0, 2, -1, -1] and labels.tolist() == [1, 0, 0, -1, -1]
    assert scores.tolist() == [np.float32(.95), np.float32(.9), np.float32(.7), -1, -1]
    assert np.array_equal(boxes[0], b[1]) and np.array_equal(r[1], rot[0]) and h[2, 5] == 2 and np.all(t[3:] == -1)
    out = D.filter_detections(b, cls, rot, -rot, hand, score_threshold=0.5, max_detections=5, class_specific_filter=False)
    assert out[6].tolist() == [1, 2, -1, -1, -1] and out[2].tolist() == [1, 0, -1, -1, -1]
    # the same anchor may come out once per class; max_detections caps every class's pass AND the final top_k
    far = np.array([[0, 0, 10, 10], [20, 20, 30, 30], [50, 50, 60, 60]], np.float32)
    out = D.filter_detections(far, cls, rot, -rot, hand, score_threshold=0.5, max_detections=2)
    assert out[6].tolist() == [1, 0] and out[2].tolist() == [1, 0]            # class 0 keeps (A, B), class 1 keeps (B, A): .95, .9
    out = D.filter_detections(far, cls, rot, -rot, hand, score_threshold=0.1, max_detections=6)
    assert out[6].tolist() == [1, 0, 1, 2, 0, 2] and out[2].tolist() == [1, 0, 0, 0, 1, 1]
    # equal scores: top_k keeps the earlier pair of the class-by-class concatenation
    tie = np.array([[.7, .7], [.7, .2], [.2, .7]], np.float32)
    out = D.filter_detections(far, tie, rot, -rot, hand, score_threshold=0.5, max_detections=4)
    assert out[6].tolist() == [0, 1, 0, 2] and out[2].tolist() == [0, 0, 1, 1]
    # one class: both modes are the same pass
    one = cls[:, :1]
    for a, c in zip(D.filter_detections(b, one, rot, -rot, hand), D.filter_detections(b, one, rot, -rot, hand, class_specific_filter=False)):
        assert np.array_equal(a, c)


def test_post_filter_matches_evaluate_loop():
    b, s = _boxes()
    rot = np.ones((8, 3), np.float32) * 0.5
    tr = np.ones((8, 3), np.float32)
    bx, sc, r, t = D.post_filter(b, s, rot, tr, scale=0.8, score_threshold=0.75, max_detections=3)
    assert sc.tolist() == [s[6], s[3], s[4]]            # stable sort keeps 3 before 4
    assert np.allclose(bx[1], b[3] / 0.8) and np.allclose(r, 0.5 * math.pi)


def test_add_and_add_s_on_synthetic_poses():
    rng = np.random.Generator(np.random.PCG64(7))
    pts = rng.uniform(-100, 100, (2500, 3))
    diameter = 380.031          # datasets/syn_colibri/models/models_info.yml:1
    R = D.rodrigues(np.array([0.1, -0.2, 0.3]))
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(R) - 1) < 1e-12
    t = np.array([10.0, -5.0, 500.0])
    ok, d = D.add_metric(pts, diameter, R, t, R, t + np.array([3.0, 4.0, 0.0]))
    assert ok and abs(d - 5.0) < 1e-9                   # pure translation error = |dt| for every point
    ok, d = D.add_metric(pts, diameter, R, t, R, t + np.array([30.0, 40.0, 0.0]))
    assert (not ok) and abs(d - 50.0) < 1e-9            # 50 > 0.1 * 380.031
    # ADD-S: subsampling step = 2500//1000+1 = 3 -> 834 points; identical pose -> 0
    ok, d = D.add_s_metric(pts, diameter, R, t, R, t)
    assert ok and d == 0.0
    # ADD-S <= ADD always (nearest neighbour instead of corresponding point)
    R2 = D.rodrigues(np.array([0.1, -0.2, 0.35]))
    _, dadd = D.add_metric(pts[::3], diameter, R, t, R2, t)
    _, dadds = D.add_s_metric(pts, diameter, R, t, R2, t)
    assert 0 < dadds <= dadd + 1e-6
    # rodrigues known answer: 90 degrees about z
    Rz = D.rodrigues(np.array([0, 0, math.pi / 2]))
    assert np.allclose(Rz, [[0, -1, 0], [1, 0, 0], [0, 0, 1]], atol=1e-12)


def test_decode_boxes_properties():
    a, ta = D.anchors_for_size(256)
    z = np.zeros((1, a.shape[0], 4), np.float32)
    # zero deltas reproduce the (clipped) anchors
    assert np.allclose(D.decode_boxes(a, z, 256)[0], np.clip(a, 0, 255), rtol=0, atol=1e-4)   # centre/size round trip
    # translation: zero offsets at the principal point give Tx=Ty=0
    cam = np.array([[480, 480, 132, 132, 1000, 1.0]], np.float32)     # (16+0.5)*8 = 132 is a P3 cell centre
    raw = np.zeros((1, a.shape[0], 3), np.float32); raw[..., 2] = 0.5
    tr = D.decode_translation(ta, raw, cam)[0]
    centre = np.nonzero((ta[:, 0] == 132) & (ta[:, 1] == 132))[0]
    assert len(centre) > 0 and np.all(tr[centre, :2] == 0) and np.all(tr[:, 2] == 500)


def test_preprocess_image_restatement():
    """Known answers of the reference's preprocess arithmetic (colibri_common.py:633-651): pixel value 0 ->
    -mean/std, 255 -> (1-mean)/std per channel in float32; padding rows/columns are exactly zero."""
    from oracle import decode_ref as D
    img = np.zeros((200, 256, 3), np.uint8)
    img[1, 2] = 255
    out, scale = D.preprocess_image(img, 256)
    assert out.shape == (256, 256, 3) and out.dtype == np.float32 and scale == 1.0
    mean, std = np.array([0.485, 0.456, 0.406]), np.array([0.229, 0.224, 0.225])
    assert np.array_equal(out[0, 0], ((np.float32(0) - mean).astype(np.float32) / std).astype(np.float32))
    assert np.array_equal(out[1, 2], ((np.float32(1) - mean).astype(np.float32) / std).astype(np.float32))
    assert not out[200:].any()


def test_min_distances_match_the_compiled_reference():
    """oracle.min_distances (numpy) against c_min_distances of the reference's own calc_min_distances.h, compiled by
    oracle/Makefile from where it lies under /root/reference (oracle/_ref/libmindist.so): bit for bit, on clouds with
    close pairs, duplicates and large offsets; and add_s_metric's subsampling (step = n // 1000 + 1)."""
    import os
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libmindist.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libmindist.so not built (needs /root/reference: `make -C oracle`)")
    rng = np.random.Generator(np.random.PCG64(5))
    for n_g, n_p, scale, off in ((1, 1, 1.0, 0.0), (7, 13, 100.0, 0.0), (500, 500, 130.0, 2.5), (1000, 777, 0.05, 1000.0)):
        g = (rng.standard_normal((n_g, 3)) * scale + off).astype(np.float32)
        p = (rng.standard_normal((n_p, 3)) * scale + off).astype(np.float32)
        if n_p > 10:
            p[3] = g[min(2, n_g - 1)]                       # an exact hit
            p[5] = p[4]                                    # duplicates
        want = D.reference_min_distances(g, p)
        got = D.min_distances(g, p)
        assert got.dtype == np.float32 and np.array_equal(got, want), (n_g, n_p)
    # the metric: 2500 model points -> step 3 -> 834 points per cloud
    pts = (rng.standard_normal((2500, 3)) * 60).astype(np.float32)
    Rg, Rp = D.rodrigues(np.array([0.3, -1.2, 0.5])), D.rodrigues(np.array([0.31, -1.18, 0.52]))
    tg, tp = np.array([10., -20., 600.]), np.array([11., -19., 604.])
    ok, d = D.add_s_metric(pts, 150.0, Rg, tg, Rp, tp)
    ref = D.reference_min_distances((pts @ Rg.T + tg)[::3], (pts @ Rp.T + tp)[::3])
    assert len(ref) == 834 and d == float(np.mean(ref)) and ok == (d <= 15.0)


def test_preprocess_matches_the_imported_reference_and_resize_convention():
    """oracle.preprocess_image, no-resize branch: bit-identical (sha256) to what the REAL reference function returned for
    the same seeded frames (tests/golden/preprocess.npz, made by tests/golden/make_golden_preprocess.py with a cv2 stub
    whose resize is the identity).  Resize branch (parity unpinned: OpenCV's 8-bit INTER_LINEAR restated): known answers
    of the fixed-point convention - a constant image stays constant, x2 upsampling of a ramp interpolates at quarter
    positions, sizes follow int(side * scale)."""
    import hashlib
    import os
    import importlib.util
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("mgp", os.path.join(here, "golden", "make_golden_preprocess.py"))
    mgp = importlib.util.module_from_spec(spec); spec.loader.exec_module(mgp)
    fx = np.load(os.path.join(here, "golden", "preprocess.npz"))
    for name, img, size in mgp.cases():
        out, scale = D.preprocess_image(img, size)
        assert scale == 1.0 and out.dtype == np.float32
        assert hashlib.sha256(out.tobytes()).digest() == fx[name + "_sha256"].tobytes(), name
        assert np.array_equal(out.reshape(-1)[::997], fx[name + "_slice"])
    # resize convention
    const = np.full((100, 128, 3), 77, np.uint8)
    out, scale = D.preprocess_image(const, 256)
    assert scale == 2.0 and out.shape == (256, 256, 3)
    v = np.float32(np.float64(np.float32(np.float32(77) / np.float32(255.0))) - 0.485)
    assert np.all(out[:200, :, 0] == np.float32(np.float64(v) / 0.229)) and np.all(out[200:] == 0)
    ramp = np.tile((np.arange(8, dtype=np.uint8) * 16)[None, :, None], (8, 1, 3))
    up = D.resize_bilinear_u8(ramp, 16, 16)
    # output x samples source (x + 0.5) / 2 - 0.5: 0 -> clamped to 0, 1 -> 0.25 => 4, 2 -> 0.75 => 12, 3 -> 1.25 => 20 ...
    assert up[0, :6, 0].tolist() == [0, 4, 12, 20, 28, 36] and up[0, -1, 0] == 112 and np.all(up[:, :, 1] == up[:, :, 0])
    down = D.resize_bilinear_u8(np.tile(np.arange(16, dtype=np.uint8)[None, :, None] * 10, (4, 1, 3)), 8, 2)
    assert down.shape == (2, 8, 3) and down[0, :3, 0].tolist() == [5, 25, 45]          # midpoints of pixel pairs
    tall = np.zeros((300, 200, 3), np.uint8)
    out, scale = D.preprocess_image(tall, 256)
    assert abs(scale - 256 / 300) < 1e-15 and np.all(out[:, int(200 * scale):] == 0) and np.all(out[:, :int(200 * scale)] != 0)


def test_anchor_targets_match_the_imported_reference():
    """oracle/train_ref.py against the REAL reference's anchor_targets_bbox / compute_gt_annotations / bbox_transform
    (generators/utils/anchors.py with its own Cython IoU extension compiled out of tree by
    tests/golden/make_golden_targets.py): sha256-identical arrays for one box, no box, a partly outside image, a box equal
    to an anchor + a box that overlaps nothing; the core assignment also for several boxes (where the reference's own
    batch assembly raises on its coords_3d reshape)."""
    import hashlib
    import importlib.util
    import os
    from oracle import train_ref as T
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("mgt", os.path.join(here, "golden", "make_golden_targets.py"))
    mgt = importlib.util.module_from_spec(spec); spec.loader.exec_module(mgt)
    fx = np.load(os.path.join(here, "golden", "anchor_targets.npz"))
    anchors, _ = D.anchors_for_size(256)
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest()
    seen = 0
    for name, shape, boxes, labels, tt, coords in mgt.cases(anchors):
        K = boxes.shape[0]
        if K:
            pos, ign, arg = T.compute_gt_annotations(anchors, boxes)
            assert sha(pos.astype(np.uint8)) == fx[f"{name}_positive_sha256"].tobytes(), name
            assert sha(ign.astype(np.uint8)) == fx[f"{name}_ignore_sha256"].tobytes(), name
            assert sha(arg.astype(np.int64)) == fx[f"{name}_argmax_sha256"].tobytes(), name
            assert sha(T.bbox_transform(anchors, boxes[arg, :]).astype(np.float64)) == fx[f"{name}_bbox_transform_sha256"].tobytes(), name
            assert [int(pos.sum()), int(ign.sum())] == fx[f"{name}_counts"].tolist()
        if f"{name}_labels_sha256" in fx:
            seen += 1
            c = np.repeat(coords[:1], max(K, 1), 0) if K else np.zeros((0, 63))
            lab, reg, tra, crd = T.anchor_targets(anchors, [shape], [boxes], [labels], [tt], [c], 1)
            for key, arr in (("labels", lab), ("regression", reg), ("transformation", tra), ("coords", crd)):
                assert arr.dtype == np.float32 and sha(arr) == fx[f"{name}_{key}_sha256"].tobytes(), (name, key)
            st = reg[..., -1]
            assert [(st == -1).sum(), (st == 0).sum(), (st == 1).sum()] == fx[f"{name}_state_hist"].tolist()
    assert seen >= 3          # one, none, exact_anchor (the reference raises on its coords_3d reshape for the others)
    # known answers: a box equal to an anchor makes that anchor positive with zero regression targets; a box nothing
    # overlaps forces anchor 0 positive
    boxes = np.stack([anchors[4000].astype(np.float64), np.array([1000., 1000., 1010., 1010.])])
    pos, ign, arg = T.compute_gt_annotations(anchors, boxes)
    assert pos[4000] and arg[4000] == 0 and pos[0] and np.allclose(T.bbox_transform(anchors, boxes[arg, :])[4000], 0, atol=1e-6)


def test_losses_match_the_imported_reference():
    """oracle/train_ref.py::batch_losses against the REAL reference's batch_iterate (hmdegopose/loss.py:54-428), replayed
    from tests/golden/losses.npz (made by tests/golden/make_golden_losses.py on the seeded cases of tests/_util.py):
    typical images, an image batch without a single object anchor (rotation 0, translation NaN - what the reference
    returns), the full 12 276-anchor grid with 500 model points, one object anchor.  float32 in another summation
    order: relative 1e-5."""
    import os
    from oracle import train_ref as T
    from tests._util import loss_cases
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "losses.npz"))
    cases = loss_cases()
    assert set(cases) == set(fx.files)
    for name, c in cases.items():
        got = T.batch_losses(c["gt_classification"], c["classification"], c["gt_regression"], c["regression"], c["gt_transformation"],
                             c["transformation"], c["gt_hand"], c["hand"], c["model_points"], 3)
        want = fx[name]
        assert np.array_equal(np.isnan(got), np.isnan(want)), (name, got, want)
        ok = ~np.isnan(want)
        assert np.allclose(got[ok], want[ok], rtol=1e-5, atol=1e-7), (name, got, want)
    assert np.isnan(fx["empty"][3]) and fx["empty"][2] == 0.0


def test_webrtc_frame_path_restatement():
    """oracle.decode_ref.yv12_to_bgr / webrtc_frame_preprocess (the C# frame callback, Program.cs:140-205): known
    answers of the BT.601 limited-range conversion that OpenCV documents (parity unpinned against cv2 itself), the U / V
    exchange the app's YV12 reading of I420 bytes causes, crop position, channel order and padding."""
    h, w = 8, 12
    grey = np.full(h * w * 3 // 2, 128, np.uint8)
    assert (D.yv12_to_bgr(grey, h, w) == 130).all()                     # (128 - 16) * 1.164 = 130.4
    black, white = grey.copy(), grey.copy()
    black[: h * w], white[: h * w] = 16, 235
    assert (D.yv12_to_bgr(black, h, w) == 0).all() and (D.yv12_to_bgr(white, h, w) == 255).all()
    redish = grey.copy()
    redish[h * w: h * w + (h // 2) * (w // 2)] = 240                     # the FIRST chroma plane: V for a YV12 reader (U in I420)
    b, g, r = D.yv12_to_bgr(redish, h, w)[0, 0]
    assert r == 255 and b == 130 and g < 130                             # V raised -> red up, green down, blue unchanged
    # geometry: a frame whose centre crop is constant gives a constant image; channel order B, G, R; no padding for a square crop
    H, W, S = 480, 640, 256
    rng = np.random.Generator(np.random.PCG64(3))
    buf = rng.integers(0, 256, H * W * 3 // 2, dtype=np.uint8)
    out, scale = D.webrtc_frame_preprocess(buf, H, W, S)
    assert out.shape == (S, S, 3) and out.dtype == np.float32 and scale == 0.5
    bgr = D.yv12_to_bgr(buf, H, W)[(H - 256) // 2:(H - 256) // 2 + 256, (W - 256) // 2:(W - 256) // 2 + 256]
    # 256 -> 512 -> 256 with half-pixel centres is a [1 6 1] / 8-like blur: the result stays within the crop's local range
    lo = np.minimum.reduce([np.roll(bgr, (dy, dx), (0, 1)) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]).astype(np.float32)
    hi = np.maximum.reduce([np.roll(bgr, (dy, dx), (0, 1)) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]).astype(np.float32)
    mean, std = np.array([0.485, 0.456, 0.406], np.float32), np.array([0.229, 0.224, 0.225], np.float32)
    back = out * std + mean
    inner = (slice(2, -2), slice(2, -2))
    assert (back[inner] * 255 >= lo[inner] - 1.01).all() and (back[inner] * 255 <= hi[inner] + 1.01).all()
