"""ORACLE (test infrastructure, never on the product path): numpy restatement of the
decode half of the reference inference path - anchors, box / translation decode,
detection filter (score threshold -> greedy NMS -> top-k -> pad), the evaluate.py
post-filter, and the ADD / ADD-S pose metrics.

PINNED: anchors against the reference's own fixtures ``onnx-models/anchors_256.txt``,
``translation_anchors_{256,512}.txt`` (bit-exact, tests/golden/anchors_*.npz);
box/translation decode against outputs of the imported reference
(tests/golden/make_golden.py).  NMS / top-k follow TensorFlow semantics that cannot
be executed here (TensorFlow is absent from the reference tree and this image):
**parity unpinned** for ties; the convention is stated in ``nms_greedy``.
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np

SIZES = (32, 64, 128, 256, 512)
STRIDES = (8, 16, 32, 64, 128)
LEVELS = (3, 4, 5, 6, 7)
# stored as float32, used in float64 arithmetic (generators/utils/anchors.py:59-66)
RATIOS = np.array([1, 0.5, 2], dtype=np.float32)
SCALES = np.array([2 ** 0, 2 ** (1.0 / 3.0), 2 ** (2.0 / 3.0)], dtype=np.float32)


def base_anchors(base_size: int) -> np.ndarray:
    """generate_anchors, generators/utils/anchors.py:385-419: 9 boxes centred on 0,
    ordered scale-major (s0r0, s0r1, s0r2, s1r0, ...), float64."""
    n = len(RATIOS) * len(SCALES)
    a = np.zeros((n, 4))
    a[:, 2:] = base_size * np.tile(np.repeat(SCALES, len(RATIOS))[None], (2, 1)).T
    areas = a[:, 2] * a[:, 3]
    a[:, 2] = np.sqrt(areas / np.tile(RATIOS, len(SCALES)))
    a[:, 3] = a[:, 2] * np.tile(RATIOS, len(SCALES))
    a[:, 0::2] -= np.tile(a[:, 2] * 0.5, (2, 1)).T
    a[:, 1::2] -= np.tile(a[:, 3] * 0.5, (2, 1)).T
    return a


def anchors_for_size(size: int) -> Tuple[np.ndarray, np.ndarray]:
    """anchors_for_shape((size,size)), anchors.py:273-318 with shift (:321-347) and
    translation_shift (:350-382): cells row-major (y outer), centres (i+0.5)*stride,
    float64 math, one final cast to float32.  Returns (N,4) x1y1x2y2 and (N,3) cx,cy,stride."""
    boxes, trans = [], []
    for lvl, base, stride in zip(LEVELS, SIZES, STRIDES):
        fm = (size + 2 ** lvl - 1) // (2 ** lvl)                      # guess_shapes :257-270
        c = (np.arange(0, fm) + 0.5) * stride
        sx, sy = np.meshgrid(c, c)
        shifts = np.stack([sx.ravel(), sy.ravel(), sx.ravel(), sy.ravel()], axis=1)   # (K,4)
        b = (base_anchors(base)[None, :, :] + shifts[:, None, :]).reshape(-1, 4)
        t = np.concatenate([np.repeat(shifts[:, :2], 9, axis=0), np.full((fm * fm * 9, 1), float(stride))], axis=1)
        boxes.append(b)
        trans.append(t)
    return np.concatenate(boxes).astype(np.float32), np.concatenate(trans).astype(np.float32)


def decode_boxes(anchors: np.ndarray, regression: np.ndarray, size: int) -> np.ndarray:
    """format_bboxes, hmdegopose/loss.py:12-23 -> bbox_transform_inv layers.py:169-200 ->
    ClipBoxes layers.py:117-136.  float32 arithmetic like the torch original; deltas are
    ordered (ty, tx, th, tw); output (xmin, ymin, xmax, ymax) clipped to [0, size-1]."""
    a = anchors.astype(np.float32)[None]
    d = regression.astype(np.float32)
    two = np.float32(2)
    cxa = (a[..., 0] + a[..., 2]) / two
    cya = (a[..., 1] + a[..., 3]) / two
    wa = a[..., 2] - a[..., 0]
    ha = a[..., 3] - a[..., 1]
    ty, tx, th, tw = d[..., 0], d[..., 1], d[..., 2], d[..., 3]
    with np.errstate(over="ignore"):
        w = np.exp(tw) * wa
        h = np.exp(th) * ha
    cy = ty * ha + cya
    cx = tx * wa + cxa
    out = np.stack([cx - w / two, cy - h / two, cx + w / two, cy + h / two], axis=-1)
    return np.clip(out, np.float32(0), np.float32(size - 1)).astype(np.float32)


def decode_translation(t_anchors: np.ndarray, raw: np.ndarray, cam: np.ndarray) -> np.ndarray:
    """format_translation, loss.py:30-51 -> translation_transform_inv layers.py:142-166 ->
    CalculateTxTy layers.py:203-249.  cam[B,6] = fx, fy, px, py, tz_scale, image_scale."""
    ta = t_anchors.astype(np.float32)[None]
    r = raw.astype(np.float32)
    cam = cam.astype(np.float32)
    stride = ta[..., 2]
    x = ta[..., 0] + r[..., 0] * stride
    y = ta[..., 1] + r[..., 1] * stride
    fx, fy, px, py, tzs, isc = (cam[:, i][:, None] for i in range(6))
    x = x / isc - px
    y = y / isc - py
    tz = r[..., 2] * tzs
    return np.stack([x * tz / fx, y * tz / fy, tz], axis=-1).astype(np.float32)


def iou_xyxy(a: np.ndarray, b: np.ndarray) -> np.float32:
    """IoU as TensorFlow's non_max_suppression computes it: plain areas, no '+1' pixel
    convention, float32; degenerate (area<=0) boxes give 0."""
    f = np.float32
    ax0, ay0, ax1, ay1 = (f(min(a[0], a[2])), f(min(a[1], a[3])), f(max(a[0], a[2])), f(max(a[1], a[3])))
    bx0, by0, bx1, by1 = (f(min(b[0], b[2])), f(min(b[1], b[3])), f(max(b[0], b[2])), f(max(b[1], b[3])))
    aa = f((ax1 - ax0) * (ay1 - ay0))
    ab = f((bx1 - bx0) * (by1 - by0))
    if aa <= 0 or ab <= 0:
        return f(0)
    iw = f(max(f(min(ax1, bx1) - max(ax0, bx0)), f(0)))
    ih = f(max(f(min(ay1, by1) - max(ay0, by0)), f(0)))
    inter = f(iw * ih)
    return f(inter / f(f(aa + ab) - inter))


def nms_greedy(boxes: np.ndarray, scores: np.ndarray, max_out: int, iou_thr: float) -> np.ndarray:
    """tf.image.non_max_suppression as called at layers.py:332: candidates in descending
    score order (CONVENTION: equal scores -> lower index first); a candidate is dropped when
    its IoU with an already kept box is STRICTLY greater than iou_thr; stops at max_out."""
    order = np.lexsort((np.arange(len(scores)), -scores.astype(np.float64)))
    keep = []
    for i in order:
        ok = True
        for j in keep:
            if iou_xyxy(boxes[i], boxes[j]) > np.float32(iou_thr):
                ok = False
                break
        if ok:
            keep.append(int(i))
            if len(keep) >= max_out:
                break
    return np.asarray(keep, dtype=np.int64)


def filter_detections(boxes, classification, rotation, translation, hand,
                      score_threshold=0.5, max_detections=100, nms_threshold=0.5, class_specific_filter=True):
    """filter_detections, layers.py:264-400, one image.  Per class c (class_specific_filter, :347-354): anchors with
    classification[:, c] > threshold -> NMS (at most max_detections survivors, in NMS order) -> (anchor, c) pairs; the
    pairs of all classes concatenated class by class (:358).  Otherwise (:359-362) one pass over the best class of every
    anchor (max / first argmax).  Then top_k over the pairs' scores (:365-367; ties: the earlier pair) -> gather -> pad with
    -1 to max_detections rows.  Returns (boxes[M,4], scores[M], labels[M] int32, rotation[M,3], translation[M,3],
    hand[M,63], anchor_index[M] int32 (-1 padded)).  With one class both modes are the same pass."""
    cls = np.asarray(classification, dtype=np.float32)
    cls = cls.reshape(cls.shape[0], -1)

    def one_pass(s, labels):            # _filter_detections, layers.py:305-345
        cand = np.nonzero(s > np.float32(score_threshold))[0]
        kept = cand[nms_greedy(boxes[cand], s[cand], max_detections, nms_threshold)] if len(cand) else cand
        return kept.astype(np.int64), labels[kept].astype(np.int64)

    if class_specific_filter:
        parts = [one_pass(cls[:, c], np.full(cls.shape[0], c, np.int64)) for c in range(cls.shape[1])]
        anchors, labels = np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])
    else:
        anchors, labels = one_pass(cls.max(axis=1), cls.argmax(axis=1))
    s = cls[anchors, labels]
    order = np.lexsort((np.arange(len(s)), -s.astype(np.float64)))[:max_detections]
    idx, lab = anchors[order], labels[order]
    n = len(idx)

    def pad(a, w=None):
        shape = (max_detections,) if w is None else (max_detections, w)
        out = np.full(shape, -1, dtype=a.dtype)
        out[:n] = a
        return out

    return (pad(boxes[idx].astype(np.float32), 4), pad(s[order]), pad(lab.astype(np.int32)),
            pad(rotation[idx].astype(np.float32), 3), pad(translation[idx].astype(np.float32), 3),
            pad(hand[idx].astype(np.float32), hand.shape[1]), pad(idx.astype(np.int32)))


def post_filter(boxes, scores, rotations, translations, scale: float, score_threshold: float, max_detections: int):
    """_get_detections, eval/common.py:419-447: boxes/=scale, rotations*=pi, keep score>thr,
    argsort(-scores)[:max_detections]."""
    boxes = boxes / np.float32(scale)
    rotations = rotations * np.float32(math.pi)
    ind = np.nonzero(scores > score_threshold)[0]
    order = np.argsort(-scores[ind], kind="stable")[:max_detections]
    sel = ind[order]
    return boxes[sel], scores[sel], rotations[sel], translations[sel]


def rodrigues(rvec: np.ndarray) -> np.ndarray:
    """Axis-angle -> rotation matrix (cv2.Rodrigues at colibri_common.py:803-813), float64."""
    r = np.asarray(rvec, dtype=np.float64).reshape(3)
    th = np.linalg.norm(r)
    if th < 1e-12:
        return np.eye(3)
    k = r / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) * math.cos(th) + (1 - math.cos(th)) * np.outer(k, k) + math.sin(th) * K


def add_metric(points, diameter, R_gt, t_gt, R_pr, t_pr, thr=0.1):
    """check_6d_pose_add, eval/common.py:682-710: mean distance over ALL model points."""
    g = points @ R_gt.T + t_gt
    p = points @ R_pr.T + t_pr
    d = float(np.mean(np.linalg.norm(g - p, axis=-1)))
    return d <= diameter * thr, d


def add_s_metric(points, diameter, R_gt, t_gt, R_pr, t_pr, thr=0.1, max_points=1000):
    """check_6d_pose_add_s, eval/common.py:713-746 + c_min_distances calc_min_distances.h:24-35:
    step = n//max_points+1 subsampling of both clouds, float32 brute-force nearest distance."""
    g = (points @ R_gt.T + t_gt)
    p = (points @ R_pr.T + t_pr)
    step = points.shape[0] // max_points + 1
    dmin = min_distances(g[::step], p[::step])
    d = float(np.mean(dmin))
    return d <= diameter * thr, d


def reference_min_distances(points_gt: np.ndarray, points_pred: np.ndarray) -> np.ndarray:
    """The REAL reference: c_min_distances of generators/utils/calc_min_distances.h compiled by oracle/Makefile into
    oracle/_ref/libmindist.so (called like wrapper_c_min_distances, compute_overlap.pyx:103-121: both clouds cast to
    C-contiguous float32).  Raises FileNotFoundError when the library has not been built."""
    import ctypes
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libmindist.so")
    if not os.path.exists(path):
        raise FileNotFoundError(path + " (run `make -C oracle` where /root/reference is present)")
    lib = ctypes.CDLL(path)
    g = np.ascontiguousarray(points_gt, dtype=np.float32)
    p = np.ascontiguousarray(points_pred, dtype=np.float32)
    out = np.zeros((g.shape[0],), dtype=np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    lib.c_min_distances.argtypes = [fp, fp, fp, ctypes.c_int, ctypes.c_int]
    lib.c_min_distances.restype = None
    lib.c_min_distances(g.ctypes.data_as(fp), p.ctypes.data_as(fp), out.ctypes.data_as(fp), g.shape[0], p.shape[0])
    return out


def min_distances(points_gt: np.ndarray, points_pred: np.ndarray) -> np.ndarray:
    """numpy restatement of c_min_distances (calc_min_distances.h:24-35): float32 differences, products and sums in the
    C order ((d1*d1 + d2*d2) + d3*d3), (float) sqrt((double) .), minimum over the predicted cloud.  Pinned bit for bit
    against the compiled reference (tests/test_decode_oracle_cpu.py)."""
    g = np.ascontiguousarray(points_gt, dtype=np.float32)
    p = np.ascontiguousarray(points_pred, dtype=np.float32)
    diff = g[:, None, :] - p[None, :, :]
    d2 = (diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]) + diff[..., 2] * diff[..., 2]
    return np.sqrt(d2.astype(np.float64)).astype(np.float32).min(axis=1)


def resize_bilinear_u8(image: np.ndarray, nw: int, nh: int) -> np.ndarray:
    """cv2.resize(image, (nw, nh)) for uint8 HWC, INTER_LINEAR, as OpenCV's 8-bit fixed-point path is written
    (imgproc/resize.cpp; restated from its source, cv2 is not in this image: PARITY UNPINNED): with an explicit size the
    per-axis inverse scale is src / dst; fx = float32((dx + 0.5) * (w / nw) - 0.5), sx = floor(fx), fx -= sx, border
    taps clamped with zero weight; weights as int round(w * 2048); horizontal pass in int32, vertical pass
    (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2."""
    h, w = image.shape[:2]

    def taps(n_out, n_in):
        inv = n_in / n_out
        f = ((np.arange(n_out, dtype=np.float64) + 0.5) * inv - 0.5).astype(np.float32)
        i = np.floor(f).astype(np.int64)
        f = f - i.astype(np.float32)
        lo, hi = i < 0, i >= n_in - 1
        f = np.where(lo | hi, np.float32(0), f)
        i = np.where(lo, 0, np.where(hi, n_in - 1, i))
        w1 = np.rint(f * np.float32(2048)).astype(np.int64)
        w0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
        return i, np.minimum(i + 1, n_in - 1), w0, w1

    x0, x1, a0, a1 = taps(nw, w)
    y0, y1, b0, b1 = taps(nh, h)
    src = image.astype(np.int64)
    S = src[:, x0, :] * a0[None, :, None] + src[:, x1, :] * a1[None, :, None]            # [h, nw, 3]
    out = (((b0[:, None, None] * (S[y0] >> 4)) >> 16) + ((b1[:, None, None] * (S[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def preprocess_image(image: np.ndarray, image_size: int):
    """generators/colibri_common.py:622-656.  Frames with max(H, W) == image_size need no resize (scale 1.0: cv2.resize
    to the same size returns the image unchanged; that branch is pinned against the imported reference function,
    tests/golden/preprocess.npz); others go through ``resize_bilinear_u8`` (parity unpinned).  The arithmetic lines are
    the reference's own: uint8 HWC RGB -> float32, /255., -mean, /std (numpy evaluates the two list operands in float64
    and rounds the in-place result to float32), zero-pad bottom/right.  Returns (image [S,S,3] float32, scale)."""
    image_height, image_width = image.shape[:2]
    if image_height > image_width:
        scale = image_size / image_height
        resized_height, resized_width = image_size, int(image_width * scale)
    else:
        scale = image_size / image_width
        resized_height, resized_width = int(image_height * scale), image_size
    if (resized_height, resized_width) != (image_height, image_width):
        image = resize_bilinear_u8(image, resized_width, resized_height)
        image_height, image_width = resized_height, resized_width
    image = image.astype(np.float32)
    image /= 255.
    mean = [0.485, 0.456, 0.406]
    std = [0.229, 0.224, 0.225]
    image -= mean
    image /= std
    pad_h = image_size - image_height
    pad_w = image_size - image_width
    image = np.pad(image, [(0, pad_h), (0, pad_w), (0, 0)], mode='constant')
    return image, scale


# ---- the WebRTC frame path of the reference app (unity-sandbox/WebRTCNetCoreSandbox/Program.cs:128-205, 381-445) ----
_CY, _CUB, _CUG, _CVG, _CVR, _YUV_SHIFT = 1220542, 2116026, -409993, -852492, 1673527, 20


def yv12_to_bgr(buf: np.ndarray, height: int, width: int) -> np.ndarray:
    """cv2.cvtColor(mat[(rows * 3 / 2), cols], COLOR_YUV2BGR_YV12) (Program.cs:161), restated from OpenCV's ITU-R BT.601
    fixed-point conversion (imgproc/color_yuv: y = max(0, Y - 16) * 1220542; r = (y + 1673527 * (V - 128) + 2^19) >> 20, g =
    (y - 852492 * (V - 128) - 409993 * (U - 128) + 2^19) >> 20, b = (y + 2116026 * (U - 128) + 2^19) >> 20, saturated) -
    PARITY UNPINNED: cv2 is not in this image.  YV12 plane order is Y, V, U; the app hands this call the I420 bytes of
    the frame (Y, U, V), so its colours have U and V exchanged - kept, it is what the deployed model sees."""
    buf = np.asarray(buf, np.uint8).ravel()
    y = buf[: height * width].reshape(height, width).astype(np.int64)
    q = (height // 2) * (width // 2)
    v = buf[height * width: height * width + q].reshape(height // 2, width // 2).astype(np.int64)
    u = buf[height * width + q: height * width + 2 * q].reshape(height // 2, width // 2).astype(np.int64)
    uu = np.repeat(np.repeat(u, 2, 0), 2, 1)[:height, :width] - 128
    vv = np.repeat(np.repeat(v, 2, 0), 2, 1)[:height, :width] - 128
    yy = np.maximum(0, y - 16) * _CY
    half = 1 << (_YUV_SHIFT - 1)
    r = (yy + half + _CVR * vv) >> _YUV_SHIFT
    g = (yy + half + _CVG * vv + _CUG * uu) >> _YUV_SHIFT
    b = (yy + half + _CUB * uu) >> _YUV_SHIFT
    return np.clip(np.stack([b, g, r], -1), 0, 255).astype(np.uint8)


def webrtc_frame_preprocess(buf: np.ndarray, height: int, width: int, image_size: int, crop: int = 256, resized: int = 512):
    """The frame callback of the reference's streaming app, Program.cs:140-205: YUV2BGR_YV12 -> CenterCropAndRescaleMat
    (crop x crop at ((cols - crop) / 2, (rows - crop) / 2), cv2.resize to resized x resized) -> ResizeAndNormalizeMat
    (cv2.resize to image_size on the longer side, float32 / 255, - (0.485, 0.456, 0.406), / (0.229, 0.224, 0.225) applied to
    the B, G, R channels in that order - the C# Scalars meet a BGR Mat -, all in float32 as OpenCV does for CV_32F,
    zero-pad) -> blobFromImage(swapRB = false): returns ([S, S, 3] float32 in B, G, R order, scale)."""
    bgr = yv12_to_bgr(buf, height, width)
    ow, oh = (width - crop) // 2, (height - crop) // 2
    roi = np.ascontiguousarray(bgr[oh: oh + crop, ow: ow + crop])
    img = resize_bilinear_u8(roi, resized, resized) if resized != crop else roi
    scale = np.float32(image_size) / np.float32(resized)            # (float)img_size / image_width
    nw = image_size                                                   # square input: both sides become image_size
    nh = int(np.float32(resized) * scale)
    if (nh, nw) != (resized, resized):
        img = resize_bilinear_u8(img, nw, nh)
    x = img.astype(np.float32) / np.float32(255.0)
    x = (x - np.array([0.485, 0.456, 0.406], np.float32)) / np.array([0.229, 0.224, 0.225], np.float32)
    out = np.zeros((image_size, image_size, 3), np.float32)
    out[:nh, :nw] = x.astype(np.float32)
    return out, float(scale)
