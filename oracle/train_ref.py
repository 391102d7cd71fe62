"""ORACLE (test infrastructure, never on the product path): numpy restatement of the training-side anchor-target
assignment of the reference - the IoU matrix of ``generators/utils/compute_overlap.pyx:33-73``, the positive / ignore /
negative split of ``compute_gt_annotations`` (``generators/utils/anchors.py:185-221``), ``bbox_transform``
(``anchors.py:422-458``) and the batch assembly of ``anchor_targets_bbox`` (``anchors.py:69-182``).

PINNED: ``tests/golden/make_golden_targets.py`` compiles the reference's own Cython extension out of tree, imports the
reference's ``anchors.py`` with it and stores digests of what ``anchor_targets_bbox`` returns for seeded boxes;
``tests/test_decode_oracle_cpu.py`` replays them bit for bit.
"""
from __future__ import annotations

import numpy as np


def compute_overlap(boxes: np.ndarray, query_boxes: np.ndarray) -> np.ndarray:
    """IoU matrix [N,K] in float64 with the "+1" pixel convention (compute_overlap.pyx:33-73): an entry is non-zero only
    when both iw > 0 and ih > 0."""
    b = np.asarray(boxes, dtype=np.float64)[:, None, :]
    q = np.asarray(query_boxes, dtype=np.float64)[None, :, :]
    box_area = (q[..., 2] - q[..., 0] + 1) * (q[..., 3] - q[..., 1] + 1)
    iw = np.minimum(b[..., 2], q[..., 2]) - np.maximum(b[..., 0], q[..., 0]) + 1
    ih = np.minimum(b[..., 3], q[..., 3]) - np.maximum(b[..., 1], q[..., 1]) + 1
    ua = (b[..., 2] - b[..., 0] + 1) * (b[..., 3] - b[..., 1] + 1) + box_area - iw * ih
    with np.errstate(divide="ignore", invalid="ignore"):
        ov = iw * ih / ua
    return np.where((iw > 0) & (ih > 0), ov, 0.0)


def compute_gt_annotations(anchors, annotations, negative_overlap=0.4, positive_overlap=0.5):
    """anchors.py:185-221: per anchor the gt box of greatest overlap (lowest index on ties); positive when that overlap
    >= positive_overlap, and the best anchor of EVERY gt box (lowest index on ties - anchor 0 for a box nothing overlaps)
    is forced positive; ignore when the overlap > negative_overlap and not positive."""
    overlaps = compute_overlap(anchors.astype(np.float64), annotations.astype(np.float64))
    argmax = np.argmax(overlaps, axis=1)
    mx = overlaps[np.arange(overlaps.shape[0]), argmax]
    positive = mx >= positive_overlap
    positive[np.argmax(overlaps, axis=0)] = True
    ignore = (mx > negative_overlap) & ~positive
    return positive, ignore, argmax


def bbox_transform(anchors, gt_boxes):
    """anchors.py:422-458 (no scale factors): targets (ty, tx, th, tw); the anchors keep their dtype (float32 from
    anchors_for_shape: `wa += 1e-7` is a float32 add), the boxes theirs (float64), mixed expressions promote to float64."""
    wa = anchors[:, 2] - anchors[:, 0]
    ha = anchors[:, 3] - anchors[:, 1]
    cxa = anchors[:, 0] + wa / 2.
    cya = anchors[:, 1] + ha / 2.
    w = gt_boxes[:, 2] - gt_boxes[:, 0]
    h = gt_boxes[:, 3] - gt_boxes[:, 1]
    cx = gt_boxes[:, 0] + w / 2.
    cy = gt_boxes[:, 1] + h / 2.
    ha = ha + np.asarray(1e-7, dtype=ha.dtype)
    wa = wa + np.asarray(1e-7, dtype=wa.dtype)
    h = h + 1e-7
    w = w + 1e-7
    return np.stack([(cy - cya) / ha, (cx - cxa) / wa, np.log(h / ha), np.log(w / wa)], axis=1)


def anchor_targets(anchors, image_shapes, boxes, labels, transformation_targets, coords_3d, num_classes, negative_overlap=0.4,
                   positive_overlap=0.5):
    """anchor_targets_bbox, anchors.py:69-182, for a batch given as lists (one entry per image: boxes [K,4] float64,
    labels [K], transformation targets [K,RT], coords [K,63]); returns float32 (labels [B,N,C+1], regression [B,N,5],
    transformation [B,N,RT+1], coords [B,N,64]); the last column is the anchor state (-1 ignore, 0 background, 1 object)."""
    B, N = len(boxes), anchors.shape[0]
    rt = transformation_targets[0].shape[1]
    lab = np.zeros((B, N, num_classes + 1), np.float32)
    reg = np.zeros((B, N, 5), np.float32)
    tra = np.zeros((B, N, rt + 1), np.float32)
    crd = np.zeros((B, N, 64), np.float32)
    for i in range(B):
        if boxes[i].shape[0]:
            pos, ign, arg = compute_gt_annotations(anchors, boxes[i], negative_overlap, positive_overlap)
            for out in (lab, reg, tra, crd):
                out[i, ign, -1] = -1
                out[i, pos, -1] = 1
            lab[i, pos, labels[i][arg[pos]].astype(int)] = 1
            reg[i, :, :4] = bbox_transform(anchors, boxes[i][arg, :])
            tra[i, :, :-1] = transformation_targets[i][arg, :]
            crd[i, :, :-1] = coords_3d[i][arg, :]
        cx = (anchors[:, 0] + anchors[:, 2]) / 2
        cy = (anchors[:, 1] + anchors[:, 3]) / 2
        outside = np.logical_or(cx >= image_shapes[i][1], cy >= image_shapes[i][0])
        for out in (lab, reg, tra, crd):
            out[i, outside, -1] = -1
    return lab, reg, tra, crd


# ------------------------------------------------------------------------------------------------------------------
# Losses (hmdegopose/loss.py:54-428).  PINNED: tests/golden/make_golden_losses.py imports the reference's loss.py
# unchanged (torchvision / Cython stubs as for the network), runs ``batch_iterate`` on seeded predictions and targets
# and stores the five returned scalars per case; tests/test_decode_oracle_cpu.py replays them (float32 arithmetic in a
# different summation order: relative 1e-5).
# ------------------------------------------------------------------------------------------------------------------
def _smooth_l1_sigma(diff: np.ndarray, sigma: float = 3.0) -> np.ndarray:
    """f(x) = 0.5 (sigma x)^2 if |x| <= 1/sigma^2 else |x| - 0.5/sigma^2 (loss.py:205-209: note the `le`)."""
    s2 = np.float32(sigma * sigma)
    d = np.abs(diff.astype(np.float32))
    return np.where(d <= np.float32(1.0) / s2, np.float32(0.5) * s2 * d * d, d - np.float32(0.5) / s2)


def focal_loss(gt_classification: np.ndarray, classification: np.ndarray, alpha: float = 0.25, gamma: float = 1.5) -> np.float32:
    """loss.py:102-167, one image: gt [N, K+1] (last column the anchor state), predictions [N, K] (post-sigmoid)."""
    gt = gt_classification.astype(np.float32)
    labels, state = gt[:, :-1], gt[:, -1]
    keep = state != -1
    labels = labels[keep]
    p = np.clip(classification.astype(np.float32), np.float32(1e-4), np.float32(1.0 - 1e-4))[keep]
    a = np.where(labels == 1, np.float32(alpha), np.float32(1.0 - alpha)).astype(np.float32)
    fw = a * np.power(np.where(labels == 1, np.float32(1) - p, p), np.float32(gamma))
    bce = -(labels * np.log(p) + (np.float32(1) - labels) * np.log(np.float32(1) - p))
    cls = np.where(labels != -1, fw * bce, np.float32(0))
    return np.float32(cls.sum(dtype=np.float32) / np.float32(max(1.0, float((state == 1).sum()))))


def smooth_l1_loss(gt: np.ndarray, pred: np.ndarray, sigma: float = 3.0) -> np.float32:
    """loss.py:170-219 (boxes) and 222-271 (hands), one image: gt [N, D+1] (last column the anchor state), pred [N, D]."""
    gt = gt.astype(np.float32)
    pos = gt[:, -1] == 1
    loss = _smooth_l1_sigma(pred.astype(np.float32)[pos] - gt[pos, :-1], sigma)
    return np.float32(loss.sum(dtype=np.float32) / np.float32(max(1.0, float(pos.sum()))))


def _rotate(points: np.ndarray, rvec: np.ndarray) -> np.ndarray:
    """loss.py:436-458 + 570-609: axis = r / |r| (plain division: a zero vector gives NaN as in the reference), Rodrigues."""
    r = rvec.astype(np.float32)
    angle = np.sqrt((r * r).sum(dtype=np.float32))
    with np.errstate(divide="ignore", invalid="ignore"):
        axis = r / angle
    c, s = np.cos(angle, dtype=np.float32), np.sin(angle, dtype=np.float32)
    p = points.astype(np.float32)
    return p * c + np.cross(axis[None, :], p).astype(np.float32) * s + axis[None, :] * (p @ axis)[:, None] * (np.float32(1) - c)


def transformation_loss(gt_transformation: np.ndarray, transformation: np.ndarray, model_points: np.ndarray,
                        num_rotation_parameter: int = 3):
    """loss.py:273-428, one image: gt [N, R+3+3] = (rotation, translation, is_symmetric, class, state), pred [N, R+3],
    model points [classes, P, 3].  Returns (rotation loss = mean over the positive anchors of the mean (nearest, when
    symmetric) point distance between the two rotated models; NaN -> 0, translation loss = torch SmoothL1Loss (beta 1,
    mean over positives x 3; NaN without positives, as the reference returns it))."""
    R = num_rotation_parameter
    gt = gt_transformation.astype(np.float32)
    pr = transformation.astype(np.float32)
    pos = np.round(gt[:, -1]).astype(np.int32) == 1
    idx = np.nonzero(pos)[0]
    dists = []
    for i in idx:
        pts = model_points[int(np.round(gt[i, -2]))].astype(np.float32)
        a = _rotate(pts, pr[i, :R] * np.float32(np.pi))
        b = _rotate(pts, gt[i, :R] * np.float32(np.pi))
        if int(np.round(gt[i, -3])) == 1:
            d = np.sqrt(((a[:, None, :] - b[None, :, :]) ** 2).sum(-1, dtype=np.float32)).min(axis=1)
        else:
            d = np.sqrt(((a - b) ** 2).sum(-1, dtype=np.float32))
        dists.append(d.mean(dtype=np.float32))
    rot = np.float32(np.mean(np.array(dists, dtype=np.float32))) if dists else np.float32(0)
    if np.isnan(rot):
        rot = np.float32(0)
    d = np.abs(pr[idx, R:] - gt[idx, R:-3])
    tl = np.where(d < 1, np.float32(0.5) * d * d, d - np.float32(0.5))
    trans = np.float32(tl.mean(dtype=np.float32)) if tl.size else np.float32(np.nan)
    return rot, trans


def batch_losses(gt_classification, classification, gt_regression, regression, gt_transformation, transformation,
                 gt_hand, hand, model_points, num_rotation_parameter: int = 3) -> np.ndarray:
    """``batch_iterate`` (loss.py:54-99): the five losses, each the mean over the batch; the box regression x 50."""
    B = classification.shape[0]
    per = np.zeros((B, 5), np.float32)
    for j in range(B):
        per[j, 0] = focal_loss(gt_classification[j], classification[j])
        per[j, 1] = smooth_l1_loss(gt_regression[j], regression[j])
        per[j, 2], per[j, 3] = transformation_loss(gt_transformation[j], transformation[j], model_points, num_rotation_parameter)
        per[j, 4] = smooth_l1_loss(gt_hand[j], hand[j])
    out = per.mean(axis=0, dtype=np.float32)
    out[1] *= np.float32(50)
    return out
