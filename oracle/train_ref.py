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
