"""Training-side ops of the reference on the GPU.

``anchor_targets`` is ``anchor_targets_bbox`` (pytorch-sandbox/generators/utils/anchors.py:69-221): the reference builds
the per-batch classification / regression / transformation / hand targets on the host with numpy and a Cython IoU matrix
(generators/utils/compute_overlap.pyx:33-73); here the assignment runs in one kernel (csrc/k_eval.hip,
hep_anchor_targets_device) and its outputs stay on the device for the losses.

``losses`` is ``batch_iterate`` (pytorch-sandbox/hmdegopose/loss.py:54-99): the forward VALUES of the focal, box, rotation
(model-point distance), translation and hand losses, one workgroup per image (hep_losses_device).  The reference computes
them with a Python loop over the batch and per-image gathers; gradients stay with the caller's autograd - training
through the HIP forward is out of scope (the inference path has no backward).
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _capi


def anchor_targets(anchors: torch.Tensor, boxes: Sequence[np.ndarray], labels: Sequence[np.ndarray],
                   transformation_targets: Sequence[np.ndarray], coords_3d: Optional[Sequence[np.ndarray]],
                   image_shapes: Sequence[Tuple[int, int]], num_classes: int = 1, negative_overlap: float = 0.4,
                   positive_overlap: float = 0.5):
    """anchors: float32 [N,4] on the device.  Per image: boxes [K,4] (x1,y1,x2,y2, any float dtype: used as float64 like
    the reference), labels [K], transformation targets [K,RT], coords_3d [K,63] (or None), image shape (height, width).
    Returns device tensors (labels [B,N,C+1], regression [B,N,5], transformation [B,N,RT+1], coords [B,N,64] or None);
    the last column of each is the anchor state (-1 ignore, 0 background, 1 object)."""
    if not anchors.is_cuda or anchors.dtype != torch.float32 or anchors.dim() != 2 or anchors.shape[1] != 4:
        raise ValueError("anchors must be a float32 ROCm tensor [N,4]")
    dev, B, N = anchors.device, len(boxes), anchors.shape[0]
    kmax = max(1, max(int(b.shape[0]) for b in boxes))
    rt = int(transformation_targets[0].shape[1]) if len(transformation_targets) else 0
    gb = np.zeros((B, kmax, 4), np.float64); gl = np.zeros((B, kmax), np.int32); gt = np.zeros((B, kmax, rt), np.float32)
    gc = np.zeros((B, kmax, 63), np.float32) if coords_3d is not None else None
    ng = np.zeros((B,), np.int32); hw = np.zeros((B, 2), np.int32)
    for i in range(B):
        k = int(boxes[i].shape[0])
        ng[i] = k; hw[i] = (int(image_shapes[i][0]), int(image_shapes[i][1]))
        if k:
            gb[i, :k] = boxes[i]; gl[i, :k] = np.asarray(labels[i]).astype(np.int32); gt[i, :k] = transformation_targets[i]
            if gc is not None:
                gc[i, :k] = np.asarray(coords_3d[i]).reshape(k, 63)
    t = lambda a: torch.from_numpy(a).to(dev)
    d_gb, d_gl, d_gt, d_ng, d_hw = t(gb), t(gl), t(gt), t(ng), t(hw)
    d_gc = t(gc) if gc is not None else None
    f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    lab, reg, tra = f(B, N, num_classes + 1), f(B, N, 5), f(B, N, rt + 1)
    crd = f(B, N, 64) if gc is not None else None
    stream = torch.cuda.current_stream(dev).cuda_stream
    a = anchors.contiguous()
    _capi.check(_capi.lib().hep_anchor_targets_device(a.data_ptr(), N, d_gb.data_ptr(), d_gl.data_ptr(), d_gt.data_ptr(), _capi.ptr(d_gc),
                                                      d_ng.data_ptr(), d_hw.data_ptr(), B, kmax, num_classes, rt, float(negative_overlap),
                                                      float(positive_overlap), lab.data_ptr(), reg.data_ptr(), tra.data_ptr(), _capi.ptr(crd), stream))
    torch.cuda.current_stream(dev).synchronize()      # the staging tensors above must outlive the launch
    return lab, reg, tra, crd


def losses(gt_classification: torch.Tensor, classification: torch.Tensor, gt_regression: torch.Tensor, regression: torch.Tensor,
           gt_transformation: torch.Tensor, transformation: torch.Tensor, gt_hand: Optional[torch.Tensor], hand: Optional[torch.Tensor],
           model_3d_points, num_rotation_parameter: int = 3):
    """``batch_iterate`` (hmdegopose/loss.py:54-99) on float32 ROCm tensors laid out as the generator / the network
    produce them (see include/hep.h: hep_losses_device); ``model_3d_points`` [classes, P, 3] (numpy or tensor).  Returns
    (losses [5] = classification, regression x 50, rotation, translation, hand - the batch means, per_image [B, 5])."""
    dev = classification.device
    def chk(x, name):
        if x is None:
            return None
        if not x.is_cuda or x.dtype != torch.float32:
            raise ValueError(f"{name} must be a float32 ROCm tensor")
        return x.contiguous()
    gc, pc, gr, pr = chk(gt_classification, "gt_classification"), chk(classification, "classification"), chk(gt_regression, "gt_regression"), chk(regression, "regression")
    gt, pt, gh, ph = chk(gt_transformation, "gt_transformation"), chk(transformation, "transformation"), chk(gt_hand, "gt_hand"), chk(hand, "hand")
    B, N, K = pc.shape
    R = int(num_rotation_parameter)
    if gc.shape != (B, N, K + 1) or gr.shape != (B, N, 5) or pr.shape != (B, N, 4) or gt.shape != (B, N, R + 6) or pt.shape != (B, N, R + 3):
        raise ValueError("loss inputs do not have the reference's shapes")
    H = 0
    if ph is not None:
        H = int(ph.shape[2])
        if gh is None or gh.shape != (B, N, H + 1) or ph.shape != (B, N, H):
            raise ValueError("gt_hand must be [B, N, H + 1] next to hand [B, N, H]")
    pts = torch.as_tensor(np.asarray(model_3d_points, dtype=np.float32) if not torch.is_tensor(model_3d_points) else model_3d_points,
                          dtype=torch.float32).to(dev).contiguous()
    if pts.dim() != 3 or pts.shape[2] != 3:
        raise ValueError("model_3d_points must be [classes, P, 3]")
    per = torch.empty((B, 5), dtype=torch.float32, device=dev)
    out = torch.empty((5,), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    _capi.check(_capi.lib().hep_losses_device(gc.data_ptr(), pc.data_ptr(), gr.data_ptr(), pr.data_ptr(), gt.data_ptr(), pt.data_ptr(),
                                              _capi.ptr(gh), _capi.ptr(ph), pts.data_ptr(), B, N, K, R, H, int(pts.shape[0]), int(pts.shape[1]),
                                              per.data_ptr(), out.data_ptr(), stream))
    torch.cuda.current_stream(dev).synchronize()      # the contiguous copies above must outlive the launch
    return out, per
