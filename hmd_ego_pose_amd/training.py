"""Training-side ops of the reference on the GPU.

``anchor_targets`` is ``anchor_targets_bbox`` (pytorch-sandbox/generators/utils/anchors.py:69-221): the reference builds
the per-batch classification / regression / transformation / hand targets on the host with numpy and a Cython IoU matrix
(generators/utils/compute_overlap.pyx:33-73); here the assignment runs in one kernel (csrc/k_eval.hip,
hep_anchor_targets_device) and its outputs stay on the device for the losses.  The losses themselves
(hmdegopose/loss.py:54-428) are plain torch code in the reference and keep running through torch autograd - training
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
