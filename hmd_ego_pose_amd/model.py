"""Host-side mirror of the reference's operator interface for the inference path.

``HMDEgoPose`` has the reference's constructor, ``forward`` signature and ``state_dict``
keys (pytorch-sandbox/backbone.py:13-16,104-125; key inventory in arch.param_spec), so
``evaluate.py`` / ``main.py``-style drivers can build it, ``load_state_dict`` a checkpoint
and call it.  In ``eval()`` on a ROCm device its forward is one call into ``libhep.so``
(hand-written gfx950 kernels behind a C ABI, csrc/) through the ``torch.ops.hep.*`` custom
ops below.  There is no other execution path: CPU tensors, training mode or a missing
library raise - the reference's training graph is out of scope for this build.

``TrainModelWithLoss`` mirrors the inference branch of pytorch-sandbox/train.py:23-85
(forward -> anchors -> translation/box decode -> detection filter) on the GPU.
"""
from __future__ import annotations

import ctypes
import os
import threading
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import nn

from . import _capi
from .arch import OUT_WIDTH, get_arch, level_sizes, num_anchors_total, param_spec
from .weights import pack_bytes

_PRECISIONS = {"fp32": _capi.HEP_F32, "f32": _capi.HEP_F32, "bf16": _capi.HEP_BF16, "fp8": _capi.HEP_FP8}


# --------------------------------------------------------------------------------------
# session: one libhep handle for (weights, phi, size, max_batch, dtype, device)
# --------------------------------------------------------------------------------------
class Session:
    """Owns a ``hep_handle``.  All tensors handed to it must live on ``device``."""

    _registry: Dict[int, "Session"] = {}
    _lock = threading.Lock()

    def __init__(self, state_dict, phi: int, size: int, max_batch: int, precision: str = "fp32",
                 device: Optional[torch.device] = None, flags: int = 0):
        if precision not in _PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_PRECISIONS)}")
        if not torch.cuda.is_available():
            raise _capi.HepError("no ROCm device visible: the MI355X path has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.phi, self.size, self.max_batch, self.precision = phi, size, max_batch, precision
        blob = pack_bytes(state_dict)
        h = ctypes.c_void_p()
        _capi.check(_capi.lib().hep_create_from_memory(blob, len(blob), phi, size, max_batch, _PRECISIONS[precision],
                                                       self.device.index or 0, flags, ctypes.byref(h)))
        self.handle = h.value
        self._class_specific = True      # the handle's filter mode (hep_set_class_specific_filter)
        self._filter_lock = threading.Lock()      # mode switch + filter call are one step for concurrent callers of one session
        self.num_anchors = _capi.lib().hep_num_anchors(self.handle)
        self.num_classes = _capi.lib().hep_num_classes(self.handle)      # read from the classifier header of the weights
        self.out_width = tuple(self.num_classes if i == 1 else k for i, k in enumerate(OUT_WIDTH))
        self.lane_batch = max_batch      # frames per launch (the batch is one lane unless HEP_LANES says otherwise)
        if os.environ.get("HEP_LANES"):
            lanes = max(1, min(int(os.environ["HEP_LANES"]), max_batch, 16))
            self.lane_batch = -(-max_batch // lanes)
        self.fpn_w = get_arch(phi).fpn_w
        self.levels = level_sizes(size)
        with Session._lock:
            Session._registry[self.handle] = self

    def close(self):
        if getattr(self, "handle", None):
            with Session._lock:
                Session._registry.pop(self.handle, None)
            _capi.lib().hep_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- forward ----------------------------------------------------------------------
    def forward(self, x: torch.Tensor, want_features: bool = True):
        """x: fp32 [B,3,S,S] on the session's device, any strides.  Returns
        (features|None, regression, classification, rotation, translation_raw, hand)."""
        if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != self.size or x.shape[3] != self.size:
            raise ValueError(f"expected a float32 ROCm tensor [B,3,{self.size},{self.size}], got {tuple(x.shape)} {x.dtype} on {x.device}")
        B = x.shape[0]
        N = self.num_anchors
        outs = [torch.empty((B, N, k), dtype=torch.float32, device=x.device) for k in self.out_width]
        feats = [torch.empty((B, self.fpn_w, s, s), dtype=torch.float32, device=x.device) for s in self.levels] if want_features else None
        strides = (ctypes.c_int64 * 4)(*x.stride())
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _capi.check(_capi.lib().hep_run_device(self.handle, x.data_ptr(), strides, B, _capi.ptr_array(outs),
                                               _capi.ptr_array(feats) if feats else None, stream))
        return (tuple(feats) if feats else None, *outs)

    def output_views(self) -> List[torch.Tensor]:
        """The handle's own head-output buffers as tensors [max_batch, N, K] (regression, classification, rotation,
        translation_raw, hand): what ``hep_run_device`` fills when it is given no output pointers.  Views, not copies: the
        next forward on this session rewrites them, and they die with the session."""
        class _Buf:       # the CUDA array interface torch.as_tensor understands (also on ROCm)
            def __init__(self, ptr, shape):
                self.__cuda_array_interface__ = {"shape": shape, "typestr": "<f4", "data": (ptr, False), "version": 2}
        views = []
        for i, k in enumerate(self.out_width):
            p = ctypes.c_void_p()
            _capi.check(_capi.lib().hep_output_device(self.handle, 5 + i, ctypes.byref(p)))
            views.append(torch.as_tensor(_Buf(p.value, (self.max_batch, self.num_anchors, k)), device=self.device))
        return views

    def preprocess(self, images_u8: torch.Tensor) -> torch.Tensor:
        """uint8 RGB [B,H,W,3] on the device -> the NCHW view of the normalised, zero-padded float32 [B,size,size,3]
        (generators/colibri_common.py:622-656; the view is what eval/common.py:397 feeds the model and ``forward``
        reads in place).  Frames whose longer side is not ``size`` are resized on the GPU (8-bit bilinear, OpenCV's
        convention restated - parity unpinned); the camera vector's image_scale is then size / max(H, W)."""
        if not images_u8.is_cuda or images_u8.dtype != torch.uint8 or images_u8.dim() != 4 or images_u8.shape[3] != 3:
            raise ValueError("expected a uint8 ROCm tensor [B,H,W,3]")
        x = images_u8.contiguous()
        B, H, W = x.shape[0], x.shape[1], x.shape[2]
        out = torch.empty((B, self.size, self.size, 3), dtype=torch.float32, device=x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _capi.check(_capi.lib().hep_preprocess_u8_device(self.handle, x.data_ptr(), B, H, W, out.data_ptr(), stream))
        return out.permute(0, 3, 1, 2)

    def preprocess_i420(self, frames_u8: torch.Tensor, height: int, width: int, crop: int = 256, resized: int = 512) -> torch.Tensor:
        """4:2:0 planar frames [B, height * width * 3 // 2] (uint8, on the device) as the reference's streaming app receives
        them -> the NCHW view (B, G, R channel order) of what its frame callback feeds the ONNX session
        (unity-sandbox/WebRTCNetCoreSandbox/Program.cs:140-205: YUV2BGR_YV12, centre crop, resize, normalise; parity
        unpinned against OpenCV itself)."""
        if not frames_u8.is_cuda or frames_u8.dtype != torch.uint8 or frames_u8.dim() != 2 or frames_u8.shape[1] != height * width * 3 // 2:
            raise ValueError("expected a uint8 ROCm tensor [B, height * width * 3 // 2]")
        x = frames_u8.contiguous()
        out = torch.empty((x.shape[0], self.size, self.size, 3), dtype=torch.float32, device=x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _capi.check(_capi.lib().hep_preprocess_i420_device(self.handle, x.data_ptr(), x.shape[0], height, width, crop, resized, out.data_ptr(), stream))
        return out.permute(0, 3, 1, 2)

    def decode(self, regression, translation_raw, camera):
        B = regression.shape[0]
        boxes = torch.empty((B, self.num_anchors, 4), dtype=torch.float32, device=regression.device)
        trans = torch.empty((B, self.num_anchors, 3), dtype=torch.float32, device=regression.device)
        cam = camera.to(regression.device, torch.float32).contiguous()
        stream = torch.cuda.current_stream(regression.device).cuda_stream
        _capi.check(_capi.lib().hep_decode_device(self.handle, regression.contiguous().data_ptr(), translation_raw.contiguous().data_ptr(),
                                                  cam.data_ptr(), B, boxes.data_ptr(), trans.data_ptr(), stream))
        return boxes, trans

    def filter(self, boxes, classification, rotation, translation, hand, score_threshold=0.5, nms_threshold=0.5, max_detections=100,
               class_specific_filter=True):
        """filter_detections (hmdegopose/layers.py:264-400) for every image of the batch; class_specific_filter=False takes every
        anchor's best class and filters those in one pass (layers.py:359-362)."""
        B, dev, M = boxes.shape[0], boxes.device, max_detections
        # the C ABI takes bare pointers and indexes them as [B][N][K] with K from the weights: a tensor of another width would be read
        # out of bounds on the device, so the shapes are checked here
        N = self.num_anchors
        for name, t, k in (("boxes", boxes, 4), ("classification", classification, self.num_classes), ("rotation", rotation, 3),
                           ("translation", translation, 3), ("hand", hand, 63)):
            if tuple(t.shape) != (B, N, k) or t.dtype != torch.float32 or t.device != dev:
                raise ValueError(f"filter: {name} must be a float32 tensor of shape ({B}, {N}, {k}) on {dev}, got {t.dtype} {tuple(t.shape)} on {t.device}")
        if not (B <= self.max_batch and 1 <= M <= 256):
            raise ValueError(f"filter: batch {B} exceeds the session's {self.max_batch} or max_detections {M} is outside 1..256")
        f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        i = lambda *s: torch.empty(s, dtype=torch.int32, device=dev)
        out = dict(boxes=f(B, M, 4), scores=f(B, M), labels=i(B, M), rotation=f(B, M, 3), translation=f(B, M, 3),
                   hand=f(B, M, 63), index=i(B, M), count=i(B))
        stream = torch.cuda.current_stream(dev).cuda_stream
        args = [t.contiguous() for t in (boxes, classification, rotation, translation, hand)]
        with self._filter_lock:
            if bool(class_specific_filter) != self._class_specific:
                _capi.check(_capi.lib().hep_set_class_specific_filter(self.handle, int(bool(class_specific_filter))))
                self._class_specific = bool(class_specific_filter)
            _capi.check(_capi.lib().hep_filter_device(self.handle, *[t.data_ptr() for t in args], B, float(score_threshold),
                                                      float(nms_threshold), M, *[out[k].data_ptr() for k in
                                                                                 ("boxes", "scores", "labels", "rotation", "translation", "hand", "index", "count")],
                                                      stream))
        return out

    # -- introspection ------------------------------------------------------------------
    def stage(self, name: str, batch: int) -> torch.Tensor:
        """fp32 NHWC copy of a stage tensor of the last forward (needs FLAG_KEEP_INTERMEDIATES)."""
        l = _capi.lib()
        n = l.hep_debug_tensor_count(self.handle)
        for i in range(n):
            nm = ctypes.c_char_p(); dims = (ctypes.c_int64 * 4)()
            l.hep_debug_tensor_info(self.handle, i, ctypes.byref(nm), dims)
            if nm.value.decode() == name:
                out = torch.empty((batch, dims[1], dims[2], dims[3]), dtype=torch.float32)
                _capi.check(l.hep_debug_tensor(self.handle, name.encode(), batch, out.data_ptr(), out.numel()))
                return out
        raise KeyError(name)

    def kernels(self, batch: int) -> List[Tuple[str, float, float, str]]:
        """(launch name, algorithmic bytes, flops, device function) of every launch in one forward."""
        l = _capi.lib()
        out = []
        for i in range(l.hep_kernel_count(self.handle, batch)):
            nm = ctypes.c_char_p(); b = ctypes.c_double(); f = ctypes.c_double()
            _capi.check(l.hep_kernel_info(self.handle, batch, i, ctypes.byref(nm), ctypes.byref(b), ctypes.byref(f)))
            sym = ctypes.c_char_p()
            _capi.check(l.hep_kernel_symbol(self.handle, i, ctypes.byref(sym)))
            out.append((nm.value.decode(), b.value, f.value, sym.value.decode()))
        return out

    def fp8_scales(self) -> Dict[str, float]:
        """launch name -> calibrated per-tensor activation scale of its e4m3 GEMM (fp8 sessions; empty otherwise)."""
        l = _capi.lib()
        out = {}
        for i, (name, *_rest) in enumerate(self.kernels(1)):
            sc = ctypes.c_float()
            _capi.check(l.hep_fp8_scale(self.handle, i, ctypes.byref(sc)))
            if sc.value > 0:
                out[name] = sc.value
        return out

    def calibrate_fp8(self, frames: torch.Tensor):
        """fp8 sessions: recompute the per-tensor activation scales on representative frames (fp32 [B,3,S,S] on the device).
        The scales fixed at creation come from synthetic noise with 2x headroom and the e4m3 conversion saturates silently."""
        x = frames.contiguous()
        if not x.is_cuda or x.dtype != torch.float32 or tuple(x.shape[1:]) != (3, self.size, self.size):
            raise ValueError(f"expected a float32 ROCm tensor [B,3,{self.size},{self.size}]")
        torch.cuda.synchronize(x.device)
        _capi.check(_capi.lib().hep_calibrate_fp8(self.handle, x.data_ptr(), x.shape[0]))

    def profile(self, batch: int, iters: int = 20, per_kernel: bool = False):
        l = _capi.lib()
        total = ctypes.c_float()
        n = l.hep_kernel_count(self.handle, batch)
        per = (ctypes.c_float * n)() if per_kernel else None
        _capi.check(l.hep_profile(self.handle, batch, iters, ctypes.byref(total), per))
        return total.value, (list(per) if per_kernel else None)

    def profile_concurrent(self, batch: int, iters: int = 20, nstreams: int = 4):
        """ms per launch of every launch of the plan when it is issued on ``nstreams`` streams at once."""
        l = _capi.lib()
        per = (ctypes.c_float * l.hep_kernel_count(self.handle, batch))()
        _capi.check(l.hep_profile_concurrent(self.handle, batch, iters, nstreams, per))
        return list(per)


# --------------------------------------------------------------------------------------
# torch custom ops (the "PyTorch-ROCm custom op" face of the C ABI)
# --------------------------------------------------------------------------------------
def _session(handle: int) -> Session:
    s = Session._registry.get(handle)
    if s is None:
        raise _capi.HepError("unknown libhep session handle")
    return s


@torch.library.custom_op("hep::forward", mutates_args=())
def hep_forward(x: torch.Tensor, handle: int) -> List[torch.Tensor]:
    return _forward_flat(x, handle)


def _forward_flat(x, handle):
    f, *outs = _session(handle).forward(x, want_features=True)
    return list(f) + list(outs)


@hep_forward.register_fake
def _(x, handle):
    s = _session(handle)
    B = x.shape[0]
    return [x.new_empty((B, s.fpn_w, l, l)) for l in s.levels] + [x.new_empty((B, s.num_anchors, k)) for k in s.out_width]


@torch.library.custom_op("hep::decode", mutates_args=())
def hep_decode(regression: torch.Tensor, translation_raw: torch.Tensor, camera: torch.Tensor, handle: int) -> List[torch.Tensor]:
    return list(_session(handle).decode(regression, translation_raw, camera))


@hep_decode.register_fake
def _(regression, translation_raw, camera, handle):
    return [torch.empty_like(regression), torch.empty_like(translation_raw)]


_FILTER_KEYS = ("boxes", "scores", "labels", "rotation", "translation", "hand", "index", "count")


@torch.library.custom_op("hep::filter", mutates_args=())
def hep_filter(boxes: torch.Tensor, classification: torch.Tensor, rotation: torch.Tensor, translation: torch.Tensor, hand: torch.Tensor,
               score_threshold: float, nms_threshold: float, max_detections: int, handle: int, class_specific_filter: bool = True) -> List[torch.Tensor]:
    """filter_detections (hmdegopose/layers.py:264-400): boxes, scores, labels, rotation, translation, hand, index, count."""
    d = _session(handle).filter(boxes, classification, rotation, translation, hand, score_threshold, nms_threshold, max_detections, class_specific_filter)
    return [d[k] for k in _FILTER_KEYS]


@hep_filter.register_fake
def _(boxes, classification, rotation, translation, hand, score_threshold, nms_threshold, max_detections, handle, class_specific_filter=True):
    B, M = boxes.shape[0], max_detections
    f = lambda *s: boxes.new_empty(s)
    i = lambda *s: boxes.new_empty(s, dtype=torch.int32)
    return [f(B, M, 4), f(B, M), i(B, M), f(B, M, 3), f(B, M, 3), f(B, M, 63), i(B, M), i(B)]


@torch.library.custom_op("hep::preprocess", mutates_args=())
def hep_preprocess(images_u8: torch.Tensor, handle: int) -> torch.Tensor:
    """preprocess_image (generators/colibri_common.py:622-656) for a batch of uint8 RGB frames; returns [B,3,S,S] float32
    (contiguous copy of the NCHW view: a custom op may not return a view of its own allocation's permutation)."""
    return _session(handle).preprocess(images_u8).contiguous()


@hep_preprocess.register_fake
def _(images_u8, handle):
    s = _session(handle)
    return images_u8.new_empty((images_u8.shape[0], 3, s.size, s.size), dtype=torch.float32)


# --------------------------------------------------------------------------------------
# nn.Module drop-in
# --------------------------------------------------------------------------------------
class _Node(nn.Module):
    """Plain container; the parameter tree only has to reproduce the reference's key names."""


def _attach(root: nn.Module, key: str, shape: tuple):
    *path, leaf = key.split(".")
    m = root
    for p in path:
        if p not in m._modules:
            m.add_module(p, _Node())
        m = m._modules[p]
    if leaf == "num_batches_tracked":
        m.register_buffer(leaf, torch.zeros((), dtype=torch.int64))
    elif leaf in ("running_mean", "running_var"):
        m.register_buffer(leaf, torch.zeros(shape) if leaf == "running_mean" else torch.ones(shape))
    else:
        m.register_parameter(leaf, nn.Parameter(torch.zeros(shape)))


class HMDEgoPose(nn.Module):
    """Drop-in for ``backbone.HMDEgoPose`` on the MI355X inference path.

    Same constructor arguments as the reference (backbone.py:13-16); extra keyword
    ``precision`` = "fp32" (default: matches the reference within 1e-3) or "bf16".
    """

    def __init__(self, params, num_classes=1, compound_coef=0, load_weights=False, onnx_export=False,
                 input_sizes=(512, 640, 768, 896, 1024, 1280, 1280, 1536, 1536), precision: str = "fp32", **kwargs):
        super().__init__()
        if int(params.get("iter", 0)) != 0:
            raise ValueError("params['iter'] must be 0: the reference's iterative refinement sub-nets hard-code their input "
                             "widths and are only usable with --iter=0 (hmdegopose/model.py:244,255-258; README.md:137,153)")
        if load_weights:
            raise ValueError("load_weights=True would download ImageNet weights; load a checkpoint with load_state_dict instead")
        self.compound_coef = int(compound_coef)
        self.num_classes = int(num_classes)
        self.onnx_export = onnx_export
        self.input_sizes = list(input_sizes)
        self.precision = precision
        self.arch = get_arch(self.compound_coef)
        for key, shape in param_spec(self.compound_coef, num_classes):
            _attach(self, key, shape)
        self.reset_parameters()
        self._sessions: Dict[tuple, Session] = {}
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate())

    def reset_parameters(self, seed: int = 0):
        from .weights import seeded_state_dict
        with torch.no_grad():
            for k, v in seeded_state_dict(self.compound_coef, seed, num_classes=self.num_classes).items():
                self.state_dict()[k].copy_(v)

    def invalidate(self):
        """Drop the packed device weights; they are rebuilt on the next forward.  Called after
        load_state_dict; call it yourself after editing parameters in place."""
        for s in getattr(self, "_sessions", {}).values():
            s.close()
        self._sessions = {}

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        self.invalidate()
        return out

    def freeze_bn(self):
        """BatchNorm is folded into the convolutions at pack time; nothing to freeze."""

    def init_backbone(self, path):
        state = torch.load(path, map_location="cpu", weights_only=True)
        try:
            print(self.load_state_dict(state, strict=True))
        except RuntimeError as e:
            print("Ignoring " + str(e) + '"')

    def session(self, size: int, batch: int, device: torch.device) -> Session:
        key = (size, device.index, self.precision)
        s = self._sessions.get(key)
        if s is None or s.max_batch < batch:
            if s is not None:
                s.close()
            mb = max(batch, 16) if s is None else max(batch, 2 * s.max_batch)
            s = Session(self.state_dict(), self.compound_coef, size, mb, self.precision, device)
            self._sessions[key] = s
        return s

    def forward(self, inputs: torch.Tensor):
        if self.training:
            raise RuntimeError("hmd_ego_pose_amd.HMDEgoPose is the MI355X inference path: call .eval() first "
                               "(training through this module is not supported; use the reference for training)")
        if not inputs.is_cuda:
            raise RuntimeError("hmd_ego_pose_amd.HMDEgoPose runs on a ROCm device only (no CPU fallback): move the input with .cuda()")
        x = inputs if inputs.dtype == torch.float32 else inputs.float()
        s = self.session(int(x.shape[-1]), int(x.shape[0]), x.device)
        flat = torch.ops.hep.forward(x, s.handle)
        return tuple(flat[:5]), flat[5], flat[6], flat[7], flat[8], flat[9]


class TrainModelWithLoss(nn.Module):
    """Inference branch of the reference wrapper (train.py:23-40,72-85) on the GPU:
    forward -> box/translation decode -> score threshold 0.5, NMS 0.5, top-100, -1 padding.

    ``forward`` returns, like the reference, the six padded tensors of the LAST batch item
    (layers.py:466-482 overwrites ``output`` per item) as CPU tensors; ``detect`` returns all
    items on the device."""

    def __init__(self, model: HMDEgoPose):
        super().__init__()
        self.model = model

    @torch.no_grad()
    def detect(self, imgs, camera_params, score_threshold=0.5, nms_threshold=0.5, max_detections=100, class_specific_filter=True):
        _, regression, classification, rotation, translation_raw, hand = self.model(imgs)
        s = self.model.session(int(imgs.shape[-1]), int(imgs.shape[0]), imgs.device)
        boxes, translation = torch.ops.hep.decode(regression, translation_raw, camera_params.to(imgs.device), s.handle)
        out = torch.ops.hep.filter(boxes, classification, rotation, translation, hand, float(score_threshold), float(nms_threshold),
                                   int(max_detections), s.handle, bool(class_specific_filter))
        return dict(zip(_FILTER_KEYS, out))

    @torch.no_grad()
    def validation_losses(self, imgs, camera_params, model_3d_points, classification_gt, regression_gt, transformation_gt, coords_3d_gt=None,
                          num_rotation_parameters: int = 3):
        """The ``is_losses=True`` branch of the reference wrapper (train.py:42-70) as FORWARD VALUES: HIP forward -> translation
        decode (``format_translation``) -> ``batch_iterate`` on the device (hep_losses_device) -> the reference's loss weights
        (rotation x 100, translation x 0.1) and their sum.  Six 0-dim float32 tensors on the device, in the reference's order
        [classification, regression, rotation, translation, hands, total].  No autograd graph: this is what a validation pass
        logs; optimising through the HIP forward is out of scope (the model refuses ``train()`` mode)."""
        from . import training
        _, regression, classification, rotation, translation_raw, hand = self.model(imgs)
        s = self.model.session(int(imgs.shape[-1]), int(imgs.shape[0]), imgs.device)
        _boxes, translation = torch.ops.hep.decode(regression, translation_raw, camera_params.to(imgs.device, torch.float32), s.handle)
        f = lambda t: None if t is None else torch.as_tensor(t).to(imgs.device, torch.float32)
        out, _per = training.losses(f(classification_gt), classification, f(regression_gt), regression, f(transformation_gt),
                                    torch.cat((rotation, translation), dim=2), f(coords_3d_gt), hand if coords_3d_gt is not None else None,
                                    model_3d_points, num_rotation_parameters)
        cls_l, reg_l, rot_l, tr_l, hand_l = out[0] * 1.0, out[1] * 1.0, out[2] * 100, out[3] * 0.1, out[4] * 1.0      # train.py:61-65
        return [cls_l, reg_l, rot_l, tr_l, hand_l, cls_l + reg_l + rot_l + tr_l + hand_l]

    def forward(self, imgs, camera_params, is_losses=False, model_3d_points=None, classification_gt=None, regression_gt=None,
                transformation_gt=None, coords_3d_gt=None, params=None, **kwargs):
        if is_losses:
            if model_3d_points is None or classification_gt is None or regression_gt is None or transformation_gt is None:
                raise ValueError("is_losses=True needs model_3d_points, classification_gt, regression_gt and transformation_gt (train.py:189-197)")
            return self.validation_losses(imgs, camera_params, model_3d_points, classification_gt, regression_gt, transformation_gt, coords_3d_gt,
                                          int((params or {}).get("num_rotation_parameters", 3)))
        d = self.detect(imgs, camera_params)
        last = imgs.shape[0] - 1
        return [d[k][last].cpu() for k in ("boxes", "scores", "labels", "rotation", "translation", "hand")]
