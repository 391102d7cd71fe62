"""Evaluator for the MI355X inference path: the host side of the reference's ``evaluate.py`` loop.

What the reference does per image (pytorch-sandbox/eval/common.py:357-447, 866-1121; evaluate.py:58-128) and where
it runs here:

  load + preprocess           Linemod-style folder reader below; uint8 -> normalised fp32 on the GPU
                              (``Session.preprocess``, hep_preprocess_u8_device)
  forward / decode / filter   libhep kernels (``TrainModelWithLoss.detect``)
  post-filter (a15)           ``post_filter``: boxes /= scale, rotations *= pi, score > threshold, score order,
                              first max_detections (eval/common.py:419-447)
  matching                    ``compute_overlap`` (IoU with the +1 pixel convention of
                              generators/utils/compute_overlap.pyx:33-73), greedy by score like common.py:942-957
  ADD / ADD-S                 hep_pose_errors on the GPU (csrc/k_eval.hip; eval/common.py:682-746,
                              calc_min_distances.h:24-35)
  5 cm / 5 degree, t / R diff ``calc_rotation_diff`` etc. below (eval/common.py:750-778,829-833), float64 numpy
  2D reprojection             ``reprojection_distance`` (eval/common.py:646-679): cv2.projectPoints with zero rotation, translation
                              and distortion is the pinhole projection u = fx X / Z + cx, v = fy Y / Z + cy (float64)
  hand joints                 mean end-point error over the 21 joints in mm (eval/common.py:970-982), ground truth from the
                              dataset's ``hands/<frame>_coords_3d.npy`` (generators/colibri.py:430-436)
  AP                          ``compute_ap`` (eval/common.py:328-354)

``python -m hmd_ego_pose_amd.evaluate --dataset-path <object folder> --weights ckpt.pth --phi 0`` runs it on a
supplied dataset; nothing ships with the reference (dataset, checkpoint and mesh are absent), so the numbers of the
paper cannot be reproduced here - tests drive this module with a synthetic folder.

Not reproduced: the visualisations (draw_samplevis / MANO hand meshes, eval/common.py:449-590); ``cv2.Rodrigues`` and
``cv2.projectPoints`` are restated (parity unpinned: cv2 is absent from the build image).
"""
from __future__ import annotations

import argparse
import ctypes
import math
import os
import struct
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _capi


# ----------------------------------------------------------------------------------------------------------------
# small host-side pieces of eval/common.py
# ----------------------------------------------------------------------------------------------------------------
def post_filter(det: Dict[str, torch.Tensor], scale: float, score_threshold: float = 0.05, max_detections: int = 10):
    """``_get_detections`` after the model call, eval/common.py:419-447, for ONE image: ``det`` holds the padded rows of
    the detection filter (boxes [M,4], scores [M], labels [M], rotation [M,3], translation [M,3], hand [M,63]; padding
    rows carry -1).  Returns numpy arrays (boxes, scores, labels, rotations in radians, translations, hands) of the
    rows with score > threshold in descending score order (equal scores: lower row first), at most max_detections."""
    boxes = det["boxes"].detach().cpu().numpy().astype(np.float32) / np.float32(scale)
    scores = det["scores"].detach().cpu().numpy().astype(np.float32)
    rotations = det["rotation"].detach().cpu().numpy().astype(np.float32) * np.float32(math.pi)
    ind = np.nonzero(scores > score_threshold)[0]
    order = np.argsort(-scores[ind], kind="stable")[:max_detections]
    sel = ind[order]
    return (boxes[sel], scores[sel], det["labels"].detach().cpu().numpy()[sel], rotations[sel],
            det["translation"].detach().cpu().numpy()[sel], det["hand"].detach().cpu().numpy()[sel])


def compute_overlap(boxes: np.ndarray, query_boxes: np.ndarray) -> np.ndarray:
    """IoU matrix [N,K] with the "+1" pixel convention of generators/utils/compute_overlap.pyx:33-73 (float64)."""
    b = np.asarray(boxes, dtype=np.float64)[:, None, :]
    q = np.asarray(query_boxes, dtype=np.float64)[None, :, :]
    iw = np.minimum(b[..., 2], q[..., 2]) - np.maximum(b[..., 0], q[..., 0]) + 1
    ih = np.minimum(b[..., 3], q[..., 3]) - np.maximum(b[..., 1], q[..., 1]) + 1
    inter = np.where((iw > 0) & (ih > 0), iw * ih, 0.0)
    ua = (b[..., 2] - b[..., 0] + 1) * (b[..., 3] - b[..., 1] + 1) + (q[..., 2] - q[..., 0] + 1) * (q[..., 3] - q[..., 1] + 1) - inter
    return np.where(inter > 0, inter / ua, 0.0)


def compute_ap(recall: np.ndarray, precision: np.ndarray) -> float:
    """py-faster-rcnn average precision as used at eval/common.py:328-354."""
    mrec = np.concatenate(([0.], recall, [1.]))
    mpre = np.concatenate(([0.], precision, [0.]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = max(mpre[i - 1], mpre[i])
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return float(np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1]))


def axis_angle_to_matrix(rvec: Sequence[float]) -> np.ndarray:
    """cv2.Rodrigues (vector -> matrix) restated, float64 (colibri_common.py:803-813)."""
    r = np.asarray(rvec, dtype=np.float64).reshape(3)
    th = float(np.linalg.norm(r))
    if th < 1e-12:
        return np.eye(3)
    k = r / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) * math.cos(th) + (1 - math.cos(th)) * np.outer(k, k) + math.sin(th) * K


def matrix_to_axis_angle(R: np.ndarray) -> np.ndarray:
    """cv2.Rodrigues (matrix -> vector) restated, float64 (colibri_common.py:791-801)."""
    R = np.asarray(R, dtype=np.float64).reshape(3, 3)
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s, c = float(np.linalg.norm(v)) / 2.0, (np.trace(R) - 1.0) / 2.0
    th = math.atan2(s, c)                        # well conditioned at 0 and at pi (acos of the trace is not)
    if s < 1e-10:
        if c > 0:
            return v / 2.0                       # theta -> 0: r = vee(R - R^T) / 2
        A = (R + np.eye(3)) / 2.0                # theta -> pi: the antisymmetric part vanishes, the axis comes from R + I = 2 k k^T
        k = np.sqrt(np.maximum(np.diag(A), 0.0))
        i = int(np.argmax(k))
        k = A[i] / max(k[i], 1e-300)
        if np.dot(k, v) < 0:
            k = -k
        return k / np.linalg.norm(k) * th
    return v / (2.0 * s) * th


def calc_rotation_diff(R_gt: np.ndarray, R_pr: np.ndarray) -> float:
    """Angular distance in degrees, eval/common.py:762-778."""
    tr = (np.trace(np.dot(R_pr, R_gt.T)) - 1.0) / 2.0
    return abs(float(np.rad2deg(np.arccos(min(1.0, max(-1.0, tr))))))


def reprojection_distance(points: np.ndarray, R_gt: np.ndarray, t_gt: np.ndarray, R_pr: np.ndarray, t_pr: np.ndarray, K: np.ndarray) -> float:
    """Mean pixel distance between the model points projected with the two poses (check_6d_pose_2d_reprojection,
    eval/common.py:646-679; correct when <= 5 px).  Pinhole projection, float64."""
    K = np.asarray(K, dtype=np.float64)

    def project(R, t):
        p = np.dot(np.asarray(points, np.float64), np.asarray(R, np.float64).T) + np.asarray(t, np.float64).reshape(1, 3)
        return np.stack([K[0, 0] * p[:, 0] / p[:, 2] + K[0, 2], K[1, 1] * p[:, 1] / p[:, 2] + K[1, 2]], axis=-1)
    return float(np.linalg.norm(project(R_gt, t_gt) - project(R_pr, t_pr), axis=-1).mean())


def pose_errors(points: np.ndarray, rvec_gt: np.ndarray, t_gt: np.ndarray, rvec_pr: np.ndarray, t_pr: np.ndarray,
                device: int = 0, max_points: int = 1000) -> Tuple[np.ndarray, np.ndarray]:
    """ADD and ADD-S mean distances of D pose pairs on the GPU (hep_pose_errors).  Rotations: axis-angle, radians."""
    pts = np.ascontiguousarray(points, dtype=np.float32)
    arrs = [np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 3) for a in (rvec_gt, t_gt, rvec_pr, t_pr)]
    d = arrs[0].shape[0]
    add, add_s = np.zeros(d, np.float64), np.zeros(d, np.float64)
    _capi.check(_capi.lib().hep_pose_errors(device, pts.ctypes.data, pts.shape[0], *[a.ctypes.data for a in arrs], d, max_points,
                                            add.ctypes.data, add_s.ctypes.data))
    return add, add_s


# ----------------------------------------------------------------------------------------------------------------
# Linemod-style object folder (generators/colibri.py:60-120,293-443,460-510)
# ----------------------------------------------------------------------------------------------------------------
def load_ply_vertices(path: str) -> np.ndarray:
    """x, y, z of every vertex of an ASCII or binary-little-endian PLY file (generators/colibri.py:293-307)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, n, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    n = int(tok[2])
            elif tok[0] == "property" and in_vertex and tok[1] != "list":
                props.append((tok[2], tok[1]))
            elif tok[0] == "end_header":
                break
        names = [p[0] for p in props]
        if not all(k in names for k in "xyz"):
            raise ValueError(f"{path}: vertex element has no x/y/z")
        if fmt == "ascii":
            rows = np.loadtxt(f, max_rows=n, ndmin=2)
            return np.stack([rows[:, names.index(k)] for k in "xyz"], axis=-1).astype(np.float32)
        if fmt != "binary_little_endian":
            raise ValueError(f"{path}: unsupported PLY format {fmt}")
        code = {"float": "f4", "float32": "f4", "double": "f8", "float64": "f8", "uchar": "u1", "uint8": "u1", "char": "i1", "int8": "i1",
                "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4"}
        dt = np.dtype([(nm, "<" + code[ty]) for nm, ty in props])
        v = np.frombuffer(f.read(n * dt.itemsize), dtype=dt, count=n)
        return np.stack([v["x"], v["y"], v["z"]], axis=-1).astype(np.float32)


class LinemodFolder:
    """One object of a Linemod-format dataset as the reference's ColibriGenerator reads it: ``<root>/data/<NN>/`` with
    ``rgb/*.png``, ``mask/*.png``, ``gt_<fold>.yml``, ``info_<fold>.yml``, ``test_<fold>.txt`` and
    ``<root>/models/obj_<NN>.ply`` + ``models_info.yml``."""

    def __init__(self, dataset_path: str, object_id: int = 1, fold: int = 0, split: str = "test", image_extension: str = ".png"):
        import yaml
        loader = getattr(yaml, "CSafeLoader", yaml.SafeLoader)
        self.object_id = object_id
        self.object_path = os.path.join(dataset_path, "data", f"{object_id:02}")
        self.model_path = os.path.join(dataset_path, "models")
        with open(os.path.join(self.object_path, f"{split}_{fold}.txt")) as f:
            examples = [e.strip() for e in f if e.strip()]
        with open(os.path.join(self.object_path, f"gt_{fold}.yml")) as f:
            gt = yaml.load(f, Loader=loader)
        with open(os.path.join(self.object_path, f"info_{fold}.yml")) as f:
            info = yaml.load(f, Loader=loader)
        with open(os.path.join(self.model_path, "models_info.yml")) as f:
            models = yaml.load(f, Loader=loader)
        self.diameter = float(models[object_id]["diameter"])
        self.points = load_ply_vertices(os.path.join(self.model_path, f"obj_{object_id:02}.ply"))
        names = sorted(n for n in os.listdir(os.path.join(self.object_path, "rgb"))
                       if n.endswith(image_extension) and n[:-len(image_extension)] in examples)
        self.image_paths = [os.path.join(self.object_path, "rgb", n) for n in names]
        self.annotations, self.camera = [], []
        for n in names:
            key = int(n.split(".")[0])
            annos = [a for a in gt[key] if a["obj_id"] == object_id]
            if not annos:
                raise ValueError(f"no annotation of object {object_id} for frame {n}")
            a = annos[0]
            R = np.array(a["cam_R_m2c"], dtype=np.float64).reshape(3, 3)
            entry = {"rotation": matrix_to_axis_angle(R), "translation": np.array(a["cam_t_m2c"], dtype=np.float64),
                     "drill_tip": np.array(a.get("drill_tip_transform", [0, 0, 0, 1]), dtype=np.float64)}
            mask_path = os.path.join(self.object_path, "mask", n)
            if os.path.exists(mask_path):
                entry["bbox"] = self._bbox_from_mask(mask_path)
            else:                                     # Linemod's own field: x, y, w, h
                x, y, w, h = a["obj_bb"]
                entry["bbox"] = np.array([x, y, x + w, y + h], dtype=np.float32)
            hands_path = os.path.join(self.object_path, "hands", n[:-len(image_extension)] + "_coords_3d.npy")      # generators/colibri.py:430-436
            if os.path.exists(hands_path):
                entry["coords_3d"] = np.load(hands_path).astype(np.float64).reshape(21, 3)
            self.annotations.append(entry)
            self.camera.append(np.array(info[key]["cam_K"], dtype=np.float64).reshape(3, 3))

    @staticmethod
    def _bbox_from_mask(path: str) -> np.ndarray:
        """get_bbox_from_mask, colibri_common.py:540-561: (min_x, min_y, max_x, max_y) of the non-zero pixels."""
        from PIL import Image
        m = np.asarray(Image.open(path))
        ys, xs = np.nonzero(m.reshape(m.shape[0], m.shape[1], -1).max(axis=2))
        if ys.size == 0:
            return np.zeros(4, np.float32)
        return np.array([xs.min(), ys.min(), xs.max(), ys.max()], dtype=np.float32)

    def __len__(self):
        return len(self.image_paths)

    def load_image(self, i: int) -> np.ndarray:
        """uint8 RGB [H,W,3] (the reference reads BGR with cv2 and flips, colibri.py:546-552)."""
        from PIL import Image
        return np.asarray(Image.open(self.image_paths[i]).convert("RGB"))

    def camera_input(self, i: int, image_scale: float, translation_scale_norm: float = 1000.0) -> np.ndarray:
        """get_camera_parameter_input, colibri_common.py:658-678: [fx, fy, px, py, translation_scale_norm, image_scale]."""
        K = self.camera[i]
        return np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2], translation_scale_norm, image_scale], dtype=np.float32)


# ----------------------------------------------------------------------------------------------------------------
# the evaluation loop
# ----------------------------------------------------------------------------------------------------------------
def evaluate(dataset: LinemodFolder, model, image_size: int, score_threshold: float = 0.05, max_detections: int = 10,
             iou_threshold: float = 0.5, diameter_threshold: float = 0.1, batch_size: int = 16, device: Optional[torch.device] = None,
             detections_out: Optional[list] = None) -> Dict[str, float]:
    """eval/common.py:866-1121 for the single-class datasets of the reference.  ``model`` is a
    ``hmd_ego_pose_amd.TrainModelWithLoss`` (eval mode, on the GPU).  Frames must not need a resize
    (max(H, W) == image_size) unless the session's preprocess supports it.  Returns the metric dictionary."""
    device = device or torch.device("cuda", torch.cuda.current_device())
    n = len(dataset)
    all_det = []
    for i0 in range(0, n, batch_size):
        idx = list(range(i0, min(n, i0 + batch_size)))
        frames = [dataset.load_image(i) for i in idx]
        h, w = frames[0].shape[:2]
        if any(f.shape[:2] != (h, w) for f in frames):
            raise ValueError("frames of one batch must share a size")
        scale = image_size / max(h, w)
        u8 = torch.from_numpy(np.stack(frames)).to(device)
        x = model.model.session(image_size, len(idx), device).preprocess(u8)
        cam = torch.from_numpy(np.stack([dataset.camera_input(i, scale) for i in idx])).to(device)
        det = model.detect(x, cam)
        for j in range(len(idx)):
            all_det.append(post_filter({k: v[j] for k, v in det.items() if k != "count" and k != "index"}, scale, score_threshold, max_detections))
    if detections_out is not None:
        detections_out.extend(all_det)

    fp, tp, scores = [], [], []
    pairs = []                      # (rvec_gt, t_gt, rvec_pr, t_pr, drill_tip) of every correct 2D detection
    pair_cam, hand_err = [], []     # its camera matrix; mean joint distance in mm where the dataset carries hand joints
    for i, (boxes, sc, labels, rots, trans, hands) in enumerate(all_det):
        # eval/common.py:603-611: detections are split by label and only the generator's labels are evaluated; the object
        # folders of the reference hold ONE object (label 0), so rows a many-class model labels otherwise are dropped
        own = labels == 0
        boxes, sc, rots, trans, hands = boxes[own], sc[own], rots[own], trans[own], hands[own]
        ann = dataset.annotations[i]
        detected = False
        for d in range(boxes.shape[0]):
            scores.append(float(sc[d]))
            ov = compute_overlap(boxes[d:d + 1], ann["bbox"][None])[0, 0]
            if ov >= iou_threshold and not detected:
                detected = True
                fp.append(0); tp.append(1)
                pairs.append((ann["rotation"], ann["translation"], rots[d].astype(np.float64), trans[d].astype(np.float64), ann["drill_tip"]))
                pair_cam.append(dataset.camera[i])
                if "coords_3d" in ann:       # eval/common.py:970-982: mean over the 21 joints of |gt - pred|, metres -> mm
                    hand_err.append(float(np.linalg.norm(ann["coords_3d"] - hands[d].astype(np.float64).reshape(21, 3), axis=-1).mean() * 1000.0))
            else:
                fp.append(1); tp.append(0)
    num_ann = float(n)
    out: Dict[str, float] = {"num_annotations": num_ann, "num_matched": float(len(pairs))}
    order = np.argsort(-np.asarray(scores), kind="stable") if scores else np.zeros(0, np.int64)
    tpc, fpc = np.cumsum(np.asarray(tp, np.float64)[order]), np.cumsum(np.asarray(fp, np.float64)[order])
    out["AP"] = compute_ap(tpc / num_ann, tpc / np.maximum(tpc + fpc, np.finfo(np.float64).eps)) if num_ann else 0.0
    if pairs:
        rg, tg, rp, tpv, tips = (np.stack([p[k] for p in pairs]) for k in range(5))
        add, add_s = pose_errors(dataset.points, rg, tg, rp, tpv, device.index or 0)
        thr = dataset.diameter * diameter_threshold
        t_diff = np.linalg.norm(tg - tpv, axis=1)
        r_diff = np.array([calc_rotation_diff(axis_angle_to_matrix(a), axis_angle_to_matrix(b)) for a, b in zip(rg, rp)])
        tip_gt = np.stack([axis_angle_to_matrix(a) @ t[:3] + b for a, b, t in zip(rg, tg, tips)])
        tip_pr = np.stack([axis_angle_to_matrix(a) @ t[:3] + b for a, b, t in zip(rp, tpv, tips)])
        reproj = np.array([reprojection_distance(dataset.points, axis_angle_to_matrix(a), b, axis_angle_to_matrix(c), d_, K)
                           for a, b, c, d_, K in zip(rg, tg, rp, tpv, pair_cam)])
        out["2D_projection"] = float(np.sum(reproj <= 5.0) / num_ann)
        out["2D_projection_distance_mean"] = float(reproj.mean())
        if hand_err:
            out["hand_mean"] = float(np.mean(hand_err)); out["hand_std"] = float(np.std(hand_err))
        out.update({"ADD": float(np.sum(add <= thr) / num_ann), "ADD-S": float(np.sum(add_s <= thr) / num_ann),
                    "5cm_5deg": float(np.sum((t_diff <= 50) & (r_diff <= 5)) / num_ann),
                    "translation_mean": float(t_diff.mean()), "translation_std": float(t_diff.std()),
                    "rotation_mean": float(r_diff.mean()), "rotation_std": float(r_diff.std()),
                    "translation_tip_mean": float(np.linalg.norm(tip_gt - tip_pr, axis=1).mean()),
                    "ADD_distance_mean": float(add.mean()), "ADD_distance_std": float(add.std()),
                    "ADD-S_distance_mean": float(add_s.mean()), "ADD-S_distance_std": float(add_s.std())})
    else:
        out.update({"ADD": 0.0, "ADD-S": 0.0, "5cm_5deg": 0.0, "2D_projection": 0.0})
    return out


def main(argv=None):
    """The reference's ``evaluate.py`` entry point for this path (evaluate.py:18-128; its argument names)."""
    ap = argparse.ArgumentParser(description="Evaluate an HMD-EgoPose checkpoint on a Linemod-format object folder (MI355X)")
    ap.add_argument("--dataset-path", required=True, help="folder holding data/<NN>/ and models/ (reference: --dataset-path)")
    ap.add_argument("--object-id", type=int, default=1)
    ap.add_argument("--fold", type=int, default=0)
    ap.add_argument("--phi", type=int, default=0)
    ap.add_argument("--img-size", default="256,256")
    ap.add_argument("--weights", required=True, help=".pth state_dict (model. / model.module. prefixes are stripped) or a HEPW pack")
    ap.add_argument("--score-threshold", type=float, default=0.5)
    ap.add_argument("--batch-size", type=int, default=16)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "fp8"])
    a = ap.parse_args(argv)
    from . import HMDEgoPose, TrainModelWithLoss, load_pack, strip_checkpoint_prefix
    size = int(str(a.img_size).split(",")[0])
    if a.weights.endswith(".hepw"):
        state = {k: torch.from_numpy(np.array(v)) for k, v in load_pack(open(a.weights, "rb").read()).items()}
    else:
        state = strip_checkpoint_prefix(torch.load(a.weights, map_location="cpu", weights_only=True))
    m = HMDEgoPose({"iter": 0}, num_classes=1, compound_coef=a.phi, onnx_export=True, input_sizes=[size] * 9, precision=a.precision)
    m.load_state_dict(state, strict=False)
    model = TrainModelWithLoss(m.to("cuda").eval()).eval()
    res = evaluate(LinemodFolder(a.dataset_path, a.object_id, a.fold), model, size, score_threshold=a.score_threshold, batch_size=a.batch_size)
    for k, v in res.items():
        print(f"{k:>24s}: {v:.6g}")
    return res


if __name__ == "__main__":
    main()
