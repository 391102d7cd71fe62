#!/usr/bin/env python3
"""Golden vectors for the training-side anchor-target assignment from the REAL reference.

Runs only in the build container (needs /root/reference, read-only, gcc and Cython).  The reference's own Cython
extension ``generators/utils/compute_overlap.pyx`` (+ ``calc_min_distances.h``) is compiled OUT OF TREE in a temporary
directory and imported as ``generators.utils.compute_overlap``; then the reference's ``generators/utils/anchors.py`` is
imported unchanged and ``anchor_targets_bbox`` is run on seeded boxes.  Stored: sha256 of every returned array, the
anchor-state histogram and a strided slice.

    python tests/golden/make_golden_targets.py       # writes tests/golden/anchor_targets.npz
"""
import hashlib
import importlib
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/pytorch-sandbox"
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)


def cases(anchors):
    """(name, image shape, boxes [K,4] f64, labels [K], transformation targets [K,6], coords [K,63]) - seeded."""
    rng = np.random.Generator(np.random.PCG64(77))
    out = []
    for name, K, shape in (("one", 1, (256, 256, 3)), ("three", 3, (256, 256, 3)), ("none", 0, (256, 256, 3)), ("small_image", 2, (200, 180, 3)),
                           ("exact_anchor", 2, (256, 256, 3))):
        x0 = rng.uniform(0, 150, K); y0 = rng.uniform(0, 150, K)
        boxes = np.stack([x0, y0, x0 + rng.uniform(20, 100, K), y0 + rng.uniform(20, 100, K)], axis=1) if K else np.zeros((0, 4))
        if name == "exact_anchor":
            boxes[0] = anchors[4000].astype(np.float64)            # IoU exactly 1 with one anchor, ties among its neighbours
            boxes[1] = [1000., 1000., 1010., 1010.]                # overlaps nothing: its "best" anchor is anchor 0
        out.append((name, shape, boxes, np.zeros(K), rng.standard_normal((K, 6)), rng.standard_normal((K, 63))))
    return out


def digest(arrs):
    d = {}
    for k, a in arrs.items():
        a = np.ascontiguousarray(a)
        d[k + "_sha256"] = np.frombuffer(hashlib.sha256(a.tobytes()).digest(), dtype=np.uint8)
        d[k + "_slice"] = a.reshape(-1)[::1201].copy()
    st = arrs["regression"][..., -1]
    d["state_hist"] = np.array([(st == -1).sum(), (st == 0).sum(), (st == 1).sum()])
    return d


def main():
    tmp = tempfile.mkdtemp(prefix="hep_cy_")
    try:
        for f in ("compute_overlap.pyx", "calc_min_distances.h"):
            shutil.copy(os.path.join(REF, "generators", "utils", f), tmp)       # out of tree, deleted below
        with open(os.path.join(tmp, "setup.py"), "w") as f:
            f.write("from setuptools import setup, Extension\nfrom Cython.Build import cythonize\nimport numpy\n"
                    "setup(ext_modules=cythonize([Extension('compute_overlap', ['compute_overlap.pyx'], include_dirs=[numpy.get_include(), '.'])], language_level=3))\n")
        subprocess.run([sys.executable, "setup.py", "build_ext", "--inplace"], cwd=tmp, check=True, capture_output=True)
        sys.path.insert(0, tmp)
        sys.modules["generators.utils.compute_overlap"] = importlib.import_module("compute_overlap")
        sys.path.insert(0, REF)
        from generators.utils.anchors import anchor_targets_bbox, anchors_for_shape
        anchors, t_anchors = anchors_for_shape((256, 256))
        fx = {}
        from generators.utils.anchors import bbox_transform, compute_gt_annotations
        for name, shape, boxes, labels, tt, coords in cases(anchors):
            K = boxes.shape[0]
            if K:
                # the core of the assignment, valid for any number of boxes
                pos, ign, arg = compute_gt_annotations(anchors, boxes)
                core = {"positive": pos.astype(np.uint8), "ignore": ign.astype(np.uint8), "argmax": arg.astype(np.int64),
                        "bbox_transform": bbox_transform(anchors, boxes[arg, :]).astype(np.float64)}
                for k, v in core.items():
                    fx[f"{name}_{k}_sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(v).tobytes()).digest(), dtype=np.uint8)
                fx[f"{name}_counts"] = np.array([pos.sum(), ign.sum()])
            # the batch assembly: the reference reshapes coords_3d to (1, 63) and indexes it with the gt index, which only
            # works while every anchor prefers box 0 (one box per image: all of the dataset)
            ann = {"bboxes": boxes, "labels": labels, "transformation_targets": tt,
                   "coords_3d": coords[:1].reshape(1, 21, 3) if K else np.zeros((1, 21, 3))}
            try:
                lab, reg, tra, crd = anchor_targets_bbox(anchors, [np.zeros(shape)], [ann], 1, 3, 3, t_anchors)
            except IndexError:
                continue
            for k, v in digest({"labels": lab, "regression": reg, "transformation": tra, "coords": crd}).items():
                fx[f"{name}_{k}"] = v
        np.savez_compressed(os.path.join(HERE, "anchor_targets.npz"), **fx)
        print("wrote anchor_targets.npz:", sorted(fx))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
