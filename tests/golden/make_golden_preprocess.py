#!/usr/bin/env python3
"""Golden vectors for ``preprocess_image`` from the REAL reference (generators/colibri_common.py:622-656).

Runs only in the build container (needs /root/reference, read-only).  The reference module imports cv2 and imgaug at
load time, which this image lacks; they are stubbed (as torchvision / tensorflow are for the network): ``cv2.resize`` is
replaced by a function that returns its input when the requested size equals the input size - what OpenCV does - and
refuses anything else, so ONLY the no-resize branch (every 256x256 syn_colibri frame at S = 256, and frames whose longer
side already equals S) is pinned.  The resize branch stays parity-unpinned (oracle/decode_ref.resize_bilinear_u8).

    python tests/golden/make_golden_preprocess.py       # writes tests/golden/preprocess.npz
"""
import hashlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/pytorch-sandbox"
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _resize_identity_only(image, dsize):
    if tuple(dsize) != (image.shape[1], image.shape[0]):
        raise NotImplementedError("cv2 is not available: only the identity resize can be pinned")
    return image.copy()


def cases():
    """(name, uint8 HWC image, network size) - seeded, regenerated identically by the tests."""
    rng = np.random.Generator(np.random.PCG64(2024))
    out = []
    for name, h, w, size in (("syn256", 256, 256, 256), ("wide", 200, 256, 256), ("tall", 256, 131, 256), ("s128", 128, 128, 128)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        img[0, 0] = (0, 255, 128)
        out.append((name, img, size))
    return out


def main():
    _stub("cv2", resize=_resize_identity_only)
    ia = _stub("imgaug"); ia.parameters = _stub("imgaug.parameters"); ia.random = _stub("imgaug.random")
    aug = _stub("imgaug.augmenters", Augmenter=object, Sequential=object)
    for sub in ("meta", "arithmetic", "pillike"):
        setattr(aug, sub, _stub("imgaug.augmenters." + sub, Augmenter=object, Sequential=object))
    ia.augmenters = aug
    _stub("generators.utils.compute_overlap", compute_overlap=None, wrapper_c_min_distances=None)
    sys.path.insert(0, REF)
    try:
        from generators.colibri_common import Generator
    except Exception as e:                                   # randaug builds augmenter classes at import time
        print("import of generators.colibri_common failed under the stubs:", repr(e))
        raise
    fx = {}
    for name, img, size in cases():
        self = types.SimpleNamespace(image_size=size)
        out, scale = Generator.preprocess_image(self, img)
        assert out.dtype == np.float32 and out.shape == (size, size, 3) and scale == 1.0
        fx[name + "_sha256"] = np.frombuffer(hashlib.sha256(out.tobytes()).digest(), dtype=np.uint8)
        fx[name + "_slice"] = out.reshape(-1)[::997].copy()
        fx[name + "_sum"] = np.array([out.astype(np.float64).sum()])
    np.savez_compressed(os.path.join(HERE, "preprocess.npz"), **fx)
    print("wrote preprocess.npz:", sorted(fx))


if __name__ == "__main__":
    main()
