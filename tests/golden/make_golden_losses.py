#!/usr/bin/env python3
"""Golden vectors for the training-side losses from the REAL reference.

Runs only in the build container (needs /root/reference, read-only).  ``hmdegopose/loss.py`` is imported unchanged
(torchvision / tensorflow / the Cython extension are stubbed as for the network: none of them is used by the losses) and
``batch_iterate`` is run on the seeded cases of ``tests/_util.py::loss_cases``.  Stored: the five returned scalars per case.

    python tests/golden/make_golden_losses.py       # writes tests/golden/losses.npz
"""
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


def main():
    import torch
    _stub("torchvision"); _stub("torchvision.ops"); _stub("torchvision.ops.boxes", nms=None)
    tf = _stub("tensorflow"); tf.keras = _stub("tensorflow.keras")
    _stub("generators.utils.compute_overlap", compute_overlap=None, wrapper_c_min_distances=None)
    sys.path.insert(0, REF)
    from hmdegopose.loss import batch_iterate                        # noqa: E402
    from tests._util import loss_cases
    out = {}
    for name, c in loss_cases().items():
        t = {k: torch.from_numpy(v) for k, v in c.items() if k != "model_points"}
        res = batch_iterate(t["gt_classification"], t["classification"], t["gt_regression"], t["regression"],
                            t["gt_transformation"], t["transformation"], t["gt_hand"], t["hand"], c["model_points"], 3)
        out[name] = np.array([float(r.reshape(-1)[0]) for r in res], dtype=np.float64)
        print(name, out[name])
    np.savez(os.path.join(HERE, "losses.npz"), **out)


if __name__ == "__main__":
    main()
