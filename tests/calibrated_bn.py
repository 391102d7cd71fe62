#!/usr/bin/env python3
"""BatchNorm running statistics CALIBRATED on seeded frames - used by tests/precision_calibrated.py.

The plain seeded recipe (hmd_ego_pose_amd.weights.seeded_state_dict) draws running_mean / running_var at random, so a BatchNorm
output is whatever the random statistics make of it and the network amplifies a rounding error ~1000 x on its way to the heads
(DESIGN.md section 3).  A trained network's statistics are those of its own activations: every BatchNorm output is ~N(beta, gamma^2).
This script makes such a recipe without training: one pass of the CPU oracle (oracle/efficientpose_ref.py) over calibration
frames in which every BatchNorm measures the per-channel mean / variance of ITS input - behind the already calibrated layers in front
of it - and normalises with them (what `model.train()` + momentum 1 would record).  Only the statistics are stored (~20k channels);
`hmd_ego_pose_amd.weights.calibrated_state_dict` overlays them on the seeded weights.  TEST INFRASTRUCTURE (it runs the oracle).

    python tests/golden/make_calibrated_bn.py            # phi 0, seed 0
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def calibrate(phi, seed, size, frames):
    import torch
    from hmd_ego_pose_amd.weights import seeded_state_dict
    from oracle import efficientpose_ref as R
    sd = dict(seeded_state_dict(phi, seed))
    rng = np.random.Generator(np.random.PCG64([seed, 0xCA1B]))             # calibration frames: not the frames any test evaluates
    x = torch.from_numpy(rng.standard_normal((frames, 3, size, size)).astype(np.float32))
    stats = {}
    plain_bn = R.bn

    def measuring_bn(sd_, p, t):
        m = t.mean(dim=(0, 2, 3)); v = t.var(dim=(0, 2, 3), unbiased=False)
        sd_[p + ".running_mean"] = m.clone(); sd_[p + ".running_var"] = v.clone()
        stats[p + ".running_mean"] = m.numpy().copy(); stats[p + ".running_var"] = v.numpy().copy()
        return plain_bn(sd_, p, t)

    R.bn = measuring_bn
    try:
        with torch.no_grad():
            R.forward(sd, x, phi)
    finally:
        R.bn = plain_bn
    return stats


def calibrated_state_dict(phi, seed=0, size=256, frames=8):
    import torch
    from hmd_ego_pose_amd.weights import seeded_state_dict
    sd = seeded_state_dict(phi, seed)
    for k, v in calibrate(phi, seed, size, frames).items():
        sd[k] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return sd
