#!/usr/bin/env python3
"""What does bf16 storage cost on a WELL-CONDITIONED network (TEST INFRASTRUCTURE, CPU only, not a pytest file)?

The seeded weights amplify a rounding error ~1000 x: is the 28 mm ADD of the bf16 session a property of
bf16 or of those weights?  Second recipe: the same weights with every BatchNorm's running statistics calibrated on seeded frames
(tests/calibrated_bn.py), i.e. every BatchNorm output ~N(beta, gamma^2) as in a trained network.  For
both recipes, on the CPU oracle: how far a 1e-6 relative input perturbation has grown at the heads (amplification), and the ADD
(mm, as bench.py's add_vs_ref) of the bf16-emulating oracle and of m-mantissa-bit storage against the fp32 oracle.

    python tests/precision_calibrated.py > profiles/r05/d_precision_calibrated_recipe.md
"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from scipy.spatial.transform import Rotation

from hmd_ego_pose_amd.weights import seeded_state_dict
from tests.calibrated_bn import calibrated_state_dict
from oracle import decode_ref as D
from oracle import efficientpose_ref as R
from tests.precision_sweep import qbits


def main():
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    phi, size, nf = 0, 256, 8
    rng = np.random.Generator(np.random.PCG64(99))
    x = torch.from_numpy(rng.standard_normal((nf, 3, size, size)).astype(np.float32))
    cam = np.array([[480, 480, 128, 128, 1000, 1.0]] * nf, np.float32)
    pts = (rng.standard_normal((1000, 3)) * np.array([40.0, 25.0, 60.0])).astype(np.float32)
    _, t_anchors = D.anchors_for_size(size)
    print(f"| recipe (phi {phi} @ {size}, {nf} frames, CPU oracle) | head amplification of a 1e-6 input perturbation | ADD mm bf16 (7 bits) | 10 bits | 12 bits | 14 bits | 16 bits | head drift bf16 (mean rel.) | score spread (max - median) |")
    print("|---|---|---|---|---|---|---|---|---|")
    for name, sd in (("seeded (random BatchNorm statistics)", seeded_state_dict(phi, 0)), ("calibrated BatchNorm statistics", calibrated_state_dict(phi, 0))):
        with torch.no_grad():
            _, reg, cls, rot, trn, hand = R.forward(sd, x, phi)
            _, _, _, rot_p, trn_p, _ = R.forward(sd, x * (1 + 1e-6), phi)
        amp = float(((trn_p - trn).abs().mean() / trn.abs().mean()) / 1e-6)
        t_ref = D.decode_translation(t_anchors, trn.numpy(), cam)
        idx = cls[:, :, 0].argmax(dim=1).numpy()
        pick = lambda v: np.stack([v[i, idx[i]] for i in range(nf)])

        def add_mm(g_rot, g_trn):
            g_t = D.decode_translation(t_anchors, g_trn.numpy(), cam)
            out = []
            for i in range(nf):
                R0 = Rotation.from_rotvec(pick(rot.numpy())[i] * math.pi).as_matrix()
                R1 = Rotation.from_rotvec(pick(g_rot.numpy())[i] * math.pi).as_matrix()
                out.append(np.linalg.norm((pts @ R0.T + pick(t_ref)[i]) - (pts @ R1.T + pick(g_t)[i]), axis=1).mean())
            return float(np.mean(out))

        cols = []
        drift = None
        for m in (7, 10, 12, 14, 16):
            with torch.no_grad():
                g = R.forward_emulated(sd, x, phi, q_act=qbits(m), q_w=qbits(m))
            cols.append(add_mm(g[3], g[4]))
            if m == 7:
                drift = float((g[4] - trn).abs().mean() / trn.abs().mean())
        sc = cls[:, :, 0]
        spread = float((sc.max(dim=1).values - sc.median(dim=1).values).mean())
        print(f"| {name} | {amp:.0f} x | " + " | ".join(f"{c:.3f}" for c in cols) + f" | {drift:.4f} | {spread:.3f} |", flush=True)


if __name__ == "__main__":
    main()
