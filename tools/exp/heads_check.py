#!/usr/bin/env python3
"""The depth-first head kernel (HEP_HEADS_FUSED=1, k_heads.hip) against the launch-by-launch towers on the same frames: bit equality of the
five head outputs, the plans and their stand-alone times.   usage: python tools/exp/heads_check.py [batch] [size]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from hmd_ego_pose_amd.model import Session
from hmd_ego_pose_amd.weights import seeded_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
sd = seeded_state_dict(0, 0)
rng = np.random.Generator(np.random.PCG64([0, 0x1234]))
x = torch.from_numpy(rng.standard_normal((B, 3, size, size)).astype(np.float32)).cuda()
res = {}
for fused in ("0", "1"):
    os.environ["HEP_HEADS_FUSED"] = fused
    s = Session(sd, 0, size, B, "bf16")
    out = [t.clone() for t in s.forward(x)[1:]]
    torch.cuda.synchronize()
    total, per = s.profile(B, 20, per_kernel=True)
    ks = s.kernels(B)
    print(f"HEP_HEADS_FUSED={fused}: {len(ks)} launches, one batch {total * 1e3:.1f} us; " + ", ".join(f"{n} {t * 1e3:.1f}" for (n, _b, _f, _y), t in zip(ks, per) if n.startswith("heads")))
    res[fused] = out
    s.close()
for n, a, b in zip(("regression", "classification", "rotation", "translation", "hand"), res["0"], res["1"]):
    print(f"{n}: equal {bool(torch.equal(a, b))}  unequal elements {int((a != b).sum())} of {a.numel()}  max |d| {float((a - b).abs().max()):.3e}  finite {bool(torch.isfinite(b).all())}")
