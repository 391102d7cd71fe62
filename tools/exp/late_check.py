#!/usr/bin/env python3
"""The image-resident late-block kernel (HEP_LATE=1, k_late.hip) against the launch-by-launch plan (HEP_LATE=0) on the same frames:
per-block differences of the stage tensors, the heads, the plans and their stand-alone times.
usage: python tools/exp/late_check.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from hmd_ego_pose_amd import _capi
from hmd_ego_pose_amd.model import Session
from hmd_ego_pose_amd.weights import seeded_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
phi, size = 0, 256
sd = seeded_state_dict(phi, 0)
rng = np.random.Generator(np.random.PCG64([0, 0x1234]))
x = torch.from_numpy(rng.standard_normal((B, 3, size, size)).astype(np.float32)).cuda()
res = {}
for late in ("0", "1"):
    os.environ["HEP_LATE"] = late
    s = Session(sd, phi, size, B, "bf16", flags=_capi.FLAG_KEEP_INTERMEDIATES)
    out = s.forward(x)
    torch.cuda.synchronize()
    ks = s.kernels(B)
    print(f"HEP_LATE={late}: {len(ks)} launches; late launches: {[n for n, _b, _f, y in ks if 'late' in y]}")
    res[late] = dict(blocks={i: s.stage(f"block{i}", B).float().cpu() for i in range(10, 16)}, heads=[t.float().cpu() for t in out[1:]])
    s.close()
    s2 = Session(sd, phi, size, B, "bf16")
    total, per = s2.profile(B, 20, per_kernel=True)
    names = [n for n, _b, _f, _y in s2.kernels(B)]
    print(f"  one batch {total * 1e3:.1f} us; " + ", ".join(f"{n} {t * 1e3:.1f}" for n, t in zip(names, per) if n.startswith(("b11", "b12", "b13", "b14", "b15"))))
    s2.close()
for i in range(10, 16):
    a, b = res["0"]["blocks"][i], res["1"]["blocks"][i]
    d = (a - b).abs()
    print(f"block{i}: max|d| {d.max().item():.3e} (max|ref| {a.abs().max().item():.3e}), mean|d| {d.mean().item():.3e} (mean|ref| {a.abs().mean().item():.3e}), finite {bool(torch.isfinite(b).all())}")
for n, a, b in zip(("regression", "classification", "rotation", "translation", "hand"), res["0"]["heads"], res["1"]["heads"]):
    print(f"{n}: mean|d|/mean|ref| {((a - b).abs().mean() / a.abs().mean()).item():.3e}")
