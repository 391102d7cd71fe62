#!/usr/bin/env python3
"""Determinism soak: D sessions of one precision in flight on D streams, the same frames every step; every result must be bit-identical to
the first one (no atomics on the data path, fixed reduction orders).  Catches load-dependent races - e.g. a missing wait behind an LDS-DMA
transfer shows up only when other streams keep the memory system busy.
    python tools/soak.py [precision] [phi] [size] [batch] [steps]      (environment knobs such as HEP_CHAIN_STREAM apply)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hmd_ego_pose_amd.model import Session
from hmd_ego_pose_amd.weights import seeded_state_dict
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
phi = int(sys.argv[2]) if len(sys.argv) > 2 else 0
size = int(sys.argv[3]) if len(sys.argv) > 3 else 256
B = int(sys.argv[4]) if len(sys.argv) > 4 else 16
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 2000
D = 4
sd = seeded_state_dict(phi, 0)
sess = [Session(sd, phi, size, B, prec) for _ in range(D)]
streams = [torch.cuda.Stream() for _ in range(D)]
x = torch.from_numpy(np.random.Generator(np.random.PCG64(7)).standard_normal((B, 3, size, size)).astype(np.float32)).cuda()
torch.cuda.synchronize()
ref, bad = None, 0
for i in range(steps):
    d = i % D
    with torch.cuda.stream(streams[d]):
        out = sess[d].forward(x, want_features=False)[1:]
        if i % 50 == d or i < D:                       # clone a result of every slot now and then (on its stream)
            snap = [o.clone() for o in out]
            streams[d].synchronize()
            if ref is None:
                ref = snap
            elif not all(torch.equal(a, b) for a, b in zip(snap, ref)):
                bad += 1
                print(f"step {i} slot {d}: result differs, max |diff| {max(float((a - b).abs().max()) for a, b in zip(snap, ref)):.3e}")
torch.cuda.synchronize()
assert all(torch.isfinite(o).all() for o in ref)
print(f"soak {prec} phi {phi} @ {size} b{B}: {steps} steps, {D} in flight, {bad} mismatching snapshots")
sys.exit(1 if bad else 0)
