#!/usr/bin/env python3
"""Every phi x input size x precision x a few batch sizes: two forwards of the same frames must be finite and bit-identical, and (fp32) a
frame's result must not depend on its batch position.  Meant for the sanitizer build (NaN-poisoned LDS, guard bands behind every tensor):
    HEP_LIB=$PWD/hmd_ego_pose_amd/libhep_poison.so python tools/sweep_shapes.py [max_phi]
A read of an LDS cell nobody wrote, or of an activation cell nobody wrote, comes out as NaN; a store past the end of a tensor aborts at
hep_destroy with the tensor's name."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hmd_ego_pose_amd.model import Session
from hmd_ego_pose_amd.weights import seeded_state_dict
max_phi = int(sys.argv[1]) if len(sys.argv) > 1 else 6
bad = 0
for phi in range(max_phi + 1):
    sd = seeded_state_dict(phi, 0)
    for size in (128, 256, 384, 512):
        for prec in ("fp32", "bf16"):
            for B in (1, 3, 16 if size <= 256 else 5):
                rng = np.random.Generator(np.random.PCG64(phi * 1000 + size + B))
                x = torch.from_numpy(rng.standard_normal((B, 3, size, size)).astype(np.float32)).cuda()
                s = Session(sd, phi, size, B, prec)
                a = [t.clone() for t in s.forward(x, want_features=True)[1:]]
                b = s.forward(x, want_features=True)[1:]
                torch.cuda.synchronize()
                ok = all(torch.isfinite(t).all() for t in a) and all(torch.equal(u, v) for u, v in zip(a, b))
                if ok and B > 1:
                    c = s.forward(torch.flip(x, dims=[0]), want_features=False)[1:]
                    ok = all(torch.equal(torch.flip(u, dims=[0]), v) for u, v in zip(a, c))
                s.close()
                if not ok:
                    bad += 1
                    print(f"FAIL phi {phi} size {size} {prec} batch {B}")
    print(f"phi {phi} done", flush=True)
print(f"sweep: {bad} failing configurations")
sys.exit(1 if bad else 0)
