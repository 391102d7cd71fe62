import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from hmd_ego_pose_amd import _capi
from hmd_ego_pose_amd.model import Session
from hmd_ego_pose_amd.weights import seeded_state_dict
from tests._util import seeded_input
sd = seeded_state_dict(0, 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 3
sync = len(sys.argv) > 2
x = torch.from_numpy(seeded_input((B, 3, 256, 256), 0))
s0 = Session(sd, 0, 256, B, "bf16", flags=_capi.FLAG_KEEP_INTERMEDIATES)
ref = [t.clone() for t in s0.forward(x.cuda())[1:]]
want12 = s0.stage("block12", B).float().cpu()
s0.close()
os.environ["HEP_LATE"] = "1"; os.environ["HEP_LATE_G"] = "3"
s = Session(sd, 0, 256, B, "bf16", flags=_capi.FLAG_KEEP_INTERMEDIATES)
outs = []
for rep in range(4):
    o = [t.clone() for t in s.forward(x.cuda())[1:]]
    if sync: torch.cuda.synchronize()
    b12 = s.stage("block12", B).float().cpu()
    outs.append(o)

    print(f"rep {rep}: block12 vs launch-by-launch max {float((b12 - want12).abs().max()):.3g}; heads vs launch-by-launch mean rel {[round(float((a - r).abs().mean() / r.abs().mean()), 4) for a, r in zip(o, ref)]}; vs rep 0 unequal {[int((a != b).sum()) for a, b in zip(o, outs[0])]}", flush=True)
s.close()
