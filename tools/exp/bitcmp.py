# bit-compare two builds of the library on the benchmark shapes (bf16): python /tmp/bitcmp.py libA libB
import sys, os, subprocess, json
code = r'''
import sys, torch, numpy as np, hashlib
sys.path.insert(0, ".")
from hmd_ego_pose_amd.model import Session
from hmd_ego_pose_amd.weights import seeded_state_dict
out = {}
for phi, size, batch, prec in ((0, 256, 16, "bf16"), (0, 256, 3, "bf16"), (3, 512, 2, "bf16"), (0, 384, 2, "bf16"), (0, 128, 5, "bf16"), (0, 256, 16, "fp32")):
    sd = seeded_state_dict(phi, 6)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(batch, 3, size, size, generator=g).cuda()
    s = Session(sd, phi, size, batch, prec)
    o = s.forward(x)[1:]
    torch.cuda.synchronize()
    out[f"{phi}-{size}-{batch}-{prec}"] = [hashlib.sha1(t.cpu().numpy().tobytes()).hexdigest()[:12] for t in o]
    s.close()
import json; print(json.dumps(out))
'''
res = []
for lib in sys.argv[1:3]:
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HEP_LIB=os.path.abspath(lib)), capture_output=True, text=True)
    res.append(json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else r.stderr[-500:])
for k in res[0]:
    print(k, "IDENTICAL" if res[0][k] == res[1][k] else f"DIFFER {res[0][k]} {res[1][k]}")
