import sys, ctypes; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hmd_ego_pose_amd import _capi
_capi.LIB_PATH = os.environ.get("HEP_TRACE_LIB") or os.path.join(os.path.dirname(_capi.LIB_PATH), "libhep_trace.so")     # make -C hmd_ego_pose_amd/csrc trace
from hmd_ego_pose_amd.model import Session
from hmd_ego_pose_amd.weights import seeded_state_dict
B=int(sys.argv[1]); nblocks=int(sys.argv[2])
s = Session(seeded_state_dict(0,0), 0, 256, B, sys.argv[3] if len(sys.argv) > 3 else "bf16")   # HEP_MBF_TRACE_SEL="Cexp,H" picks the layer
x = torch.randn(B,3,256,256, device="cuda")
for _ in range(3): s.forward(x, want_features=False)
torch.cuda.synchronize()
l = _capi.lib()
f = l.hep_dbg_mbf_trace; f.restype = ctypes.c_int; f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
f(None, 0, 1)
s.forward(x, want_features=False); torch.cuda.synchronize()
nw = nblocks*8
buf = np.zeros((nw, 8), np.uint64)
f(buf.ctypes.data, nw, 0)
t = buf[:, :7].astype(np.int64)
ok = t[:, 0] > 0
t = t[ok]
t0 = t[:, 0].min()
rel = (t - t0) * 10e-3
names = ["start", "A_issued", "A_barrier", "B_done", "B_barrier", "C_done", "end"]
print("waves", len(t), "span us", rel[:, 6].max())
d = np.diff(rel, axis=1)
for i in range(6):
    print(f"phase {names[i]:>10s} -> {names[i+1]:10s} mean {d[:, i].mean():6.2f} p50 {np.percentile(d[:, i], 50):6.2f} p90 {np.percentile(d[:, i], 90):6.2f}")
print("block life mean", (rel[:, 6] - rel[:, 0]).mean())
h, e = np.histogram(rel[:, 0], bins=10); print("start hist", list(zip(np.round(e[:-1], 1), h)))
