# per-wave phase timeline of the single-node BiFPN launches (sep_kernel, mode 0) on the level of side <hw> (profiling build:
# make -C hmd_ego_pose_amd/csrc trace).  The LAST such launch of the forward is the one read back.
# usage (GPU box): python tools/trace_sep.py <batch> <hw> [phi size precision]
import sys, ctypes, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hmd_ego_pose_amd import _capi
_capi.LIB_PATH = os.path.join(os.path.dirname(_capi.LIB_PATH), "libhep_trace.so")
from hmd_ego_pose_amd.model import Session
from hmd_ego_pose_amd.weights import seeded_state_dict
B = int(sys.argv[1]); hw = int(sys.argv[2])
phi = int(sys.argv[3]) if len(sys.argv) > 3 else 0; size = int(sys.argv[4]) if len(sys.argv) > 4 else 256; prec = sys.argv[5] if len(sys.argv) > 5 else "bf16"
s = Session(seeded_state_dict(phi, 0), phi, size, B, prec)
x = torch.randn(B, 3, size, size, device="cuda")
for _ in range(3): s.forward(x, want_features=False)
torch.cuda.synchronize()
l = _capi.lib()
f = l.hep_dbg_sep_trace; f.restype = ctypes.c_int; f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
f(None, 0, hw)
s.forward(x, want_features=False); torch.cuda.synchronize()
nw = 1 << 16
buf = np.zeros((nw, 8), np.uint64); f(buf.ctypes.data, nw, hw)
t = buf.astype(np.int64); t = t[t[:, 0] > 0]
rel = (t - t[:, 0].min()) * 10e-3
names = ["start", "gather_done", "barrier1", "dw_done", "barrier2(weights parked)", "mfma_done", "barrier3", "stores_acked"]
print("waves", len(t), "span us", rel[:, 7].max())
d = np.diff(rel, axis=1)
for i in range(7): print(f"phase {names[i]:>26s} -> {names[i+1]:26s} mean {d[:, i].mean():6.2f} p50 {np.percentile(d[:, i], 50):6.2f} p90 {np.percentile(d[:, i], 90):6.2f}")
print("wave life mean", (rel[:, 7] - rel[:, 0]).mean())
h, e = np.histogram(rel[:, 0], bins=8); print("start hist", list(zip(np.round(e[:-1], 1), h)))
