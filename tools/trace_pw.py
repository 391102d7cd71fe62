# per-wave phase timeline of one pointwise GEMM launch (profiling build: make -C hmd_ego_pose_amd/csrc trace)
# usage (GPU box): HEP_PW_TRACE_SEL="K,N" python tools/trace_pw.py <batch> <workgroups to read>
import sys, ctypes, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hmd_ego_pose_amd import _capi
_capi.LIB_PATH = os.path.join(os.path.dirname(_capi.LIB_PATH), "libhep_trace.so")
from hmd_ego_pose_amd.model import Session
from hmd_ego_pose_amd.weights import seeded_state_dict
B = int(sys.argv[1]); nblocks = int(sys.argv[2])
l = _capi.lib()
f = l.hep_dbg_pw_trace; f.restype = ctypes.c_int; f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
f(None, 0, 1)                                     # before the session: the plan's graph captures the buffer pointer
s = Session(seeded_state_dict(0, 0), 0, 256, B, sys.argv[3] if len(sys.argv) > 3 else "bf16")
x = torch.randn(B, 3, 256, 256, device="cuda")
for _ in range(3): s.forward(x, want_features=False)
torch.cuda.synchronize()
nw = nblocks * 4
buf = np.zeros((nw, 8), np.uint64); f(buf.ctypes.data, nw, 1)
t = buf[:, :6].astype(np.int64); t = t[t[:, 0] > 0]
rel = (t - t[:, 0].min()) * 10e-3
names = ["start", "issued", "se_done", "k_done", "reduced", "end"]
print("waves", len(t), "span us", rel[:, 5].max())
d = np.diff(rel, axis=1)
for i in range(5): print(f"phase {names[i]:>8s} -> {names[i+1]:8s} mean {d[:, i].mean():6.2f} p50 {np.percentile(d[:, i], 50):6.2f} p90 {np.percentile(d[:, i], 90):6.2f}")
print("wave life mean", (rel[:, 5] - rel[:, 0]).mean(), "start spread", rel[:, 0].max())
