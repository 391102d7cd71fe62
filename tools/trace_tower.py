# per-wave phase timeline of the LAST tower launch (profiling build: make -C hmd_ego_pose_amd/csrc trace)
# usage (GPU box): python tools/trace_tower.py <batch> <waves to read> [phi size precision]    the last map layer; HEP_TOWER_TRACE_HDR=1: the headers launch
# (sessions whose head layers run as tower_coop_kernel print that kernel's eight stamps)
import sys, ctypes, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hmd_ego_pose_amd import _capi
_capi.LIB_PATH = os.path.join(os.path.dirname(_capi.LIB_PATH), "libhep_trace.so")
from hmd_ego_pose_amd.model import Session
from hmd_ego_pose_amd.weights import seeded_state_dict
B = int(sys.argv[1]); nw = int(sys.argv[2])
phi = int(sys.argv[3]) if len(sys.argv) > 3 else 0; size = int(sys.argv[4]) if len(sys.argv) > 4 else 256; prec = sys.argv[5] if len(sys.argv) > 5 else "bf16"
s = Session(seeded_state_dict(phi, 0), phi, size, B, prec)
x = torch.randn(B, 3, size, size, device="cuda")
for _ in range(3): s.forward(x, want_features=False)
torch.cuda.synchronize()
l = _capi.lib()
f = l.hep_dbg_tower_trace; f.restype = ctypes.c_int; f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
f(None, 0, 1 if os.environ.get('HEP_TOWER_TRACE_HDR') else 2)
s.forward(x, want_features=False); torch.cuda.synchronize()
buf = np.zeros((nw, 8), np.uint64); n = f(buf.ctypes.data, nw, 0)
coop = any("tower_coop_kernel" in k[3] for k in s.kernels(B))
nc = 8 if coop else 7
t = buf[:, :nc].astype(np.int64); t = t[t[:, 0] > 0]
rel = (t - t[:, 0].min()) * 10e-3
names = ["start", "descriptor", "issued", "barrier", "dw_done", "mfma_stores_issued", "stores_acked"]
if coop: names = ["start", "issued", "halo_parked", "barrier1", "dw_done", "barrier2", "image0_done", "all_done"]
print("coop" if coop else "wave-per-patch", "waves", len(t), "span us", rel[:, nc - 1].max())
d = np.diff(rel, axis=1)
for i in range(nc - 1): print(f"phase {names[i]:>18s} -> {names[i+1]:18s} mean {d[:, i].mean():6.2f} p50 {np.percentile(d[:, i], 50):6.2f} p90 {np.percentile(d[:, i], 90):6.2f}")
print("wave life mean", (rel[:, nc - 1] - rel[:, 0]).mean())
h, e = np.histogram(rel[:, 0], bins=8); print("start hist", list(zip(np.round(e[:-1], 1), h)))
