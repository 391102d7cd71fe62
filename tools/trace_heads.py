"""Per-wave phase stamps of heads_kernel (k_heads.hip) from the profiling build (make -C hmd_ego_pose_amd/csrc trace).
usage: python tools/trace_heads.py [batch]"""
import sys, ctypes, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hmd_ego_pose_amd import _capi
_capi.LIB_PATH = os.path.join(os.path.dirname(_capi.LIB_PATH), "libhep_trace.so")
from hmd_ego_pose_amd.model import Session
from hmd_ego_pose_amd.weights import seeded_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
os.environ.setdefault("HEP_HEADS_FUSED", "1")
s = Session(seeded_state_dict(0, 0), 0, 256, B, "bf16")
x = torch.randn(B, 3, 256, 256, device="cuda")
for _ in range(3): s.forward(x, want_features=False)
torch.cuda.synchronize()
f = _capi.lib().hep_dbg_heads_trace; f.restype = ctypes.c_int; f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
f(None, 0, 1)
s.forward(x, want_features=False); torch.cuda.synchronize()
NI = 40
buf = np.zeros((NI, 16, 16), np.uint64)
f(buf.ctypes.data, buf.size, 0)
t = buf.astype(np.int64)
t0 = t[t > 0].min()
names = ["load", "layer0", "layer1", "layer2", "-", "hdr dw", "hdr"]
print(f"image 0 of {B}: kernel span {(t[:, :, 7].max() - t0) * 0.01:.1f} us")
for i in range(NI):
    w = t[i]
    if w[0, 0] == 0: continue
    st = (w[:, 0].min() - t0) * 0.01
    ph = [(w[:, 1] - w[:, 0]).mean(), (w[:, 2] - w[:, 1]).mean(), (w[:, 3] - w[:, 2]).mean(), (w[:, 4] - w[:, 3]).mean(), (w[:, 6] - w[:, 4]).mean(), (w[:, 7] - w[:, 6]).mean()]
    print(f"item {i:2d}: start {st:6.1f}  load {ph[0]*0.01:5.2f}  layers {ph[1]*0.01:5.2f} {ph[2]*0.01:5.2f} {ph[3]*0.01:5.2f}  header dw {ph[4]*0.01:5.2f}  header {ph[5]*0.01:5.2f}  total {(w[:, 7].max() - w[:, 0].min())*0.01:6.2f}")
