import sys, ctypes, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hmd_ego_pose_amd import _capi
_capi.LIB_PATH = os.path.join(os.path.dirname(_capi.LIB_PATH), "libhep_trace.so")     # make -C hmd_ego_pose_amd/csrc trace
from hmd_ego_pose_amd.model import Session
from hmd_ego_pose_amd.weights import seeded_state_dict
B = 16
s = Session(seeded_state_dict(0, 0), 0, 256, B, "bf16")
x = torch.randn(B, 3, 256, 256, device="cuda")
for _ in range(3): s.forward(x, want_features=False)
torch.cuda.synchronize()
l = _capi.lib()
f = l.hep_dbg_chain_trace; f.restype = ctypes.c_int; f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
f(None, 0, 1)
s.forward(x, want_features=False); torch.cuda.synchronize()       # the LAST chain launch (3 nodes) leaves its stamps
buf = np.zeros((B, 64), np.uint64); f(buf.ctypes.data, B, 0)
n = int(buf[0, 63]); t = buf[:, :n].astype(np.int64); rel = (t - t[:, :1]) * 10e-3
print("stamps per block", n, "; mean us since kernel start:"); print(np.round(rel.mean(axis=0), 2))
print("deltas:", np.round(np.diff(rel.mean(axis=0)), 2))
print("shader clock during the kernel: %.2f GHz" % float((buf[:, 62].astype(np.float64) / ((t[:, -1] - t[:, 0]) * 10.0)).mean()))
