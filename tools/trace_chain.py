import sys, ctypes, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hmd_ego_pose_amd import _capi
_capi.LIB_PATH = os.path.join(os.path.dirname(_capi.LIB_PATH), "libhep_trace.so")     # make -C hmd_ego_pose_amd/csrc trace
from hmd_ego_pose_amd.model import Session
from hmd_ego_pose_amd.weights import seeded_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
phi = int(sys.argv[2]) if len(sys.argv) > 2 else 0; size = int(sys.argv[3]) if len(sys.argv) > 3 else 256
s = Session(seeded_state_dict(phi, 0), phi, size, B, "bf16")
x = torch.randn(B, 3, size, size, device="cuda")
for _ in range(3): s.forward(x, want_features=False)
torch.cuda.synchronize()
l = _capi.lib()
f = l.hep_dbg_chain_trace; f.restype = ctypes.c_int; f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
f(None, 0, 1)
s.forward(x, want_features=False); torch.cuda.synchronize()       # the LAST chain launch (3 nodes) leaves its stamps
W = 16                                                             # waves per workgroup: every wave stamps
buf = np.zeros((B, W, 64), np.uint64); f(buf.ctypes.data, B, 0)
n = int(buf[0, 0, 63]); t = buf[:, :, :n].astype(np.int64); rel = (t - t[:, :1, :1]) * 10e-3       # relative to wave 0's first stamp
print("stamps per wave", n, "; mean us since kernel start, wave 0:"); print(np.round(rel[:, 0].mean(axis=0), 2))
d = np.diff(rel, axis=2).mean(axis=0)                              # [wave][phase]
print("phase durations (us), rows = waves 0..15:")
np.set_printoptions(linewidth=250, suppress=True)
print(np.round(d, 2))
print("slowest wave per phase:", np.round(d.max(axis=0), 2))
