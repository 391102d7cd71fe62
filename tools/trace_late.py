"""Per-wave phase stamps of the image-resident late-block kernel (k_late.hip) from the profiling build (make -C hmd_ego_pose_amd/csrc trace).
usage: HEP_LATE=1 python tools/trace_late.py [batch]"""
import sys, ctypes, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hmd_ego_pose_amd import _capi
_capi.LIB_PATH = os.path.join(os.path.dirname(_capi.LIB_PATH), "libhep_trace.so")
from hmd_ego_pose_amd.model import Session
from hmd_ego_pose_amd.weights import seeded_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
os.environ.setdefault("HEP_LATE", "1")
s = Session(seeded_state_dict(0, 0), 0, 256, B, "bf16")
x = torch.randn(B, 3, 256, 256, device="cuda")
for _ in range(3): s.forward(x, want_features=False)
torch.cuda.synchronize()
f = _capi.lib().hep_dbg_late_trace; f.restype = ctypes.c_int; f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
f(None, 0, 1)
s.forward(x, want_features=False); torch.cuda.synchronize()
NB = 6
buf = np.zeros((B, 16, NB, 64), np.uint64)
f(buf.ctypes.data, buf.size, 0)
t = buf.astype(np.int64)
t0 = t[t > 0].min()
us = lambda a: (a - t0) * 0.01
nblk = int((t[0, 0, :, 0] > 0).sum())
print(f"{B} workgroups, {nblk} blocks; kernel span {us(t[:, :, :, 5].max()):.1f} us")
for bi in range(nblk):
    tb = t[:, :, bi, :]
    def span(a, b, waves=slice(None)):
        x0, x1 = tb[:, waves, a], tb[:, waves, b]
        ok = (x0 > 0) & (x1 > 0)
        return ((x1 - x0)[ok] * 0.01).mean() if ok.any() else float("nan")
    nc = int((tb[0, 0, 8:32] > 0).sum())
    print(f"block {bi}: start {us(tb[:, :, 0].min()):.1f}  prologue {span(0, 1):.2f}  chunk loop {span(1, 2):.2f}  squeeze-excite {span(2, 3):.2f}  park As {span(3, 4):.2f}  "
          f"project {span(4, 5):.2f} (K loop {span(4, 6):.2f}, partials {span(6, 7):.2f}, epilogue {span(7, 5):.2f})  total {span(0, 5):.2f}")
    ch = [span(8 + c - 1 if c else 1, 8 + c) for c in range(nc)]
    print("   chunks: " + " ".join(f"{v:.2f}" for v in ch))
    mm, dw = slice(8, 16), slice(0, 8)
    print(f"   chunk 1, mm waves: expand {span(32, 33, mm):.2f}  se partial {span(33, 34, mm):.2f}  park {span(34, 35, mm):.2f}  wait {span(35, 9, mm):.2f}")
    print(f"   chunk 1, dw waves: taps {span(40, 41, dw):.2f}  swish+store {span(41, 42, dw):.2f}  sums {span(42, 43, dw):.2f}  wait {span(43, 9, dw):.2f}")
