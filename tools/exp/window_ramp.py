"""Frames/s of consecutive short windows right after session set-up (is a 20-step window after 5 warm-up steps measured on a GPU that
has not reached its steady clocks yet?), and of windows separated by idle gaps.  GPU box: python tools/exp/window_ramp.py [steps] [precision]"""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hmd_ego_pose_amd import _capi
from hmd_ego_pose_amd.model import Session
from hmd_ego_pose_amd.weights import seeded_state_dict
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
D, B, S = 4, 16, 256
dev = torch.device("cuda", 0)
sd = seeded_state_dict(0, 0)
lib = _capi.lib()
xs = [torch.randn(B, 3, S, S, device=dev) for _ in range(D)]
cam = torch.tensor([[480, 480, 128, 128, 1000, 1.0]] * B, dtype=torch.float32, device=dev)
strides = (ctypes.c_int64 * 4)(*xs[0].stride())
streams = [torch.cuda.Stream(dev) for _ in range(D)]
sess = [Session(sd, 0, S, B, prec, dev) for _ in range(D)]
N = sess[0].num_anchors
boxes = [torch.empty((B, N, 4), device=dev) for _ in range(D)]; trans = [torch.empty((B, N, 3), device=dev) for _ in range(D)]
def step(i):
    d = i % D; st = streams[d].cuda_stream
    _capi.check(lib.hep_run_device(sess[d].handle, xs[d].data_ptr(), strides, B, None, None, st))
    _capi.check(lib.hep_decode_device(sess[d].handle, None, None, cam.data_ptr(), B, boxes[d].data_ptr(), trans[d].data_ptr(), st))
def window(k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(k): step(i)
    torch.cuda.synchronize(); return B * k / (time.perf_counter() - t0)
for d in range(D): step(d)            # graph capture
torch.cuda.synchronize()
t_begin = time.perf_counter()
out = []
for w in range(60):
    out.append((round((time.perf_counter() - t_begin) * 1e3, 1), round(window(K))))
print(f"{prec}: consecutive {K}-step windows right after set-up (ms since set-up, frames/s):")
print(" ", out[:12]); print("  ...", out[-6:])
for gap in (0.002, 0.01, 0.05, 0.2, 1.0):
    r = []
    for _ in range(6):
        time.sleep(gap); r.append(round(window(K)))
    print(f"windows after {gap*1e3:.0f} ms of idle: {r}")
r = []
for _ in range(6):
    time.sleep(0.2); window(5); r.append(round(window(K)))
print(f"windows after 200 ms of idle + 5 warm-up steps: {r}")
