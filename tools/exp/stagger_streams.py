"""Do the four streams run better when they are a quarter of a forward apart?  The loop submits batches round-robin, so the streams start
within ~0.15 ms of each other and stay that way (equal work per stream): all four are in the big-map launches at the same time, then all
four in the fronts, then all four in the heads.  Here the first batch of stream k is held back by k * X us (a host-side wait before its
first submission; everything after that is the normal loop).  GPU box: python tools/exp/stagger_streams.py [steps]"""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from hmd_ego_pose_amd import _capi
from hmd_ego_pose_amd.model import Session
from hmd_ego_pose_amd.weights import seeded_state_dict
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
D, B, S = 4, 16, 256
dev = torch.device("cuda", 0)
sd = seeded_state_dict(0, 0)
lib = _capi.lib()
xs = [torch.randn(B, 3, S, S, device=dev) for _ in range(D)]
cam = torch.tensor([[480, 480, 128, 128, 1000, 1.0]] * B, dtype=torch.float32, device=dev)
strides = (ctypes.c_int64 * 4)(*xs[0].stride())
streams = [torch.cuda.Stream(dev) for _ in range(D)]
sess = [Session(sd, 0, S, B, "bf16", dev) for _ in range(D)]
N = sess[0].num_anchors
boxes = [torch.empty((B, N, 4), device=dev) for _ in range(D)]; trans = [torch.empty((B, N, 3), device=dev) for _ in range(D)]
def step(i):
    d = i % D; st = streams[d].cuda_stream
    _capi.check(lib.hep_run_device(sess[d].handle, xs[d].data_ptr(), strides, B, None, None, st))
    _capi.check(lib.hep_decode_device(sess[d].handle, None, None, cam.data_ptr(), B, boxes[d].data_ptr(), trans[d].data_ptr(), st))
for i in range(40): step(i)
torch.cuda.synchronize()
def run(X):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(K):
        if 0 < i < D and X > 0:
            t = time.perf_counter()
            while (time.perf_counter() - t) * 1e6 < X: pass
        step(i)
    torch.cuda.synchronize()
    return B * K / (time.perf_counter() - t0)
for rep in range(3):
    print("  ".join(f"X={X:3d}us {run(X):7.0f}" for X in (0, 75, 150, 300, 0, 150, 300, 450)), flush=True)
