"""Does one submission thread per stream shorten a SHORT timed run (the driver's 20 steps)?  After a device synchronize the first
submission costs the host 130-150 us (tools/exp/short_run_anatomy.py); with one Python thread the other three streams wait behind it.
Interleaved: single-thread loop against four persistent worker threads (ctypes releases the GIL inside the C call), 20- and 200-step
windows.  GPU box: python tools/exp/threaded_submit.py [reps]"""
import os, sys, time, ctypes, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from hmd_ego_pose_amd import _capi
from hmd_ego_pose_amd.model import Session
from hmd_ego_pose_amd.weights import seeded_state_dict
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 6
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

def single(K):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(K): step(i)
    torch.cuda.synchronize()
    return time.perf_counter() - t0

class Pool:
    def __init__(self):
        self.go = [threading.Semaphore(0) for _ in range(D)]; self.done = threading.Semaphore(0); self.K = 0; self.stop = False
        self.th = [threading.Thread(target=self.run, args=(d,), daemon=True) for d in range(D)]
        for t in self.th: t.start()
    def run(self, d):
        torch.cuda.set_device(dev)
        while True:
            self.go[d].acquire()
            if self.stop: return
            for i in range(d, self.K, D): step(i)
            self.done.release()
    def timed(self, K):
        self.K = K
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for g in self.go: g.release()
        for _ in range(D): self.done.acquire()
        torch.cuda.synchronize()
        return time.perf_counter() - t0
pool = Pool()
for K in (20, 200):
    a, b = [], []
    for _ in range(REPS):
        a.append(B * K / single(K)); b.append(B * K / pool.timed(K))
    print(f"steps {K}: one submission thread {np.median(a):.0f} frames/s (min {min(a):.0f} max {max(a):.0f}); one thread per stream {np.median(b):.0f} (min {min(b):.0f} max {max(b):.0f})")
