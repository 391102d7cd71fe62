"""Where the fixed cost of a short timed run goes (bench.py --steps 20 reads ~5 % below the sustained figure): host enqueue time per
step, time to the first kernel, and the spread of the streams' finishing times.  GPU box: python tools/exp/short_run_anatomy.py [steps]"""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from hmd_ego_pose_amd import _capi
from hmd_ego_pose_amd.model import Session
from hmd_ego_pose_amd.weights import seeded_state_dict
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
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
calls = []
def step(i):
    d = i % D; st = streams[d].cuda_stream
    a = time.perf_counter()
    _capi.check(lib.hep_run_device(sess[d].handle, xs[d].data_ptr(), strides, B, None, None, st))
    b = time.perf_counter()
    _capi.check(lib.hep_decode_device(sess[d].handle, None, None, cam.data_ptr(), B, boxes[d].data_ptr(), trans[d].data_ptr(), st))
    calls.append((b - a, time.perf_counter() - b))
for i in range(40): step(i)
torch.cuda.synchronize()
for rep in range(3):
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(D)]; ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(D)]
    base = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if os.environ.get("ANATOMY_SPIN"):          # experiment: how long does the FIRST HIP call of any kind take after the synchronize?
        q0 = time.perf_counter(); streams[3].query(); print("first stream query after the synchronize: %.0f us" % ((time.perf_counter() - q0) * 1e6))
    base.record(streams[0])
    marks = []; calls.clear()
    for i in range(K):
        if i < D: ev0[i].record(streams[i])
        step(i)
        marks.append(time.perf_counter() - t0)
    for d in range(D): ev1[d].record(streams[d])
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize(); t_all = time.perf_counter() - t0
    starts = [base.elapsed_time(e) for e in ev0]; ends = [base.elapsed_time(e) for e in ev1]
    print("first six steps, host us (forward call, decode call):", [(round(a * 1e6), round(b * 1e6)) for a, b in calls[:6]])
    print(f"steps {K}: host total {t_all*1e3:.3f} ms ({B*K/t_all:.0f} frames/s); enqueue of all steps {t_enq*1e3:.3f} ms, first step returned after {marks[0]*1e3:.3f} ms, "
          f"4th after {marks[3]*1e3:.3f} ms; stream start offsets (ms) {[round(s, 3) for s in starts]}; stream end times {[round(e, 3) for e in ends]}")
