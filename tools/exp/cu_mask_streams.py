"""Four batches in flight on four streams that each own a disjoint quarter of the CUs (hipExtStreamCreateWithCUMask) against four
ordinary streams.  CU-mask bit i belongs to XCD i % 8 (the mask is dealt round-robin over the XCDs), so
  xcd    stream d owns the XCDs 2 d and 2 d + 1 (its own two L2s),
  slice  stream d owns a quarter of every XCD's CUs.
GPU box: python tools/exp/cu_mask_streams.py [precision] [seconds]"""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hmd_ego_pose_amd import _capi
from hmd_ego_pose_amd.model import Session
from hmd_ego_pose_amd.weights import seeded_state_dict
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
D, B, S, NCU, NXCD = 4, 16, 256, 256, 8
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
hip = ctypes.CDLL("libamdhip64.so")
sd = seeded_state_dict(0, 0)
lib = _capi.lib()
xs = [torch.randn(B, 3, S, S, device=dev) for _ in range(D)]
cam = torch.tensor([[480, 480, 128, 128, 1000, 1.0]] * B, dtype=torch.float32, device=dev)
strides = (ctypes.c_int64 * 4)(*xs[0].stride())
sess = [Session(sd, 0, S, B, prec, dev) for _ in range(D)]
N = sess[0].num_anchors
boxes = [torch.empty((B, N, 4), device=dev) for _ in range(D)]; trans = [torch.empty((B, N, 3), device=dev) for _ in range(D)]

def masked_stream(bits):
    words = (ctypes.c_uint32 * (NCU // 32))()
    for i in bits:
        words[i // 32] |= 1 << (i % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), NCU // 32, words)
    assert rc == 0, rc
    return st.value

def streams_for(mode):
    if mode == "none":
        return [torch.cuda.Stream(dev).cuda_stream for _ in range(D)]
    if mode == "xcd":
        return [masked_stream([i for i in range(NCU) if (i % NXCD) // 2 == d]) for d in range(D)]
    if mode == "slice":
        return [masked_stream([i for i in range(NCU) if (i // NXCD) % D == d]) for d in range(D)]
    if mode == "half":      # two streams per half of the chip (XCDs 0-3 / 4-7)
        return [masked_stream([i for i in range(NCU) if (i % NXCD) // 4 == d % 2]) for d in range(D)]
    raise ValueError(mode)

def run(streams, seconds):
    def window(k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(k):
            d = i % D; st = streams[d]
            _capi.check(lib.hep_run_device(sess[d].handle, xs[d].data_ptr(), strides, B, None, None, st))
            _capi.check(lib.hep_decode_device(sess[d].handle, None, None, cam.data_ptr(), B, boxes[d].data_ptr(), trans[d].data_ptr(), st))
        torch.cuda.synchronize(); return time.perf_counter() - t0
    window(200)
    k, t = 0, 0.0
    while t < seconds:
        t += window(400); k += 400
    return B * k / t

ref = None
for rep in range(1):           # (one pass: masked streams keep their hardware queues until destroyed, and more than four queues in use
    for mode in ("none", "xcd", "slice", "half"):      #  is a known loss by itself - NOTEBOOK.md section 11, hardware queues)
        st = streams_for(mode)
        v = run(st, secs)
        if mode != "none":
            for h_ in st:
                hip.hipStreamDestroy(ctypes.c_void_p(h_))
        if mode == "none" and ref is None:
            ref = [t.clone() for t in boxes]
        ok = all(torch.equal(a, b) for a, b in zip(ref, boxes))
        print(f"{prec} {mode:6s} {v:9.0f} frames/s   outputs identical to ordinary streams: {ok}", flush=True)
