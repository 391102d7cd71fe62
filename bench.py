#!/usr/bin/env python3
"""bench.py - frames/sec of the MI355X EfficientPose path at 256x256, batch 16 per GPU, phi 0.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

With --gpus N > 1 and no torchrun environment the script starts its N rank processes itself (fresh
children, started before anything touches a GPU; the parent supervises them and never initialises HIP).

One step = one pass of the hot path over one batch of 16 synthetic, HBM-resident frames per
GPU: the full forward (stem .. heads, one hipGraph replay) + box/translation decode, through
the C ABI of libhep.so.  --inflight D (default 4) keeps D steps in flight on D HIP streams (D
sessions): a single forward is a dependent chain of small kernels and leaves most of the
256 CUs idle; `one_batch_in_flight` reports the strictly sequential number as well.  Weak scaling:
every rank owns 16 frames, the forward needs no collective (frames are independent); ranks meet
only at the timing barriers.  Rank 0 prints ONE JSON line.  It also carries
  sustained     the same loop run for >= 2 s: frames/s and min / median / max over 10 sub-windows
  fp32          (N=1) the same two figures for an fp32 session - the precision that meets the 0.1 mm ADD bound
                (no bf16 storage boundary does on the seeded weights)
  comm          the serving loop around the same step with the data movement SURVEY 8(e) describes:
                rank 0 owns the global batch of uint8 frames and scatters 16 to every rank (point to point
                over RCCL/xGMI), each rank runs preprocess -> forward -> decode -> detection filter, and the
                post-filter rows are gathered on rank 0; --comm-depth batches in flight per GPU (default 4, as the main loop)
  roofline      (N=1) the dominant device function: algorithmic bytes per launch / its in-sequence
                launch duration measured here with HIP events, against 8 TB/s HBM3E; `layers` lists every launch
  cpu_baseline  (N=1) the CPU oracle (torch fp32 restatement of the reference) timed on the host
                cores in the evaluate.py regime (batch 1, anchors rebuilt per call, decode).
  latency_b1    (N=1) the only regime the reference publishes a number for (unity-sandbox/WebRTCNetCoreSandbox/Program.cs:24-33:
                "effnet_b0_512", FP32, batch 1, prep + inference 175 / 40 / 16 ms on the ONNXRuntime CPU / CUDA / TensorRT providers,
                + 6-8 ms preprocessing): phi 0 at 512x512, fp32, batch 1, called through the C ABI the way the C# host would -
                p50 / p99 over >= 200 back-to-back calls, the same call paced at 60 Hz, and the model-load time (hep_create).
The timed loop is fed by one submission thread per stream (joined before the closing barrier; a thread that raises fails the run):
after a synchronize the first graph launch costs the host ~140 us, and with one Python thread the other streams wait behind it.
`single_thread_value` is the same loop submitted from one thread.
Before the --warmup steps every loop keeps the device busy with the same step for --preheat-ms (default 50, reported as `preheat_ms`;
untimed set-up): an MI355X that has had no work for >= 10 ms runs its next ~20 ms of launches up to 8 % slower
(tools/exp/window_ramp.py), and session set-up ends with such a gap.  The timed region itself is unchanged: exactly --steps steps
between barrier + synchronize on both sides.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "frames/sec at 256x256 bs16 EfficientPose-phi0; ADD(-S) vs ref"
HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec); ~6.3 TB/s achievable
PRECISIONS = ["bf16", "fp32", "fp8"]
CLS_BIAS_KEY = "classifier.header.pointwise_conv.conv.bias"


def cpu_baseline(phi, size, budget_s=15.0):
    """The oracle, on the host cores, exactly in the reference's evaluate.py regime: one frame per
    call, anchors regenerated every call (train.py:36), box + translation decode; no_grad is used
    (the reference does not) - see BASELINE.md section 4."""
    import numpy as np
    import torch
    from hmd_ego_pose_amd.weights import seeded_state_dict
    from oracle import decode_ref as D
    from oracle import efficientpose_ref as R
    sd = seeded_state_dict(phi, 0)
    rng = np.random.Generator(np.random.PCG64(0))
    x = torch.from_numpy(rng.standard_normal((1, 3, size, size)).astype(np.float32))
    cam = np.array([[480, 480, 128, 128, 1000, 1.0]], np.float32)

    def one():
        _, reg, cls, rot, trn, hand = R.forward(sd, x, phi)
        anchors, t_anchors = D.anchors_for_size(size)
        D.decode_boxes(anchors, reg.numpy(), size)
        D.decode_translation(t_anchors, trn.numpy(), cam)

    # the box has far more CPUs than these small convolutions can use: take the fastest thread count
    best = None
    for nt in (8, 16, 32, 64, 128):
        if nt > (os.cpu_count() or 1):
            break
        torch.set_num_threads(nt)
        one()
        t0 = time.perf_counter(); one(); one(); dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, nt)
    torch.set_num_threads(best[1])
    for _ in range(3):
        one()
    n, t0 = 0, time.perf_counter()
    while True:
        one(); n += 1
        el = time.perf_counter() - t0
        if (el >= budget_s and n >= 20) or n >= 2000:
            break
    return {"value": round(n / el, 3), "unit": "frames/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "sample": f"{n} frames, batch 1 per call (evaluate.py regime), phi {phi} {size}x{size} fp32, "
                      f"forward + anchors + box/translation decode, {el:.1f} s, best of 8..128 torch threads on {os.cpu_count()} host CPUs"}


def add_vs_ref(phi, size, precisions=("fp32", "bf16", "fp8")):
    """The accuracy half of the metric ("ADD(-S) vs ref"), part of the CPU-baseline leg because it needs the oracle: for
    a few seeded frames the oracle (the reference's arithmetic, fp32) and the HIP path predict a pose at the same anchor
    (the oracle's best-scoring one); ADD / ADD-S between the two poses over a 1000-point cloud of the drill's size
    (hep_pose_errors), in the translation unit (mm).  No dataset or checkpoint ships with the reference, so this is the
    distance to the reference's OUTPUT on synthetic weights, not an accuracy against ground truth; random-init networks
    amplify rounding (16 significant bits everywhere are needed for 0.1 mm), so the bf16 / fp8
    figures are upper bounds for trained weights."""
    import math
    import numpy as np
    import torch
    from hmd_ego_pose_amd.evaluate import pose_errors
    from hmd_ego_pose_amd.model import Session
    from hmd_ego_pose_amd.weights import seeded_state_dict
    from oracle import decode_ref as D
    from oracle import efficientpose_ref as R
    sd = seeded_state_dict(phi, 0)
    nf = 4
    rng = np.random.Generator(np.random.PCG64(99))
    x = torch.from_numpy(rng.standard_normal((nf, 3, size, size)).astype(np.float32))
    cam = np.array([[480, 480, 128, 128, 1000, 1.0]] * nf, np.float32)
    pts = (rng.standard_normal((1000, 3)) * np.array([40.0, 25.0, 60.0])).astype(np.float32)
    _, reg, cls, rot, trn, hand = R.forward(sd, x, phi)
    _, t_anchors = D.anchors_for_size(size)
    t_ref = D.decode_translation(t_anchors, trn.numpy(), cam)
    idx = cls[:, :, 0].argmax(dim=1).numpy()
    pick = lambda a: np.stack([a[i, idx[i]] for i in range(nf)])
    out = {}
    from hmd_ego_pose_amd._capi import HepUnsupported
    done = []
    for prec in precisions:
        try:
            s_ = Session(sd, phi, size, nf, prec)
        except HepUnsupported:        # fp8 is an opt-in build (make FP8=1)
            continue
        done.append(prec)
        _, g_reg, _g_cls, g_rot, g_trn, _g_hand = s_.forward(x.cuda(), want_features=False)
        _, g_t = s_.decode(g_reg, g_trn, torch.from_numpy(cam).cuda())
        add, add_s = pose_errors(pts, pick(rot.numpy()) * math.pi, pick(t_ref), pick(g_rot.cpu().numpy()) * math.pi, pick(g_t.cpu().numpy()))
        out[prec] = {"add_mm": round(float(add.mean()), 5), "add_s_mm": round(float(add_s.mean()), 5)}
        s_.close()
    out["bound_mm"] = 0.1
    out["meets_bound"] = [p for p in done if out[p]["add_mm"] <= 0.1]
    out["sample"] = f"{nf} seeded frames, pose at the oracle's best-scoring anchor, 1000-point cloud (sigma 40/25/60 mm), translations ~ N(0, 1) * 1000 mm"
    return out


def latency_b1(dev, calls=300):
    """The reference's only published performance regime (unity-sandbox/WebRTCNetCoreSandbox/Program.cs:24-33: "effnet_b0_512", FP32,
    batch 1; preparation of a frame 6-8 ms, prep + inference 175 ms / 40 ms / 16 ms on the ONNXRuntime CPU / CUDA / TensorRT providers
    of an RTX 3090 + Ryzen 3900X box, model load < 10 s / ~1 min / ~10 min), through the C ABI as the C# host would call it:
      host  : hep_run (host float[1,3,512,512] in, the five head arrays out - what replaces Session.Run, Program.cs:211-229)
              -> hep_decode -> hep_filter, everything in host memory, one call after the other, each returning when its result is there;
      frame : the app's frame callback (Program.cs:140-205) with the frame bytes in host memory: H2D copy of one 1280x720 I420 frame ->
              hep_preprocess_i420_device (YV12 -> BGR, centre crop 256, resize 512, normalise) -> hep_run_device -> hep_decode_device ->
              hep_filter_device -> D2H copy of the detection rows -> stream synchronize.
    Wall-clock per call, p50 / p99 over `calls` calls after 20 warm-up calls; the classifier header's bias is shifted so that ~30 of the
    49 104 anchors pass the 0.5 threshold (a trained network's rate; the seeded classifier passes ~40 %, which would time a pathological
    sort + NMS) - the same load knob as the comm loop's, every other tensor and the arithmetic are the parity-tested ones."""
    import numpy as np
    import torch
    from hmd_ego_pose_amd import _capi
    from hmd_ego_pose_amd.model import Session
    from hmd_ego_pose_amd.weights import pack_bytes, seeded_state_dict
    lib = _capi.lib()
    phi, S, M = 0, 512, 100
    sd = seeded_state_dict(phi, 0)
    blob = pack_bytes(sd)
    t0 = time.perf_counter()
    h = ctypes.c_void_p()
    _capi.check(lib.hep_create_from_memory(blob, len(blob), phi, S, 1, _capi.HEP_F32, dev.index or 0, 0, ctypes.byref(h)))
    load_ms = (time.perf_counter() - t0) * 1e3          # weight pack parse + BatchNorm fold + layout + upload + arena (the graph is captured by the first run)
    lib.hep_destroy(h)
    rng = np.random.Generator(np.random.PCG64(5))
    x = rng.standard_normal((1, 3, S, S)).astype(np.float32)
    # classifier bias shift: the quantile of the logits that leaves ~30 candidates, found in a few rounds (the top scores of the seeded
    # network saturate to 1.0 in fp32, so one round cannot see how far above the threshold they sit)
    shift = 0.0
    for _ in range(6):
        sd_l = dict(sd); sd_l[CLS_BIAS_KEY] = sd[CLS_BIAS_KEY] - shift
        s = Session(sd_l, phi, S, 1, "fp32", dev)
        p = s.forward(torch.from_numpy(x).to(dev), want_features=False)[2].double().flatten()
        n_pass = int((p > 0.5).sum())
        if 10 <= n_pass <= 60:
            break
        pc = p.clamp(1e-6, 1 - 1e-6)
        shift += float(torch.quantile(torch.log(pc / (1 - pc)), 1.0 - 30.0 / p.numel()))
        s.close()
    N = s.num_anchors
    cam = np.array([[480, 480, 128, 128, 1000, 1.0]], np.float32)
    ho = [np.empty((1, N, k), np.float32) for k in s.out_width]
    hb, ht = np.empty((1, N, 4), np.float32), np.empty((1, N, 3), np.float32)
    hd_ = [np.empty((1, M, 4), np.float32), np.empty((1, M), np.float32), np.empty((1, M), np.int32), np.empty((1, M, 3), np.float32),
           np.empty((1, M, 3), np.float32), np.empty((1, M, 63), np.float32), np.empty((1, M), np.int32), np.empty((1,), np.int32)]
    t0 = time.perf_counter()
    _capi.check(lib.hep_run(s.handle, x.ctypes.data, 1, None, *[o.ctypes.data for o in ho]))
    first_ms = (time.perf_counter() - t0) * 1e3          # first host-ABI call of the handle (staging buffers)
    thr = 0.5

    def host_call():
        _capi.check(lib.hep_run(s.handle, x.ctypes.data, 1, None, *[o.ctypes.data for o in ho]))
        _capi.check(lib.hep_decode(s.handle, ho[0].ctypes.data, ho[3].ctypes.data, cam.ctypes.data, 1, hb.ctypes.data, ht.ctypes.data))
        _capi.check(lib.hep_filter(s.handle, hb.ctypes.data, ho[1].ctypes.data, ho[2].ctypes.data, ht.ctypes.data, ho[4].ctypes.data, 1, thr, 0.5, M,
                                   *[a.ctypes.data for a in hd_]))

    FH, FW = 720, 1280
    frame = torch.from_numpy(rng.integers(0, 256, (1, FH * FW * 3 // 2), dtype=np.uint8))
    st = torch.cuda.Stream(dev)
    cam_d = torch.from_numpy(cam).to(dev)
    pre = torch.empty((1, S, S, 3), dtype=torch.float32, device=dev)
    xv = pre.permute(0, 3, 1, 2)
    xstr = (ctypes.c_int64 * 4)(*xv.stride())
    bx, tr = torch.empty((1, N, 4), device=dev), torch.empty((1, N, 3), device=dev)
    f = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
    i = lambda *sh: torch.empty(sh, dtype=torch.int32, device=dev)
    det = [f(1, M, 4), f(1, M), i(1, M), f(1, M, 3), f(1, M, 3), f(1, M, 63), i(1, M), i(1)]
    counts = []

    def frame_call():
        with torch.cuda.stream(st):
            fd = frame.to(dev, non_blocking=True)
            _capi.check(lib.hep_preprocess_i420_device(s.handle, fd.data_ptr(), 1, FH, FW, 256, 512, pre.data_ptr(), st.cuda_stream))
            _capi.check(lib.hep_run_device(s.handle, xv.data_ptr(), xstr, 1, None, None, st.cuda_stream))
            _capi.check(lib.hep_decode_device(s.handle, None, None, cam_d.data_ptr(), 1, bx.data_ptr(), tr.data_ptr(), st.cuda_stream))
            _capi.check(lib.hep_filter_device(s.handle, bx.data_ptr(), None, None, tr.data_ptr(), None, 1, thr, 0.5, M, *[d.data_ptr() for d in det], st.cuda_stream))
            rows = [d.cpu() for d in det]                # D2H of the detection rows (synchronises the stream)
        counts.append(int(rows[7][0]))

    def pct(fn):
        for _ in range(20):
            fn()
        torch.cuda.synchronize(dev)
        ts = []
        for _ in range(calls):
            t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        return {"p50_ms": round(ts[len(ts) // 2], 4), "p99_ms": round(ts[min(len(ts) - 1, int(len(ts) * 0.99))], 4), "min_ms": round(ts[0], 4), "mean_ms": round(sum(ts) / len(ts), 4)}

    def paced(fn, hz, n):
        """the same call once per 1 / hz seconds (a camera's frame period): the device is idle between frames, and an MI355X that has had no
        work for >= 10 ms runs its next launches slower (tools/exp/window_ramp.py) - what a 60 Hz frame source sees"""
        ts, period = [], 1.0 / hz
        nxt = time.perf_counter() + period
        for _ in range(n):
            while time.perf_counter() < nxt:
                time.sleep(max(0.0, min(0.002, nxt - time.perf_counter())))
            nxt += period
            t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        return {"p50_ms": round(ts[len(ts) // 2], 4), "p99_ms": round(ts[min(len(ts) - 1, int(len(ts) * 0.99))], 4), "min_ms": round(ts[0], 4), "calls": n, "hz": hz}

    host = pct(host_call)
    n_host = int(hd_[7][0])
    fr = pct(frame_call)
    fr60 = paced(frame_call, 60.0, 90)
    s.close()
    return {"host": host, "frame": fr, "frame_paced_60hz": fr60, "calls": calls, "model_load_ms": round(load_ms, 1), "first_call_ms": round(first_ms, 1),
            "detections_per_frame": {"host": n_host, "frame": counts[-1] if counts else None}, "score_threshold": thr, "classifier_bias_shift": round(-shift, 3),
            "config": "EfficientPose phi=0 512x512 fp32 batch=1 (the reference's \"effnet_b0_512\", FP32), seeded random-init weights, N(0,1) input / random I420 frame bytes",
            "what": "host: hep_run -> hep_decode -> hep_filter on host arrays (in: 3 MB, out: 14.5 MB of raw heads per call over PCIe, as the C# host consumes them); "
                    "frame: 1280x720 I420 bytes in host memory -> H2D -> hep_preprocess_i420_device -> hep_run_device -> hep_decode_device -> hep_filter_device -> D2H of the <= 100 detection rows",
            "reference_published": {"prep_ms": "6-8", "prep_plus_inference_ms": {"onnxruntime_cpu": 175, "onnxruntime_cuda": 40, "onnxruntime_tensorrt": 16},
                                    "model_load": {"cpu": "< 10 s", "cuda": "~1 min", "tensorrt": "~10 min"},
                                    "hardware": "RTX 3090 + Ryzen 3900X, Windows, with Unity running", "source": "unity-sandbox/WebRTCNetCoreSandbox/Program.cs:24-33"}}


def pmc_traffic(symbol, tag=""):
    """HBM bytes per launch of `symbol` from the newest committed PMC pass of this configuration
    (profiles/r*/*_pmc_per_kernel.json for the default workload, *_phi3_pmc_per_kernel.json for phi 3 @ 512 b8, *_fp32_pmc_per_kernel.json for fp32 sessions:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, KB per launch).  gfx950 correction from
    /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE counts 128-byte requests as 64 bytes, so reads are
    doubled; WRITE_SIZE is exact for 16-byte stores.  None when no pass holds the kernel."""
    import glob
    def tag_of(path):
        b = os.path.basename(path)
        return "phi3_" if b.endswith("_phi3_pmc_per_kernel.json") else ("fp32_" if b.endswith("_fp32_pmc_per_kernel.json") else "")
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "*_pmc_per_kernel.json")), reverse=True):
        if tag_of(f) != tag:
            continue
        try:
            d = json.load(open(f))
            rd, wr = d["FETCH_SIZE"].get(symbol), d["WRITE_SIZE"].get(symbol)
            if rd and wr:
                return {"bytes_per_launch": round((2.0 * rd["KB_per_launch"] + wr["KB_per_launch"]) * 1024),
                        "read_bytes": round(2.0 * rd["KB_per_launch"] * 1024), "write_bytes": round(wr["KB_per_launch"] * 1024),
                        "source": os.path.relpath(f, ROOT)}
        except (OSError, KeyError, ValueError):
            continue
    return None


def launch_profile(sess, B, ms_step, ms_one_batch, traffic_of=None, layers=True):
    """Per-launch durations of one session (one batch in flight: the only regime in which a launch can be timed alone) and the
    roofline block built from them.  Durations come from an eager pass with a HIP event in front of every launch; the event
    pairs add a constant to each launch, calibrated live: the same launches replayed as one hipGraph (no events) take
    total_ms, so the per-launch overhead is (sum(eager) - total_ms) / n."""
    total_ms, per = sess.profile(B, 20, per_kernel=True)
    ks = sess.kernels(B)
    ev_overhead = max(0.0, (sum(per) - total_ms) / len(per))
    per = [max(t - ev_overhead, 1e-6) for t in per]
    agg = {}
    for (name, nbytes, flops, sym), t in zip(ks, per):
        a = agg.setdefault(sym, [0.0, 0.0, 0.0, 0])
        a[0] += t; a[1] += nbytes; a[2] += flops; a[3] += 1
    sym, (t, nbytes, flops, calls) = max(agg.items(), key=lambda kv: kv[1][0])
    achieved = nbytes / (t * 1e-3) / 1e9
    step_bytes = sum(k[1] for k in ks)
    fracs = sorted(k[1] / (t_ * 1e-3) / 1e9 / HBM_PEAK_GBS for k, t_ in zip(ks, per))
    out = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
           "traffic": None, "traffic_ratio": None, "traffic_detail": traffic_of(sym) if traffic_of else None,
           "kernel": sym, "launches_per_step": calls, "avg_launch_us": round(t / calls * 1e3, 2),
           "algorithmic_bytes_per_launch": round(nbytes / calls), "share_of_step": round(t / sum(per), 3),
           "measured_with_batches_in_flight": 1,
           "graph_replay_ms": round(total_ms, 4), "event_overhead_us_subtracted": round(ev_overhead * 1e3, 2),
           # the north-star target is ">= 0.60 of the per-layer roofline": how many launches are there, and where the step is as a whole
           "layers_total": len(ks), "layers_at_or_above_0p6": sum(f >= 0.6 for f in fracs),
           "best_layer_frac": round(fracs[-1], 4), "median_layer_frac": round(fracs[len(fracs) // 2], 4),
           "time_weighted_frac": round(step_bytes / (sum(per) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           "whole_step_algorithmic_GBps": round(step_bytes / (ms_step * 1e-3) / 1e9, 1),
           "end_to_end_frac": round(step_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           "end_to_end_frac_one_batch": round(step_bytes / (ms_one_batch * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    if out["traffic_detail"]:      # flat scalars (the driver's record keeps scalars only): HBM bytes per launch from the PMC pass, and over the algorithmic bytes
        out["traffic"] = out["traffic_detail"]["bytes_per_launch"]
        out["traffic_ratio"] = round(out["traffic"] / max(1.0, nbytes / calls), 3)
    # context: the next device functions by total time, same definitions
    out["top"] = [{"kernel": k, "launches_per_step": v[3], "avg_launch_us": round(v[0] / v[3] * 1e3, 2), "share_of_step": round(v[0] / sum(per), 3),
                   "achieved": round(v[1] / (v[0] * 1e-3) / 1e9, 1), "frac": round(v[1] / (v[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                  for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:5]]
    if layers:
        # every launch, stand-alone in sequence: name, device function, us, algorithmic MB, GB/s, fraction of 8 TB/s
        out["layers"] = [[name, sym_, round(t_ * 1e3, 2), round(nb / 1e6, 3), round(nb / (t_ * 1e-3) / 1e9, 1), round(nb / (t_ * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)]
                         for (name, nb, _fl, sym_), t_ in zip(ks, per)]
        out["layers_columns"] = ["launch", "device_function", "us", "algorithmic_MB", "GB/s", "frac_of_8TB/s"]
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=16, help="frames per GPU per step")
    ap.add_argument("--phi", type=int, default=0)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--precision", default="bf16", choices=PRECISIONS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-comm", action="store_true", help="skip the scatter -> preprocess -> forward -> filter -> gather loop")
    ap.add_argument("--no-fp32", action="store_true", help="skip the fp32 (accuracy-bound-meeting) throughput block")
    ap.add_argument("--no-layers", action="store_true", help="omit roofline.layers (one row per launch)")
    ap.add_argument("--no-latency", action="store_true", help="skip the latency_b1 block (phi 0 @ 512 fp32 batch 1 through the host C ABI)")
    ap.add_argument("--submit-threads", type=int, default=1, choices=[0, 1], help="1: one submission thread per stream feeds the timed loop (default); 0: the single-thread loop")
    ap.add_argument("--preheat-ms", type=float, default=50.0, help="untimed set-up before the warm-up steps: the same step for this long, so that the device has left its idle power state (0: none)")
    ap.add_argument("--sustain-seconds", type=float, default=2.0, help="length of the `sustained` run (0: skip)")
    ap.add_argument("--comm-score-threshold", type=float, default=0.5)
    ap.add_argument("--comm-depth", type=int, default=4, help="batches in flight in the serving (comm) loop, at most --inflight")
    ap.add_argument("--comm-candidates", type=float, default=30.0, help="mean candidates per frame the comm loop's classifier bias is set for")
    ap.add_argument("--inflight", type=int, default=4, help="batches in flight per GPU (sessions on separate HIP streams)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for wiring tests)")
    ap.add_argument("--single-device", action="store_true", help="wiring test: every rank uses cuda:0 (needs --backend gloo)")
    return ap.parse_args(argv)


def visible_gpu_count():
    """GPUs this process would see, WITHOUT initialising HIP (the supervising parent of a --gpus N run must never do that:
    it forks N rank processes next).  Counts the KFD topology nodes that have SIMDs (/sys/class/kfd: CPUs are nodes with
    simd_count 0) and applies HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES; None when the topology cannot be read (the
    ranks then check for themselves before the rendezvous)."""
    import glob
    n = 0
    files = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not files:
        return None
    for f in files:
        try:
            for line in open(f):
                k, _, v = line.partition(" ")
                if k == "simd_count" and int(v) > 0:
                    n += 1
        except (OSError, ValueError):
            return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([t for t in v.split(",") if t.strip() != ""]))
    return n


def self_launch(args):
    """--gpus N without a torchrun environment: start N rank processes of this script (one per GPU, env as
    torchrun sets it), supervise them and relay rank 0's line.  This parent never initialises a GPU and never
    exec()s.  A rank that dies takes the others with it (they would otherwise sit in the rendezvous / a barrier
    until the collective time-out)."""
    ndev = visible_gpu_count()                # sysfs only: nothing here may initialise HIP / HSA before the ranks are forked
    if ndev is None:                          # no KFD topology at all (a host without the amdgpu driver): torch's own count
        import torch
        ndev = torch.cuda.device_count()
    if not args.single_device and ndev < args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but only {ndev} GPU(s) visible\n")
        return 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs, out0 = [], tempfile.TemporaryFile()
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad:
            rc = max(abs(c) for c in bad) or 1
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t_end = time.time() + 10
            for p in procs:
                try:
                    p.wait(max(0.1, t_end - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.05)
    out0.seek(0)
    sys.stdout.write(out0.read().decode())
    sys.stdout.flush()
    return rc


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))

    import numpy as np
    import torch
    from hmd_ego_pose_amd import _capi, dist as hd
    from hmd_ego_pose_amd.model import Session
    from hmd_ego_pose_amd.weights import seeded_state_dict

    if args.single_device:
        os.environ["LOCAL_RANK"] = "0"
    rank0, local0, world0 = hd.env_world()
    if world0 != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world0}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if local0 >= torch.cuda.device_count():       # before the rendezvous: the other ranks are taken down by the launcher instead of waiting
        raise SystemExit(f"rank {rank0}: local rank {local0} but only {torch.cuda.device_count()} GPU(s) visible")
    rank, local_rank, world = hd.init(args.backend, timeout_s=180)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    B, S, phi = args.batch, args.size, args.phi
    if world > 1:
        sys.stderr.write("bench.py " + hd.describe(local_rank) + "\n"); sys.stderr.flush()      # before the first collective (the weight broadcast below)

    # weights: synthetic (no checkpoint ships with the reference), rank 0's copy broadcast once over RCCL
    sd = hd.broadcast_state_dict(seeded_state_dict(phi, 0), dev)
    D = max(1, args.inflight)
    # D sessions = D batches in flight on D HIP streams: the forward of one batch is a dependent chain of
    # small kernels that cannot fill 256 CUs, so a serving loop keeps several batches in flight
    # (step i runs on slot i % D; every step is a full forward + decode of its own 16 frames)
    lib = _capi.lib()
    rng = np.random.Generator(np.random.PCG64(1000 + rank))
    xs = [torch.from_numpy(rng.standard_normal((B, 3, S, S)).astype(np.float32)).to(dev) for _ in range(D)]    # resident in HBM
    cam = torch.tensor([[480, 480, 128, 128, 1000, 1.0]] * B, dtype=torch.float32, device=dev)
    strides = (ctypes.c_int64 * 4)(*xs[0].stride())
    streams = [torch.cuda.Stream(dev) for _ in range(D)]

    class Submitters:
        """One persistent submission thread per stream: thread d enqueues the steps d, d + D, d + 2 D, ... of a window on stream d
        (ctypes releases the GIL inside the C call).  submit() returns when every thread has enqueued its share; an exception in
        a thread is re-raised there, a thread that does not come back within two minutes fails the run instead of hanging it."""
        def __init__(self, step):
            self.step = step
            self.go = [threading.Semaphore(0) for _ in range(D)]
            self.done = threading.Semaphore(0)
            self.K, self.err, self.stop = 0, None, False
            self.th = [threading.Thread(target=self._run, args=(d,), daemon=True) for d in range(D)]
            for t in self.th:
                t.start()

        def _run(self, d):
            torch.cuda.set_device(dev)
            while True:
                self.go[d].acquire()
                if self.stop:
                    return
                try:
                    for i in range(d, self.K, D):
                        self.step(i)
                except BaseException as e:      # noqa: BLE001  (handed to the submitting thread)
                    self.err = e
                finally:
                    self.done.release()

        def submit(self, K):
            self.K = K
            for g in self.go:
                g.release()
            for _ in range(D):
                if not self.done.acquire(timeout=120):
                    raise RuntimeError("bench.py: a submission thread did not return within 120 s")
            if self.err is not None:
                raise self.err

        def close(self):
            self.stop = True
            for g in self.go:
                g.release()
            for t in self.th:
                t.join(5)

    class Loop:
        """D sessions of one precision and the step they run."""
        def __init__(self, precision):
            self.sess = [Session(sd, phi, S, B, precision, dev) for _ in range(D)]
            N = self.sess[0].num_anchors
            self.boxes = [torch.empty((B, N, 4), dtype=torch.float32, device=dev) for _ in range(D)]
            self.trans = [torch.empty((B, N, 3), dtype=torch.float32, device=dev) for _ in range(D)]
            torch.cuda.synchronize(dev)
            for d in range(D):            # set-up, not measurement: every session captures its hipGraph on first use
                self.step(d)
            torch.cuda.synchronize(dev)
            self.pool = Submitters(self.step) if (args.submit_threads and D > 1) else None

        def run(self, steps, depth=D, threads=True):
            """enqueue `steps` steps: from the per-stream submission threads when all D streams are in use, else from this thread"""
            if self.pool is not None and threads and depth == D:
                self.pool.submit(steps)
            else:
                for i in range(steps):
                    self.step(i, depth)

        def step(self, i, depth=D):
            # forward into the handle's own output buffers (no copies), then decode from them
            d = i % depth
            st = streams[d].cuda_stream
            _capi.check(lib.hep_run_device(self.sess[d].handle, xs[d].data_ptr(), strides, B, None, None, st))
            _capi.check(lib.hep_decode_device(self.sess[d].handle, None, None, cam.data_ptr(), B, self.boxes[d].data_ptr(), self.trans[d].data_ptr(), st))

        def preheat(self, seconds):
            """Untimed set-up: keep the device busy with the same step for `seconds`.  After >= 10 ms without work an MI355X runs the next
            ~20 ms of launches 8 % slower (tools/exp/window_ramp.py, profiles/r06/q_window_ramp_*.txt: consecutive 20-step windows right
            after set-up 49.9k, 51.6k, 52.6k, 53.3k frames/s ...; 5 warm-up steps = 1.5 ms do not cover it)."""
            t0 = time.perf_counter()
            while seconds > 0 and time.perf_counter() - t0 < seconds:
                self.run(4 * D)
                torch.cuda.synchronize(dev)

        def timed(self, steps, depth=D, threads=True):
            torch.cuda.synchronize(dev); t0 = time.perf_counter()
            self.run(steps, depth, threads)
            torch.cuda.synchronize(dev)
            return time.perf_counter() - t0

        def close(self):
            if self.pool is not None:
                self.pool.close(); self.pool = None
            for s_ in self.sess:
                s_.close()

    main_loop = Loop(args.precision)
    N = main_loop.sess[0].num_anchors
    hd.barrier()
    main_loop.preheat(args.preheat_ms / 1e3)      # device out of its idle power state (reported as `preheat_ms`), THEN the --warmup steps
    main_loop.run(args.warmup)
    hd.barrier(); torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    main_loop.run(args.steps)           # EXACTLY --steps steps: submission thread d enqueues steps d, d + D, ... on stream d and is joined here
    torch.cuda.synchronize(dev); hd.barrier()
    elapsed = hd.max_over_ranks(time.perf_counter() - t0, dev)
    # the same window submitted from ONE thread (what `value` was until round 5): a side figure
    single_elapsed = None
    if main_loop.pool is not None:
        main_loop.preheat(args.preheat_ms / 1e3)
        main_loop.timed(args.warmup, threads=False)
        single_elapsed = hd.max_over_ranks(main_loop.timed(args.steps, threads=False), dev)
    assert all(torch.isfinite(t).all() for t in main_loop.boxes) and all(torch.isfinite(t).all() for t in main_loop.trans)

    # ---- the same loop for >= sustain_seconds, in 10 sub-windows (a 20-step run is 7 ms: this is the figure to trust) ----
    sustained = None
    if args.sustain_seconds > 0:
        per = max(args.steps, int(args.sustain_seconds / 10 / max(elapsed / args.steps, 1e-6)) + 1)
        wins = []
        hd.barrier()
        for _ in range(10):
            wins.append(hd.max_over_ranks(main_loop.timed(per), dev))
        fps = sorted(B * world * per / w for w in wins)
        sustained = {"value": round(B * world * per * 10 / sum(wins), 2), "unit": "frames/s", "seconds": round(sum(wins), 3), "steps": per * 10,
                     "windows": 10, "min": round(fps[0], 2), "median": round((fps[4] + fps[5]) / 2, 2), "max": round(fps[-1], 2)}

    # ---- the same step inside a serving loop with its data movement: scatter of uint8 frames from
    #      rank 0, preprocess, forward, decode, detection filter, gather of the post-filter rows on rank 0; CD
    #      batches in flight per GPU (slot i % CD: own session, own stream), so the scatter of step i + 1 and the
    #      gather of step i - 1 overlap the compute of step i (measured at N = 1: 2 / 3 / 4 in flight 36.6k / 42.3k /
    #      46.5k frames/s) ----
    comm = None
    if not args.no_comm:
        G = B * world
        M = 100
        frames_u8 = torch.from_numpy(np.random.Generator(np.random.PCG64(7)).integers(0, 256, (G, S, S, 3), dtype=np.uint8)).to(dev) if rank == 0 else None
        CD = max(1, min(args.comm_depth, D))          # batches in flight in the serving loop (slot i % CD: own session, own stream)
        cstreams = [torch.cuda.Stream(dev) for _ in range(CD)]

        hd.warm_up_p2p(dev)       # RCCL sets its point-to-point communicators up lazily: not inside the timed loop

        def serve_loop(sessions, steps):
            got = [None] * CD
            views = [s_.output_views() for s_ in sessions]      # the handles' own head buffers: the forward below copies nothing

            def serve(i):
                d = i % CD
                with torch.cuda.stream(cstreams[d]):
                    mine = hd.scatter_frames(frames_u8, G, (S, S, 3), dev, dtype=torch.uint8)       # 196 KB per frame instead of 786 KB fp32
                    x = sessions[d].preprocess(mine)                                              # NCHW view of normalised NHWC memory
                    _capi.check(lib.hep_run_device(sessions[d].handle, x.data_ptr(), (ctypes.c_int64 * 4)(*x.stride()), x.shape[0], None, None, cstreams[d].cuda_stream))
                    reg, cls, rot, trn, hand = (v[:x.shape[0]] for v in views[d])
                    bx, tr = sessions[d].decode(reg, trn, cam)
                    det = sessions[d].filter(bx, cls, rot, tr, hand, args.comm_score_threshold, 0.5, M)
                    got[d] = hd.gather_detections(det, G)
            for i in range(2 * CD):
                serve(i)
            # the device out of its idle power state, as for the main loop: the same number of steps on every rank (scatter and gather
            # are collective), sized from the main loop's step time (`elapsed` is the maximum over ranks, identical everywhere)
            for i in range(min(256, int(args.preheat_ms / max(elapsed / args.steps * 1e3, 1e-3)) + 1) if args.preheat_ms > 0 else 0):
                serve(i)
            torch.cuda.synchronize(dev); hd.barrier()
            t1 = time.perf_counter()
            for i in range(steps):
                serve(i)
            torch.cuda.synchronize(dev); hd.barrier()
            return hd.max_over_ranks(time.perf_counter() - t1, dev), got[(steps - 1) % CD]

        # realistic candidate rate: a trained classifier passes a handful of the 12 276 anchors; the seeded one passes ~40 %
        # (scores straddle 0.5), which times a pathological sort + NMS.  The comm loop therefore shifts the classifier
        # header's bias so that ~comm_candidates anchors per frame pass the threshold (quantile of the seeded logits on this
        # loop's own frames); the unshifted weights are reported as `pathological`.
        k2 = max(40, args.steps)          # (40 steps = 13 ms: a 10-step window scattered by +-4 %)
        x_probe = main_loop.sess[0].preprocess(frames_u8[:B] if rank == 0 else torch.zeros((B, S, S, 3), dtype=torch.uint8, device=dev))
        p = main_loop.sess[0].forward(x_probe, want_features=False)[2].float().clamp(1e-7, 1 - 1e-7)
        logit_thr = float(np.log(args.comm_score_threshold / (1 - args.comm_score_threshold)))
        q = torch.quantile(torch.log(p / (1 - p)).flatten()[:1 << 24], 1.0 - args.comm_candidates / N).item() - logit_thr
        qt = torch.tensor([q], dtype=torch.float64, device=dev)
        if world > 1:
            import torch.distributed as td
            td.broadcast(qt, 0)
        sd_comm = dict(sd); sd_comm[CLS_BIAS_KEY] = sd[CLS_BIAS_KEY] - float(qt.item())
        csess = [Session(sd_comm, phi, S, B, args.precision, dev) for _ in range(CD)]
        e2, got = serve_loop(csess, k2)
        e3, got_p = serve_loop(main_loop.sess[:CD], k2)
        for s_ in csess:
            s_.close()
        if rank == 0:
            assert got["count"].shape[0] == G and got["boxes"].shape == (G, M, 4)
            row_bytes = M * (4 + 1 + 1 + 3 + 3 + 63 + 1) * 4 + 4
            comm = {"value": round(G * k2 / e2, 2), "unit": "frames/s", "ms_per_step": round(e2 / k2 * 1e3, 4), "steps": k2, "batches_in_flight": CD,
                    "what": f"{CD} batches in flight per GPU (slot i % {CD}: own session and stream): scatter uint8 frames from rank 0 (point to point) -> preprocess -> forward -> decode -> "
                            f"filter (score > {args.comm_score_threshold}, NMS 0.5, top {M}) -> gather detection rows on rank 0; classifier header bias shifted by "
                            f"{-float(qt.item()):.3f} so that ~{args.comm_candidates:g} anchors per frame pass the threshold (a trained network's rate) - "
                            f"this bias-shifted session is a LOAD knob, not the parity-tested weight set (every other tensor and the arithmetic are the same; `pathological` runs the unshifted, parity-tested weights)",
                    "scatter_bytes_per_step": int((G - B) * S * S * 3), "gather_bytes_per_step": int((G - B) * row_bytes),
                    "backend": args.backend if world > 1 else None, "ranks": world,
                    "mean_detections_per_frame": round(float(got["count"].float().mean()), 1),
                    "pathological": {"value": round(G * k2 / e3, 2), "ms_per_step": round(e3 / k2 * 1e3, 4),
                                     "mean_detections_per_frame": round(float(got_p["count"].float().mean()), 1),
                                     "what": "the same loop with the unshifted seeded weights: ~40 % of the anchors pass the threshold"}}

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        out = {
            "metric": METRIC if (B, S, phi) == (16, 256, 0) else f"frames/sec at {S}x{S} bs{B} EfficientPose-phi{phi}; ADD(-S) vs ref", "value": round(B * world * args.steps / elapsed, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "preheat_ms": args.preheat_ms, "ms_per_step": round(ms, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"EfficientPose phi={phi} {S}x{S} {args.precision} batch={B} per GPU: forward (stem, MBConv, BiFPN, 5 heads) "
                                   f"+ box/translation decode; seeded random-init weights, N(0,1) frames resident in HBM; "
                                   f"{D} batches of {B} in flight per GPU on {D} HIP streams",
                       "phi": phi, "size": S, "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}",
                       "batches_in_flight": D, "anchors": N, "launches_per_step": len(main_loop.sess[0].kernels(B)) + 1,
                       "submission_threads": D if main_loop.pool is not None else 1},
        }
        if single_elapsed is not None:
            out["single_thread_value"] = round(B * world * args.steps / single_elapsed, 2)
        if sustained is not None:
            out["sustained"] = sustained
        if comm is not None:
            out["comm"] = comm
        if world == 1:
            # the strictly sequential number (one batch in flight): latency of a step
            k1 = max(50, args.steps // 2)
            main_loop.timed(10, 1)
            e1 = main_loop.timed(k1, 1)
            out["one_batch_in_flight"] = {"value": round(B * k1 / e1, 2), "ms_per_step": round(e1 / k1 * 1e3, 4), "steps": k1}
            if comm is not None:
                comm["vs_one_batch_in_flight"] = round(comm["value"] / out["one_batch_in_flight"]["value"], 3)
            # per-launch durations and the roofline block (launch_profile above)
            s0 = main_loop.sess[0]
            cfg = (phi, S, B, args.precision)
            traffic_of = (lambda y: pmc_traffic(y)) if cfg == (0, 256, 16, "bf16") else ((lambda y: pmc_traffic(y, "phi3_")) if cfg == (3, 512, 8, "bf16") else ((lambda y: pmc_traffic(y, "fp32_")) if cfg == (0, 256, 16, "fp32") else None))
            out["roofline"] = launch_profile(s0, B, ms, e1 / k1 * 1e3, traffic_of, layers=not args.no_layers)
            torch.cuda.synchronize(dev)
            if not args.no_fp32 and args.precision != "fp32":
                # the precision that meets the 0.1 mm ADD bound on the seeded weights (add_vs_ref below): same loops, fp32 sessions
                main_loop.close()
                f32 = Loop("fp32")
                f32.preheat(args.preheat_ms / 1e3)
                f32.timed(args.warmup)
                # (at least 200 steps: a 20-step window is 11 ms - fill, drain and clock noise are 2-3 % of it - and this block,
                #  unlike `value`, is not bound to exactly --steps)
                kf = max(args.steps, 200)
                ef = f32.timed(kf) * args.steps / kf          # (scaled to --steps: the fields below keep their meaning)
                f32.timed(10, 1)
                ef1 = f32.timed(k1, 1)
                out["fp32"] = {"value": round(B * args.steps / ef, 2), "ms_per_step": round(ef / args.steps * 1e3, 4), "steps": kf,
                               "one_batch_in_flight": {"value": round(B * k1 / ef1, 2), "ms_per_step": round(ef1 / k1 * 1e3, 4)},
                               "what": "the same step and loops with fp32 sessions (fp32 storage, exact-fp32 MFMA): the only precision within 0.1 mm ADD of the reference on the seeded weights",
                               "roofline": launch_profile(f32.sess[0], B, ef / args.steps * 1e3, ef1 / k1 * 1e3,
                                                          (lambda y: pmc_traffic(y, "fp32_")) if (phi, S, B) == (0, 256, 16) else None, layers=not args.no_layers)}
                f32.close()
            if not args.no_latency and (phi, S, B) == (0, 256, 16):
                lat = latency_b1(dev)
                out["latency_b1"] = lat
                out["latency_b1_ms_p50"] = lat["frame"]["p50_ms"]; out["latency_b1_ms_p99"] = lat["frame"]["p99_ms"]       # prep + inference from frame bytes (the reference's 6-8 ms + 16 / 40 / 175 ms)
                out["latency_b1_host_ms_p50"] = lat["host"]["p50_ms"]; out["latency_b1_host_ms_p99"] = lat["host"]["p99_ms"]   # Session.Run replacement on host arrays + decode + filter
                out["latency_b1_model_load_ms"] = lat["model_load_ms"]
                out["latency_b1_paced_60hz_ms_p50"] = lat["frame_paced_60hz"]["p50_ms"]      # the frame path called once per 16.7 ms (idle device between frames)
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(phi, S)
                out["add_vs_ref"] = add_vs_ref(phi, S)
                # ONE number that satisfies both halves of the metric (frames/s AND ADD(-S) within 0.1 mm of the reference): the
                # fastest measured precision among those inside the bound
                cands = {args.precision: (out["value"], out["one_batch_in_flight"]["value"])}
                if "fp32" in out:
                    cands["fp32"] = (out["fp32"]["value"], out["fp32"]["one_batch_in_flight"]["value"])
                ok = [p for p in out["add_vs_ref"]["meets_bound"] if p in cands]
                if ok:
                    best = max(ok, key=lambda p: cands[p][0])
                    out["meets_add_bound"] = {"dtype": best, "value": cands[best][0], "unit": "frames/s", "one_batch_in_flight": cands[best][1],
                                              "timed_steps": {args.precision: args.steps, **({"fp32": out["fp32"]["steps"]} if "fp32" in out else {})},
                                              "add_mm": out["add_vs_ref"][best]["add_mm"], "add_s_mm": out["add_vs_ref"][best]["add_s_mm"], "bound_mm": 0.1,
                                              "what": f"frames/s of the {best} sessions ({D} batches in flight) - the fastest measured precision whose pose stays within 0.1 mm ADD of the reference's"}
                else:
                    out["meets_add_bound"] = None
            # the same figures as flat scalars next to the nested blocks (a record that keeps only top-level scalars still carries them)
            if "fp32" in out:
                out["fp32_value"] = out["fp32"]["value"]; out["fp32_one_batch"] = out["fp32"]["one_batch_in_flight"]["value"]
            out["one_batch_value"] = out["one_batch_in_flight"]["value"]
            if out.get("meets_add_bound"):
                out["meets_add_bound_value"] = out["meets_add_bound"]["value"]; out["meets_add_bound_dtype"] = out["meets_add_bound"]["dtype"]
            for p_ in ("bf16", "fp32"):
                if p_ in out.get("add_vs_ref", {}):
                    out[f"add_mm_{p_}"] = out["add_vs_ref"][p_]["add_mm"]
            out["roofline_traffic_ratio"] = out["roofline"]["traffic_ratio"]; out["roofline_traffic_bytes_per_launch"] = out["roofline"]["traffic"]
            out["roofline_end_to_end_frac"] = out["roofline"]["end_to_end_frac"]; out["roofline_time_weighted_frac"] = out["roofline"]["time_weighted_frac"]
        print(json.dumps(out), flush=True)
    main_loop.close()
    if world > 1:
        import torch.distributed as td
        td.destroy_process_group()


if __name__ == "__main__":
    main()
