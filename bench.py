#!/usr/bin/env python3
"""bench.py - frames/sec of the MI355X EfficientPose path at 256x256, batch 16 per GPU, phi 0.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

With --gpus N > 1 and no torchrun environment the script starts its N rank processes itself (fresh
children, started before anything touches a GPU) and relays rank 0's line.

One step = one pass of the hot path over one batch of 16 synthetic, HBM-resident frames per
GPU: the full forward (stem .. heads, one hipGraph replay) + box/translation decode, through
the C ABI of libhep.so.  --inflight D (default 4) keeps D steps in flight on D HIP streams (D
sessions): a single forward is a dependent chain of small kernels and leaves most of the
256 CUs idle; `one_batch_in_flight` reports the strictly sequential number as well.  Weak scaling:
every rank owns 16 frames, the forward needs no collective (frames are independent); ranks meet
only at the timing barriers.  Rank 0 prints ONE JSON line.  It also carries
  comm          the serving loop around the same step with the data movement SURVEY 8(e) describes:
                rank 0 owns the global batch of uint8 frames and scatters 16 to every rank (point to point
                over RCCL/xGMI), each rank runs preprocess -> forward -> decode -> detection filter, and the
                post-filter rows are gathered on rank 0; frames/s with all of that inside the timed loop
  roofline      (N=1) the dominant device function: algorithmic bytes per launch / its in-sequence
                launch duration measured here with HIP events, against 8 TB/s HBM3E
  cpu_baseline  (N=1) the CPU oracle (torch fp32 restatement of the reference) timed on the host
                cores in the evaluate.py regime (batch 1, anchors rebuilt per call, decode).
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "frames/sec at 256x256 bs16 EfficientPose-phi0; ADD(-S) vs ref"
HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec); ~6.3 TB/s achievable
PRECISIONS = ["bf16", "fp32", "fp8"]


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
    amplify rounding, so the bf16 / fp8 figures are upper bounds for trained weights."""
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
    for prec in precisions:
        s_ = Session(sd, phi, size, nf, prec)
        _, g_reg, _g_cls, g_rot, g_trn, _g_hand = s_.forward(x.cuda(), want_features=False)
        _, g_t = s_.decode(g_reg, g_trn, torch.from_numpy(cam).cuda())
        add, add_s = pose_errors(pts, pick(rot.numpy()) * math.pi, pick(t_ref), pick(g_rot.cpu().numpy()) * math.pi, pick(g_t.cpu().numpy()))
        out[prec] = {"add_mm": round(float(add.mean()), 5), "add_s_mm": round(float(add_s.mean()), 5)}
        s_.close()
    out["sample"] = f"{nf} seeded frames, pose at the oracle's best-scoring anchor, 1000-point cloud (sigma 40/25/60 mm), translations ~ N(0, 1) * 1000 mm"
    return out


def pmc_traffic(symbol):
    """HBM bytes per launch of `symbol` from the newest committed PMC pass (profiles/r*/*_pmc_per_kernel.json:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, KB per launch).  gfx950 correction from
    /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE counts 128-byte requests as 64 bytes, so reads are
    doubled; WRITE_SIZE is exact for 16-byte stores.  None when no pass holds the kernel."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "*_pmc_per_kernel.json")), reverse=True):
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
    ap.add_argument("--comm-score-threshold", type=float, default=0.5)
    ap.add_argument("--inflight", type=int, default=4, help="batches in flight per GPU (sessions on separate HIP streams)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for wiring tests)")
    ap.add_argument("--single-device", action="store_true", help="wiring test: every rank uses cuda:0 (needs --backend gloo)")
    return ap.parse_args(argv)


def self_launch(args):
    """--gpus N without a torchrun environment: start N rank processes of this script (one per GPU, env as
    torchrun sets it) and relay their output.  This parent never initialises a GPU and never exec()s."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


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
    rank, local_rank, world = hd.init(args.backend)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: local rank {local_rank} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    B, S, phi = args.batch, args.size, args.phi

    # weights: synthetic (no checkpoint ships with the reference), rank 0's copy broadcast once over RCCL
    sd = hd.broadcast_state_dict(seeded_state_dict(phi, 0), dev)
    D = max(1, args.inflight)
    # D sessions = D batches in flight on D HIP streams: the forward of one batch is a dependent chain of
    # small kernels that cannot fill 256 CUs, so a serving loop keeps several batches in flight
    # (step i runs on slot i % D; every step is a full forward + decode of its own 16 frames)
    sess = [Session(sd, phi, S, B, args.precision, dev) for _ in range(D)]
    streams = [torch.cuda.Stream(dev) for _ in range(D)]
    lib = _capi.lib()
    N = sess[0].num_anchors
    rng = np.random.Generator(np.random.PCG64(1000 + rank))
    xs = [torch.from_numpy(rng.standard_normal((B, 3, S, S)).astype(np.float32)).to(dev) for _ in range(D)]    # resident in HBM
    cam = torch.tensor([[480, 480, 128, 128, 1000, 1.0]] * B, dtype=torch.float32, device=dev)
    boxes = [torch.empty((B, N, 4), dtype=torch.float32, device=dev) for _ in range(D)]
    trans = [torch.empty((B, N, 3), dtype=torch.float32, device=dev) for _ in range(D)]
    strides = (ctypes.c_int64 * 4)(*xs[0].stride())
    torch.cuda.synchronize(dev)

    def step(i, depth=D):
        # forward into the handle's own output buffers (no copies), then decode from them
        d = i % depth
        st = streams[d].cuda_stream
        _capi.check(lib.hep_run_device(sess[d].handle, xs[d].data_ptr(), strides, B, None, None, st))
        _capi.check(lib.hep_decode_device(sess[d].handle, None, None, cam.data_ptr(), B, boxes[d].data_ptr(), trans[d].data_ptr(), st))

    for d in range(D):            # set-up, not measurement: every session captures its hipGraph on first use
        step(d)
    torch.cuda.synchronize(dev)
    for i in range(args.warmup):
        step(i)
    hd.barrier(); torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize(dev); hd.barrier()
    elapsed = hd.max_over_ranks(time.perf_counter() - t0, dev)
    assert all(torch.isfinite(t).all() for t in boxes) and all(torch.isfinite(t).all() for t in trans)

    # ---- the same step inside a serving loop with its data movement: scatter of uint8 frames from
    #      rank 0, preprocess, forward, decode, detection filter, gather of the post-filter rows on rank 0 ----
    comm = None
    if not args.no_comm:
        G = B * world
        M = 100
        frames_u8 = torch.from_numpy(np.random.Generator(np.random.PCG64(7)).integers(0, 256, (G, S, S, 3), dtype=np.uint8)).to(dev) if rank == 0 else None

        def serve():
            mine = hd.scatter_frames(frames_u8, G, (S, S, 3), dev, dtype=torch.uint8)       # 196 KB per frame instead of 786 KB fp32
            x = sess[0].preprocess(mine)                                                  # NCHW view of normalised NHWC memory
            _, reg, cls, rot, trn, hand = sess[0].forward(x, want_features=False)
            bx, tr = sess[0].decode(reg, trn, cam)
            det = sess[0].filter(bx, cls, rot, tr, hand, args.comm_score_threshold, 0.5, M)
            return hd.gather_detections(det, G)

        for _ in range(3):
            got = serve()
        torch.cuda.synchronize(dev); hd.barrier()
        k2 = max(10, args.steps // 4)
        t1 = time.perf_counter()
        for _ in range(k2):
            got = serve()
        torch.cuda.synchronize(dev); hd.barrier()
        e2 = hd.max_over_ranks(time.perf_counter() - t1, dev)
        if rank == 0:
            assert got["count"].shape[0] == G and got["boxes"].shape == (G, M, 4)
            row_bytes = M * (4 + 1 + 1 + 3 + 3 + 63 + 1) * 4 + 4
            comm = {"value": round(G * k2 / e2, 2), "unit": "frames/s", "ms_per_step": round(e2 / k2 * 1e3, 4), "steps": k2,
                    "what": "one batch in flight per GPU: scatter uint8 frames from rank 0 (point to point) -> preprocess -> forward -> decode -> "
                            f"filter (score > {args.comm_score_threshold}, NMS 0.5, top {M}) -> gather detection rows on rank 0",
                    "scatter_bytes_per_step": int((G - B) * S * S * 3), "gather_bytes_per_step": int((G - B) * row_bytes),
                    "backend": args.backend if world > 1 else None, "mean_detections_per_frame": round(float(got["count"].float().mean()), 1)}

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        out = {
            "metric": METRIC if (B, S, phi) == (16, 256, 0) else f"frames/sec at {S}x{S} bs{B} EfficientPose-phi{phi}; ADD(-S) vs ref", "value": round(B * world * args.steps / elapsed, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"EfficientPose phi={phi} {S}x{S} {args.precision} batch={B} per GPU: forward (stem, MBConv, BiFPN, 5 heads) "
                                   f"+ box/translation decode; seeded random-init weights, N(0,1) frames resident in HBM; "
                                   f"{D} batches of {B} in flight per GPU on {D} HIP streams",
                       "phi": phi, "size": S, "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}",
                       "batches_in_flight": D, "anchors": N, "launches_per_step": len(sess[0].kernels(B)) + 1},
        }
        if comm is not None:
            out["comm"] = comm
        if world == 1:
            # the strictly sequential number (one batch in flight): latency of a step
            for i in range(10):
                step(i, 1)
            torch.cuda.synchronize(dev); t1 = time.perf_counter()
            k1 = max(20, args.steps // 4)
            for i in range(k1):
                step(i, 1)
            torch.cuda.synchronize(dev); e1 = time.perf_counter() - t1
            out["one_batch_in_flight"] = {"value": round(B * k1 / e1, 2), "ms_per_step": round(e1 / k1 * 1e3, 4)}
            # per-launch durations: one batch in flight (the only regime in which a launch can be timed alone;
            # with several batches in flight launches of different batches overlap on the chip)
            total_ms, per = sess[0].profile(B, 20, per_kernel=True)
            torch.cuda.synchronize(dev)
            ks = sess[0].kernels(B)
            # per-launch durations come from an eager pass with a HIP event in front of every launch; the
            # event pairs add a constant to each launch.  Calibrate it live: the same launches replayed as one
            # hipGraph (no events) take total_ms, so the per-launch overhead is (sum(eager) - total_ms) / n.
            ev_overhead = max(0.0, (sum(per) - total_ms) / len(per))
            per = [max(t - ev_overhead, 0.0) for t in per]
            agg = {}
            for (name, nbytes, flops, sym), t in zip(ks, per):
                a = agg.setdefault(sym, [0.0, 0.0, 0.0, 0])
                a[0] += t; a[1] += nbytes; a[2] += flops; a[3] += 1
            sym, (t, nbytes, flops, calls) = max(agg.items(), key=lambda kv: kv[1][0])
            achieved = nbytes / (t * 1e-3) / 1e9
            step_bytes = sum(k[1] for k in ks)
            out["roofline"] = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(sym),
                               "kernel": sym, "launches_per_step": calls, "avg_launch_us": round(t / calls * 1e3, 2),
                               "algorithmic_bytes_per_launch": round(nbytes / calls), "share_of_step": round(t / sum(per), 3),
                               "measured_with_batches_in_flight": 1,
                               "graph_replay_ms": round(total_ms, 4), "event_overhead_us_subtracted": round(ev_overhead * 1e3, 2),
                               "whole_step_algorithmic_GBps": round(step_bytes / (ms * 1e-3) / 1e9, 1),
                               "end_to_end_frac": round(step_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            # context for the block above: the next device functions by total time, same definitions
            out["roofline"]["top"] = [
                {"kernel": k, "launches_per_step": v[3], "avg_launch_us": round(v[0] / v[3] * 1e3, 2), "share_of_step": round(v[0] / sum(per), 3),
                 "achieved": round(v[1] / (v[0] * 1e-3) / 1e9, 1), "frac": round(v[1] / (v[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:5]]
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(phi, S)
                out["add_vs_ref"] = add_vs_ref(phi, S)
        print(json.dumps(out), flush=True)
    for s_ in sess:
        s_.close()
    if world > 1:
        import torch.distributed as td
        td.destroy_process_group()


if __name__ == "__main__":
    main()
