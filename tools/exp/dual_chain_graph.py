#!/usr/bin/env python3
"""Does ONE captured graph with k independent forward chains (k sessions, forked streams inside the capture) run them concurrently on the
graph's launch stream - i.e. more than four chains in flight on the four hardware queues?  frames/s for k chains per graph x 4 graphs in flight.
usage: python tools/exp/dual_chain_graph.py [k ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from hmd_ego_pose_amd import _capi
from hmd_ego_pose_amd.model import Session
from hmd_ego_pose_amd.weights import seeded_state_dict
B, NG = 16, 4
sd = seeded_state_dict(0, 0)
x = torch.randn(B, 3, 256, 256, device="cuda")
for k in [int(a) for a in sys.argv[1:]] or [1, 2, 3]:
    sess = [[Session(sd, 0, 256, B, "bf16", flags=_capi.FLAG_NO_GRAPH) for _ in range(k)] for _ in range(NG)]
    streams = [torch.cuda.Stream() for _ in range(NG)]
    graphs = []
    for gi in range(NG):
        for s in sess[gi]: s.forward(x, want_features=False)          # warm up outside capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        side = [torch.cuda.Stream() for _ in range(k - 1)]
        with torch.cuda.graph(g, stream=streams[gi]):
            cur = torch.cuda.current_stream()
            for st in side: st.wait_stream(cur)
            outs = [sess[gi][0].forward(x, want_features=False)]
            for st, s in zip(side, sess[gi][1:]):
                with torch.cuda.stream(st):
                    outs.append(s.forward(x, want_features=False))
            for st in side: cur.wait_stream(st)
        graphs.append((g, outs))
    def run(iters):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(iters):
            gi = i % NG
            with torch.cuda.stream(streams[gi]): graphs[gi][0].replay()
        torch.cuda.synchronize(); return time.perf_counter() - t0
    run(40)
    it = 400
    dt = run(it)
    print(f"{k} chain(s) per graph x {NG} graphs in flight: {B * k * it / dt:.0f} frames/s ({dt / it * 1e6:.1f} us per graph launch)", flush=True)
    for row in sess:
        for s in row: s.close()
