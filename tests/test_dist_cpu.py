"""CPU, world_size 2 over gloo: the N>1 plumbing of the path - contiguous frame sharding,
the one-time weight broadcast, frame scatter and detection gather (hmd_ego_pose_amd/dist.py).
The forward itself needs no collective (frames are independent), so this is the whole
multi-GPU surface; on the GPU box the same code runs over RCCL (backend "nccl")."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hmd_ego_pose_amd.dist import broadcast_state_dict, describe, gather_detections, max_over_ranks, scatter_frames, shard_range, warm_up_p2p


def test_shard_range_covers_batch_contiguously():
    for g in (1, 7, 16, 128, 130):
        for w in (1, 2, 3, 8):
            spans = [shard_range(g, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == g
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [h - l for l, h in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(128, 3, 8) == (48, 64)
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q, G=5):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    try:
        # the per-rank job-log line and the point-to-point warm-up bench.py issues before its timed serving loop
        line = describe(0)
        assert f"rank {rank}/{world}" in line and "backend gloo" in line
        warm_up_p2p(dev)
        # weights: rank 0's values must arrive everywhere
        state = {"a.weight": torch.full((3, 4), float(rank + 1)), "b.num_batches_tracked": torch.tensor(rank), "c.bias": torch.arange(5.) * (rank + 1)}
        out = broadcast_state_dict(state, dev)
        assert torch.equal(out["a.weight"], torch.ones(3, 4)) and torch.equal(out["c.bias"], torch.arange(5.))
        assert out["b.num_batches_tracked"].item() == rank          # non-float entries stay local
        # frames: global batch G over the ranks (5 on 2 ranks -> 3 + 2; 3 on 4 ranks -> 1 + 1 + 1 + 0)
        frames = torch.arange(G * 6, dtype=torch.float32).reshape(G, 1, 2, 3) if rank == 0 else None
        mine = scatter_frames(frames, G, (1, 2, 3), dev)
        lo, hi = shard_range(G, rank, world)
        assert mine.shape[0] == hi - lo and torch.equal(mine, torch.arange(G * 6, dtype=torch.float32).reshape(G, 1, 2, 3)[lo:hi])
        # "forward": per-frame result depends only on the frame -> gather restores global order
        det = {"scores": mine.sum(dim=(1, 2, 3)), "index": torch.arange(lo, hi, dtype=torch.int32)}
        got = gather_detections(det, G)
        if rank == 0:
            assert torch.equal(got["index"], torch.arange(G, dtype=torch.int32))
            assert torch.equal(got["scores"], torch.arange(G * 6, dtype=torch.float32).reshape(G, -1).sum(1))
        else:
            assert got is None
        assert max_over_ranks(float(rank + 1), dev) == float(world)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_roundtrip():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


@pytest.mark.parametrize("G", [6, 3])
def test_four_rank_gloo_uneven_and_empty_shards(G):
    """world size 4: 6 frames -> 2 + 2 + 1 + 1 (uneven), 3 frames -> 1 + 1 + 1 + 0 (rank 3 owns NOTHING: it must neither
    send nor be waited for in the scatter and in the gather)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 4, port, q, G)) for r in range(4)]
    for p in ps:
        p.start()
    res = [q.get(timeout=180) for _ in ps]
    for p in ps:
        p.join(60)
    assert sorted(res) == [(r, "ok") for r in range(4)], res


def _worker8(rank, world, port, q, G):
    """Configs 3 / 5 of BASELINE.json at their real shard arithmetic: G = 128 / 256 uint8 frames of 256x256x3 on eight ranks
    (16 / 32 per rank), the flat weight broadcast of a phi-0-sized state_dict, and the gather of the post-filter rows."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    try:
        torch.set_num_threads(1)
        warm_up_p2p(dev)
        # weights: a state_dict of the real phi 0 shapes (3.9 M floats in 1048 tensors), rank 0's values everywhere
        from hmd_ego_pose_amd.weights import seeded_state_dict
        sd0 = seeded_state_dict(0, 0)
        mine_sd = sd0 if rank == 0 else {k: torch.zeros_like(v) for k, v in sd0.items()}
        got_sd = broadcast_state_dict(mine_sd, dev)
        assert all(torch.equal(got_sd[k], sd0[k]) for k in sd0 if torch.is_floating_point(sd0[k]))
        # frames: row r of the global batch is filled with a value derived from r (cheap to build, exact to check)
        S, M = 256, 100
        if rank == 0:
            frames = (torch.arange(G, dtype=torch.int32) * 7 % 251).to(torch.uint8).reshape(G, 1, 1, 1).expand(G, S, S, 3).contiguous()
            frames[:, 0, 0, 0] = (torch.arange(G, dtype=torch.int32) % 256).to(torch.uint8)
        else:
            frames = None
        lo, hi = shard_range(G, rank, world)
        assert hi - lo == G // world                                  # 16 (config 3) / 32 (config 5) frames per rank
        for _step in range(2):                                        # two steps: the grouped sends / receives are reusable
            mine = scatter_frames(frames, G, (S, S, 3), dev, dtype=torch.uint8)
            assert mine.shape == (hi - lo, S, S, 3) and mine.dtype == torch.uint8
            idx = torch.arange(lo, hi, dtype=torch.int32)
            assert torch.equal(mine[:, 5, 7, 1].to(torch.int32), idx * 7 % 251) and torch.equal(mine[:, 0, 0, 0].to(torch.int32), idx % 256)
            # post-filter rows as hep_filter_device leaves them: [n, M, k] per key, int32 / float32
            n = hi - lo
            det = {"boxes": idx.float().reshape(n, 1, 1).expand(n, M, 4).contiguous(), "scores": idx.float().reshape(n, 1).expand(n, M).contiguous(),
                   "labels": torch.zeros((n, M), dtype=torch.int32), "rotation": torch.zeros((n, M, 3)), "translation": torch.zeros((n, M, 3)),
                   "hand": idx.float().reshape(n, 1, 1).expand(n, M, 63).contiguous(), "index": idx.reshape(n, 1).expand(n, M).contiguous(), "count": idx.clone()}
            got = gather_detections(det, G)
            if rank == 0:
                want = torch.arange(G, dtype=torch.int32)
                assert got["count"].shape == (G,) and torch.equal(got["count"], want)                 # global frame order on rank 0
                assert got["boxes"].shape == (G, M, 4) and torch.equal(got["boxes"][:, 3, 2], want.float())
                assert got["hand"].shape == (G, M, 63) and torch.equal(got["hand"][:, 99, 62], want.float())
                assert got["index"].dtype == torch.int32 and torch.equal(got["index"][:, 0], want)
            else:
                assert got is None
        assert max_over_ranks(float(rank + 1), dev) == float(world)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, repr(e) + traceback.format_exc()[-600:]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("G", [128, 256])
def test_eight_rank_gloo_config_shapes(G):
    """BASELINE configs 3 (batch 128) and 5 (batch 256) on eight ranks: weight broadcast -> scatter of the uint8 frames (16 / 32 per
    rank, 196 608 bytes each) -> gather of the detection rows, rows in global order on rank 0.  The first real 8-GPU run is the
    driver's; this is the same code over gloo."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker8, args=(r, 8, port, q, G)) for r in range(8)]
    for p in ps:
        p.start()
    res = [q.get(timeout=300) for _ in ps]
    for p in ps:
        p.join(60)
    assert sorted(res) == [(r, "ok") for r in range(8)], res


def test_bench_refuses_more_gpus_than_visible_before_touching_one():
    """bench.py --gpus N without a torchrun environment: the parent checks the device count (no HIP initialisation) and
    exits non-zero instead of starting ranks that would hang in the rendezvous."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "GPU(s) visible" in r.stderr, (r.returncode, r.stderr[-400:])
    # under a torchrun-style environment a rank whose local rank has no device exits before the rendezvous (the local rank is
    # the first one BEYOND the devices this host has - on a multi-GPU box LOCAL_RANK=1 owns a device and would wait in the
    # rendezvous for its peers)
    import torch
    nd = torch.cuda.device_count()
    env2 = dict(env, WORLD_SIZE=str(nd + 1), RANK=str(nd), LOCAL_RANK=str(nd), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(nd + 1), "--steps", "1", "--warmup", "0"], env=env2,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "GPU(s) visible" in (r.stderr + r.stdout), (r.returncode, r.stderr[-400:])
