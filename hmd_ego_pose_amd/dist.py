"""Multi-GPU sharding of frame batches: one process per GPU, ``torch.distributed``
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Frames are independent in eval mode (no cross-sample op anywhere on the path), so the
forward itself needs NO collective: every rank holds a full copy of the weights and runs
its own contiguous slice of the global batch (weak scaling).  Collectives appear only at
the edges of a serving job and are kept off the per-frame critical path:

  * ``broadcast_state_dict``  one-time weight broadcast from rank 0 (8.7 MB bf16 for phi 0),
  * ``scatter_frames``        optional: rank 0 holds the global batch and deals slices
                              (point-to-point sends: xGMI gives rank 0 a direct link to each peer),
  * ``gather_detections``     post-filter rows only (<=100 x 75 floats per frame, 30 KB) - never
                              the raw heads (3.6 MB per frame), SURVEY.md section 8e.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1 process = 1 GPU)."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def shard_range(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of the global batch owned by ``rank``; the first
    ``global_batch % world`` ranks take one extra frame."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    base, extra = divmod(global_batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def init(backend: Optional[str] = None, timeout_s: Optional[float] = None) -> Tuple[int, int, int]:
    """Join the job's process group (no-op for a single process).  ``timeout_s`` bounds the rendezvous and every
    collective: a rank that died before joining fails the others after that time instead of after the backend's
    default (10-30 minutes)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        be = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank)
        kw = {"timeout": datetime.timedelta(seconds=timeout_s)} if timeout_s else {}
        dist.init_process_group(be, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def broadcast_state_dict(state: Dict[str, torch.Tensor], device: torch.device, src: int = 0) -> Dict[str, torch.Tensor]:
    """One flat broadcast of every float tensor (one large message instead of ~900 small ones)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return state
    keys = [k for k, v in state.items() if torch.is_floating_point(v)]
    flat = torch.cat([state[k].reshape(-1).float() for k in keys]).to(device)
    dist.broadcast(flat, src)
    out, off = dict(state), 0
    for k in keys:
        n = state[k].numel()
        out[k] = flat[off:off + n].reshape(state[k].shape).to(state[k].device)
        off += n
    return out


def _wire(t: torch.Tensor) -> torch.Tensor:
    """What actually goes into send/recv: device memory over RCCL; gloo (CPU wiring tests, or the one-GPU
    wiring run of bench.py) has no point-to-point ops for device tensors, so those are staged on the host."""
    return t.cpu() if (t.is_cuda and dist.get_backend() == "gloo") else t


def scatter_frames(global_frames: Optional[torch.Tensor], global_batch: int, shape: Tuple[int, ...], device: torch.device,
                   src: int = 0, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """Rank ``src`` holds [global_batch, *shape] (uint8 frames before preprocessing: 4x fewer bytes than fp32);
    every rank returns its contiguous slice.  Point to point: every peer has its own xGMI link to the root, and
    the slices of a contiguous tensor on the root are sent in place (no staging copy); the sends are posted as
    one group."""
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
    lo, hi = shard_range(global_batch, rank, world)
    if world == 1:
        return global_frames[lo:hi].to(device)
    if rank == src:
        g = global_frames if global_frames.is_contiguous() else global_frames.contiguous()
        if g.device != device and dist.get_backend() != "gloo":
            g = g.to(device)
        ops = []
        for r in range(world):
            l, h = shard_range(global_batch, r, world)
            if r != src and h > l:
                ops.append(dist.P2POp(dist.isend, _wire(g[l:h]), r))
        reqs = dist.batch_isend_irecv(ops) if ops else []
        mine = g[lo:hi].to(device)
        for q in reqs:
            q.wait()
        return mine
    mine = torch.empty((hi - lo, *shape), dtype=dtype, device=device)
    if hi > lo:
        buf = _wire(mine)
        dist.recv(buf, src)
        if buf is not mine:
            mine.copy_(buf)
    return mine


def gather_detections(det: Dict[str, torch.Tensor], global_batch: int, dst: int = 0) -> Optional[Dict[str, torch.Tensor]]:
    """Gather the per-frame detection rows of every rank on ``dst`` in global frame order (post-filter rows only:
    <= 30 KB per frame, never the 3.6 MB of raw heads).  All receives of the root are posted as one group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return det
    rank, world = dist.get_rank(), dist.get_world_size()
    keys = sorted(det)
    if rank != dst:
        ops = [dist.P2POp(dist.isend, _wire(det[k].contiguous()), dst) for k in keys if det[k].shape[0] > 0]
        for q in (dist.batch_isend_irecv(ops) if ops else []):
            q.wait()
        return None
    bufs: Dict[str, List[torch.Tensor]] = {k: [] for k in keys}
    ops, staged = [], []
    for k in keys:
        t = det[k].contiguous()
        for r in range(world):
            l, h = shard_range(global_batch, r, world)
            if r == dst:
                bufs[k].append(t)
            elif h > l:
                buf = torch.empty((h - l, *t.shape[1:]), dtype=t.dtype, device=t.device)
                w = _wire(buf)
                ops.append(dist.P2POp(dist.irecv, w, r))
                staged.append((buf, w))
                bufs[k].append(buf)
    for q in (dist.batch_isend_irecv(ops) if ops else []):
        q.wait()
    for buf, w in staged:
        if w is not buf:
            buf.copy_(w)
    return {k: torch.cat(v, 0) for k, v in bufs.items()}


def describe(local_rank: int) -> str:
    """One line per rank for the job log, printed BEFORE the first collective: rank / world, backend, RCCL (nccl) version as
    torch reports it and the device this rank drives - the first thing to read when an N-GPU run misbehaves."""
    rank, world = (dist.get_rank(), dist.get_world_size()) if (dist.is_available() and dist.is_initialized()) else (0, 1)
    be = dist.get_backend() if (dist.is_available() and dist.is_initialized()) else "none"
    ver = "n/a"
    try:
        if torch.cuda.is_available():
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as e:       # (a build without nccl bindings: report, never fail the job over a log line)
        ver = f"unavailable ({type(e).__name__})"
    name = torch.cuda.get_device_properties(local_rank).name if torch.cuda.is_available() else "cpu"
    return f"rank {rank}/{world}: backend {be}, nccl/RCCL {ver}, device cuda:{local_rank} = {name}, HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '(unset)')}"


def warm_up_p2p(device: torch.device, root: int = 0) -> None:
    """One tiny message root -> peer and peer -> root for every peer, then a barrier.  RCCL creates its point-to-point
    communicators (and maps the peer's memory over xGMI) lazily on the first send / receive between two ranks; without this
    that set-up would sit inside the first timed steps of the scatter / gather loop."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    rank, world = dist.get_rank(), dist.get_world_size()
    peers = [r for r in range(world) if r != root] if rank == root else [root]
    out = [torch.full((16,), float(rank), dtype=torch.float32, device=device) for _ in peers]
    inp = [torch.empty((16,), dtype=torch.float32, device=device) for _ in peers]
    ops = []
    for p, o, i in zip(peers, out, inp):
        ops.append(dist.P2POp(dist.isend, _wire(o), p))
        w = _wire(i)
        ops.append(dist.P2POp(dist.irecv, w, p))
    for q in dist.batch_isend_irecv(ops):
        q.wait()
    dist.barrier()


def max_over_ranks(seconds: float, device: torch.device) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
