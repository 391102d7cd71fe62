"""Batches in flight: the serving loop bench.py measures, as a reusable helper.

One forward is a chain of ~80 small dependent launches that cannot fill the 256 CUs of an
MI355X, so a serving loop keeps several batches in flight: ``depth`` sessions (each a
``hep_handle`` with its own activation arena, graph and output buffers, all built from the
same weights) take the submitted batches round-robin on ``depth`` HIP streams.  Measured at
batch 16, phi 0, bf16 (round 3): 26.2k frames/s with one batch in flight, 49.7k with four.  HIP
spreads streams over four hardware queues per process and streams beyond four share one (five busy
streams: 41.6k frames/s), so the pool owns exactly ``depth`` streams - one per batch in flight - and
the ``depth + 1`` slots take turns on them (with one stream per slot, depth 4 measured 37.3k).  Through
this pool: 25.6k / 38.9k / 43.8k / 47.2k frames/s at depth 1 / 2 / 3 / 4.

    pool = InflightPool(state_dict, phi=0, size=256, max_batch=16, precision="bf16", depth=4)
    for frames, camera in loader:                 # frames: fp32 [B,3,S,S] on the GPU
        done = pool.submit(frames, camera)        # returns the oldest finished result or None
        if done is not None: consume(done)        # ... before the next submit()
    for done in pool.drain(): consume(done)

``submit`` enqueues on a FREE slot and then hands back the oldest batch in flight (synchronised), whose
slot is the one the next ``submit`` will reuse.  A returned dict of tensors is owned by its slot and
stays untouched until the NEXT call of ``submit``/``drain``: nothing is enqueued on those buffers while
the caller reads them (consume or copy them before submitting again).  ``depth`` is the number of batches IN FLIGHT while
the caller consumes a result: the pool holds ``depth + 1`` slots (sessions), one of which is always the one being read.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Iterator, List, Optional

import torch

from . import _capi
from .model import Session


class InflightPool:
    def __init__(self, state_dict, phi: int, size: int, max_batch: int, precision: str = "bf16", depth: int = 4,
                 device: Optional[torch.device] = None):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.depth = depth
        depth = depth + 1          # slots: `depth` in flight + the one whose result the caller is reading
        self.sessions: List[Session] = [Session(state_dict, phi, size, max_batch, precision, device) for _ in range(depth)]
        self.device = self.sessions[0].device
        self.streams = [torch.cuda.Stream(self.device) for _ in range(self.depth)]      # one per batch IN FLIGHT (see the module docstring)
        self._done = [torch.cuda.Event() for _ in range(depth)]                         # recorded behind a slot's last launch
        self._seq = 0                                                                    # submissions so far: picks the stream
        n = self.sessions[0].num_anchors
        f = lambda *s: torch.empty(s, dtype=torch.float32, device=self.device)
        # the five head outputs of a slot ARE its session's own output buffers (hep_output_device): handing the forward
        # caller-side tensors instead costs five device-to-device copies per batch (75 MB at batch 16: 45.4k -> 49k frames/s)
        self.slots: List[Dict[str, torch.Tensor]] = []
        for sess in self.sessions:
            reg, cls, rot, trn, hand = sess.output_views()
            self.slots.append(dict(regression=reg, classification=cls, rotation=rot, translation_raw=trn, hand=hand,
                                   boxes=f(max_batch, n, 4), translation=f(max_batch, n, 3)))
        self._busy: List[Optional[int]] = [None] * depth          # batch size of the step in flight on a slot
        self._next = 0
        self._inputs: List[Optional[tuple]] = [None] * depth      # keeps the caller's tensors alive while in flight

    def close(self):
        for s in self.sessions:
            s.close()

    def _collect(self, d: int) -> Optional[Dict[str, torch.Tensor]]:
        if self._busy[d] is None:
            return None
        self._done[d].synchronize()          # (the slot's own work only: its stream may already carry the newest submission)
        b, self._busy[d], self._inputs[d] = self._busy[d], None, None
        return {k: v[:b] for k, v in self.slots[d].items()}

    def submit(self, frames: torch.Tensor, camera: torch.Tensor) -> Optional[Dict[str, torch.Tensor]]:
        """Enqueue forward + box/translation decode of ``frames`` on a free slot.  Returns the oldest batch
        in flight (synchronised) once every slot is occupied, else None.  The returned tensors are not
        written again before the next call of ``submit`` (the slot they live in is the next one to be reused)."""
        d = self._next
        assert self._busy[d] is None, "slot was not collected"      # invariant: the slot submitted to is always free
        # the `depth` batches in flight are the last `depth` submissions: consecutive submissions take consecutive streams, so
        # they never share one; this submission queues behind the oldest one, which is collected below
        sess, st, out = self.sessions[d], self.streams[self._seq % self.depth], self.slots[d]
        self._seq += 1
        b = frames.shape[0]
        if not frames.is_cuda or frames.dtype != torch.float32 or tuple(frames.shape[1:]) != (3, sess.size, sess.size) or b > sess.max_batch:
            raise ValueError(f"expected float32 ROCm frames [<= {sess.max_batch},3,{sess.size},{sess.size}]")
        cam = camera.to(self.device, torch.float32).contiguous()
        st.wait_stream(torch.cuda.current_stream(self.device))      # the caller produced `frames` (and read this slot) on its own stream
        lib = _capi.lib()
        strides = (ctypes.c_int64 * 4)(*frames.stride())
        _capi.check(lib.hep_run_device(sess.handle, frames.data_ptr(), strides, b, None, None, st.cuda_stream))       # results stay in the session's buffers
        _capi.check(lib.hep_decode_device(sess.handle, None, None, cam.data_ptr(), b, out["boxes"].data_ptr(), out["translation"].data_ptr(), st.cuda_stream))
        self._done[d].record(st)
        self._busy[d], self._inputs[d] = b, (frames, cam)
        self._next = (d + 1) % len(self.sessions)
        return self._collect(self._next)          # frees the slot the next submit uses; None while the pool is filling

    def drain(self) -> Iterator[Dict[str, torch.Tensor]]:
        """Results still in flight, oldest first."""
        n = len(self.sessions)
        for i in range(n):
            r = self._collect((self._next + i) % n)
            if r is not None:
                yield r
