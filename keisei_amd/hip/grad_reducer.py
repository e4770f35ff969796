"""Bucketed gradient all-reduce launched DURING the hand-written backward pass (collective C2).

The reference gets compute/communication overlap from DistributedDataParallel's reducer: autograd hooks fire per
parameter while the backward graph is still running (katago_loop.py:494-504, katago_ppo.py:927).  The HIP engine's
backward is ONE autograd node, so DDP's hooks could only fire after the whole backward.  The engine therefore drives
the exchange itself: the 3x3-convolution weight gradients (88 % of the bytes) are written by the weight-gradient
kernels straight into flat bucket buffers (~25 MB, DDP's default bucket size), and as soon as the last kernel of a
bucket has been queued, its all-reduce is issued on the communication stream behind an event -- it then travels over
xGMI while the remaining blocks are still being differentiated.  The small tensors (FC / BatchNorm / head gradients)
follow in two coalesced collectives at the end.  SUM followed by a division by the world size = DDP's averaging.

``KataGoPPOAlgorithm._fused_step`` installs a reducer on the engine and runs forward/backward under ``ddp.no_sync()``;
anything that calls ``loss.backward()`` on the DDP-wrapped model directly keeps DDP's own (post-backward) reduction.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist

BUCKET_BYTES = 25 * 1024 * 1024          # DDP's default bucket_cap_mb


class OverlappedGradReducer:
    def __init__(self, group=None, bucket_bytes: int = BUCKET_BYTES) -> None:
        self.group = group
        self.world = dist.get_world_size(group)
        self.bucket_bytes = bucket_bytes
        self._comm: Optional[torch.cuda.Stream] = None
        self._pending: List[tuple] = []          # (work, flat tensor)
        self._small: List[torch.Tensor] = []
        self.log: List[tuple] = []               # ("launch", tag, numel) / ("finish", n): what tests assert on
        self.collectives = 0

    # -- called by the engine while it differentiates
    def launch(self, flat: torch.Tensor, tag: str, ready_event=None) -> None:
        """all-reduce `flat` (a whole bucket) asynchronously; `ready_event` = recorded on the producing stream after the
        last kernel that writes the bucket."""
        self.collectives += 1
        self.log.append(("launch", tag, flat.numel()))
        if flat.is_cuda:
            if self._comm is None or self._comm.device != flat.device:
                self._comm = torch.cuda.Stream(flat.device)
            with torch.cuda.stream(self._comm):
                if ready_event is not None:
                    self._comm.wait_event(ready_event)
                work = dist.all_reduce(flat, group=self.group, async_op=True)
            flat.record_stream(self._comm)
        else:
            work = dist.all_reduce(flat, group=self.group, async_op=True)
        self._pending.append((work, flat))

    def add_small(self, t: torch.Tensor) -> None:
        """a small gradient tensor: coalesced with the others into one collective at the end of the pass"""
        self._small.append(t)

    def finish(self):
        """end of the backward pass: coalesce the small tensors, wait (stream-ordered on the GPU) for every collective,
        average.  Returns, for the tensors passed to add_small (same order), VIEWS of the reduced packed buffer: the caller
        replaces its gradients by them (copying each of ~500 small tensors back cost ~500 launches per step)."""
        smalls, self._small = self._small, []
        packed = None
        if smalls:
            packed = torch.cat([t.reshape(-1) for t in smalls])
            self.launch(packed, "small", self._event_now(packed))
        flats = []
        for work, flat in self._pending:
            work.wait()                       # NCCL/RCCL: the current stream waits for the collective; gloo: the host does
            flats.append(flat)
        self._pending = []
        if self.world > 1 and flats:
            torch._foreach_div_(flats, float(self.world))
        views = []
        if packed is not None:
            off = 0
            for t in smalls:
                n = t.numel()
                views.append(packed[off:off + n].view_as(t))
                off += n
        self.log.append(("finish", len(flats)))
        return views

    @staticmethod
    def _event_now(t: torch.Tensor):
        if not t.is_cuda:
            return None
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(t.device))
        return ev
