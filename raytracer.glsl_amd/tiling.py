"""Framebuffer tiling across the GPUs of one node + the frame-end gather.

Partition: the image is cut into strips of `strip_rows` rows; strip s belongs to rank s % world.
Every rank renders only its strips into a compact local buffer (strips packed in increasing order)
and accumulates them locally across frames; pixel seeds and camera rays use absolute coordinates,
so the union over ranks is bit-identical to a single-GPU render.  The only exchange step is the
gather of the local buffers when the image is consumed: one `torch.distributed.gather` (RCCL over
xGMI with the nccl backend; gloo in the CPU tests) of equal-size padded buffers to rank 0, followed
by an un-permute into image order.  No reduction is involved (accumulation is pixel-local), so no
all-reduce and no ring: with 7 point-to-point xGMI links per GPU every sender reaches rank 0
directly and the payload per rank is W*H*16/world bytes (4.15 MB at 1080p / 8 GPUs).
"""
from __future__ import annotations

import numpy as np

DEFAULT_STRIP_ROWS = 16


def strip_rows_of(height: int, rank: int, world: int, strip_rows: int = DEFAULT_STRIP_ROWS) -> np.ndarray:
    """Global row indices owned by `rank`, in local (packed) order."""
    rows = []
    n_strips = (height + strip_rows - 1) // strip_rows
    for s in range(rank, n_strips, world):
        rows.extend(range(s * strip_rows, min((s + 1) * strip_rows, height)))
    return np.asarray(rows, np.int64)


def padded_rows(height: int, world: int, strip_rows: int = DEFAULT_STRIP_ROWS) -> int:
    """Row count every rank pads its buffer to, so the gather moves equal-size messages."""
    return max(len(strip_rows_of(height, r, world, strip_rows)) for r in range(world))


def row_permutation(height: int, world: int, strip_rows: int = DEFAULT_STRIP_ROWS) -> np.ndarray:
    """index[y] = position of global row y inside the concatenation of the padded rank buffers."""
    pad = padded_rows(height, world, strip_rows)
    index = np.full(height, -1, np.int64)
    for r in range(world):
        rows = strip_rows_of(height, r, world, strip_rows)
        index[rows] = r * pad + np.arange(len(rows))
    assert (index >= 0).all()
    return index


class FrameGatherer:
    """Gathers the per-rank tile buffers to rank 0 and restores image order.

    local buffers are torch tensors of shape (padded_rows, width, 4) float32 living on the device
    the process renders on (CPU tensors with the gloo backend in tests)."""

    def __init__(self, width: int, height: int, rank: int, world: int, device, strip_rows: int = DEFAULT_STRIP_ROWS,
                 force_collective: bool = False):
        """force_collective: take the gather + un-permute path even for a single rank (a one-rank process group must be
        initialised).  Exists so that the RCCL calls of the multi-GPU path can be exercised on a one-GPU box."""
        import torch
        self.torch = torch
        self.width, self.height, self.rank, self.world, self.strip_rows = width, height, rank, world, strip_rows
        self.collective = world > 1 or force_collective
        self.send = self.pending = None
        self.pad = padded_rows(height, world, strip_rows)
        self.local = torch.zeros((self.pad, width, 4), dtype=torch.float32, device=device)
        self.n_local = len(strip_rows_of(height, rank, world, strip_rows))
        if not self.collective:
            # a single rank owns every row in image order: the local buffer IS the assembled image, nothing to exchange or copy
            self.gathered = self.perm = None
            self.full = self.local[:height]
        elif rank == 0:
            self.gathered = torch.zeros((world, self.pad, width, 4), dtype=torch.float32, device=device)
            self.perm = torch.from_numpy(row_permutation(height, world, strip_rows)).to(device)
            self.full = torch.zeros((height, width, 4), dtype=torch.float32, device=device)
        else:
            self.gathered = self.perm = self.full = None

    def gather(self, overlap: bool = False):
        """One exchange step.  Returns the assembled (height, width, 4) image on rank 0, None elsewhere.

        overlap=False: the exchange of THIS frame completes (stream-ordered) before anything enqueued afterwards runs.
        overlap=True: the tile buffer is snapshotted (one device copy) and gathered asynchronously while the caller renders
        the next frame into `local`; the image returned on rank 0 is the one completed by the PREVIOUS call (None the first
        time) -- a display one frame late, the usual trade of a progressive renderer.  Call finish() after the last frame."""
        torch = self.torch
        if not self.collective:
            return self.full
        import torch.distributed as dist
        if overlap:
            had = self._complete()
            if self.send is None:
                self.send = torch.empty_like(self.local)
            self.send.copy_(self.local)                              # snapshot on the current stream
            self.pending = dist.gather(self.send, gather_list=list(self.gathered.unbind(0)) if self.rank == 0 else None, dst=0, async_op=True)
            return self.full if (self.rank == 0 and had) else None
        self._complete()
        if self.rank == 0:
            dist.gather(self.local, gather_list=list(self.gathered.unbind(0)), dst=0)
            torch.index_select(self.gathered.view(self.world * self.pad, self.width, 4), 0, self.perm, out=self.full)
            return self.full
        dist.gather(self.local, gather_list=None, dst=0)
        return None

    def _complete(self) -> bool:
        """Finish the exchange started by the last gather(overlap=True): wait (stream-ordered for RCCL, blocking for gloo),
        then un-permute on rank 0."""
        if self.pending is None:
            return False
        self.pending.wait()
        self.pending = None
        if self.rank == 0:
            self.torch.index_select(self.gathered.view(self.world * self.pad, self.width, 4), 0, self.perm, out=self.full)
        return True

    def finish(self):
        """Completes an overlapped exchange; returns the assembled image of the last frame on rank 0."""
        self._complete()
        return self.full if (not self.collective or self.rank == 0) else None
