"""Multi-GPU: replicate the scene, partition the pixels, gather the image.

One process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over xGMI on ROCm).  The frame's
rows are cut into blocks of ``row_block`` rows; block ``b`` belongs to rank ``b % world_size``
(interleaving balances sky rows against geometry rows).  Every rank renders only its rows into a
compact ``[rows_r, W, 3]`` buffer — there is no exchange inside a frame — and ONE collective per
frame, a gather to rank 0, assembles the ``HdrImage``.  Per-pixel PCG seeds depend only on the global
pixel index, so the assembled image is bit-identical for every world size.

The local renderer is pluggable (``render_local(params) -> tensor``) so the partition/gather logic is
exercised on CPU with the ``gloo`` backend in the tests; in production it is ``DeviceScene.render_into``.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch
import torch.distributed as dist

from . import abi


def shard_rows(height: int, row_block: int, world_size: int, rank: int) -> List[int]:
    return abi.rows_for_rank(height, row_block, world_size, rank)


def max_shard_rows(height: int, row_block: int, world_size: int) -> int:
    return max(len(shard_rows(height, row_block, world_size, r)) for r in range(world_size))


def gather_image(local: torch.Tensor, height: int, row_block: int, group=None, dst: int = 0,
                 out: Optional[torch.Tensor] = None, scratch: Optional[List[torch.Tensor]] = None,
                 row_index: Optional[List[torch.Tensor]] = None) -> Optional[torch.Tensor]:
    """Gather the ranks' compact row shards to ``dst`` and de-interleave them into ``[H, W, 3]``.

    ``local`` is ``[max_shard_rows, W, 3]`` (shards padded to a uniform size).  Returns the full
    image on ``dst`` and ``None`` elsewhere.  ``row_index[r]`` (optional, on the device) caches rank r's
    global row numbers so a frame loop does not rebuild and upload them every frame."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1:
        rows = shard_rows(height, row_block, 1, 0)
        return local[: len(rows)]
    if rank == dst:
        if scratch is None:
            scratch = [torch.empty_like(local) for _ in range(world)]
        dist.gather(local, scratch, dst=dst, group=group)
        if out is None:
            out = torch.empty((height,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        for r in range(world):
            idx = row_index[r] if row_index is not None else torch.as_tensor(
                shard_rows(height, row_block, world, r), dtype=torch.long, device=local.device)
            out.index_copy_(0, idx, scratch[r][: idx.numel()])
        return out
    dist.gather(local, None, dst=dst, group=group)
    return None


def render_sharded(render_local: Callable[[abi.Params], torch.Tensor], params: abi.Params, group=None,
                   row_block: int = 8, dst: int = 0) -> Optional[torch.Tensor]:
    """Render this rank's rows with ``render_local`` and gather the frame to ``dst``.

    ``render_local(p)`` must return this rank's ``[rows_for_rank(p), W, 3]`` tensor for the partition
    written into ``p`` (``row_block``, ``n_ranks``, ``rank``)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    p = abi.copy_params(params, row_block=row_block, n_ranks=world, rank=rank)
    shard = render_local(p)
    pad_rows = max_shard_rows(params.height, row_block, world)
    if shard.shape[0] != pad_rows:
        padded = torch.zeros((pad_rows,) + tuple(shard.shape[1:]), dtype=shard.dtype, device=shard.device)
        padded[: shard.shape[0]] = shard
        shard = padded
    return gather_image(shard.contiguous(), params.height, row_block, group=group, dst=dst)


class ShardedFrameLoop:
    """Frame loop for one GPU rank: render into HBM, gather over RCCL on a side stream.

    Double-buffered: the gather of frame ``i`` (comm stream) overlaps the render of frame ``i+1``
    (compute stream).  ``finish()`` drains both streams."""

    def __init__(self, scene, cam: abi.Camera, params: abi.Params, group=None, row_block: int = 8,
                 device: Optional[torch.device] = None):
        self.scene, self.cam = scene, cam
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.row_block = row_block
        self.params = abi.copy_params(params, row_block=row_block, n_ranks=self.world, rank=self.rank)
        self.height, self.width = params.height, params.width
        self.rows = len(shard_rows(self.height, row_block, self.world, self.rank))
        self.pad_rows = max_shard_rows(self.height, row_block, self.world)
        dt = torch.float32 if params.out_format == abi.OUT_F32 else torch.float64
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.bufs = [torch.zeros((self.pad_rows, self.width, 3), dtype=dt, device=self.device) for _ in range(2)]
        self.nbytes = self.rows * self.width * 3 * self.bufs[0].element_size()
        # a dedicated non-blocking stream: launches on the legacy default stream serialise the host with
        # the device (measured: ~51 us/launch on the null stream vs ~4 us on a side stream)
        self.stream = torch.cuda.Stream(device=self.device)
        self.comm = torch.cuda.Stream() if self.world > 1 else None
        self.scratch = None
        self.full = None
        self.row_index = None
        if self.world > 1 and self.rank == 0:
            self.scratch = [[torch.empty_like(self.bufs[0]) for _ in range(self.world)] for _ in range(2)]
            self.full = [torch.empty((self.height, self.width, 3), dtype=dt, device=self.device) for _ in range(2)]
            self.row_index = [torch.as_tensor(shard_rows(self.height, row_block, self.world, r), dtype=torch.long,
                                              device=self.device) for r in range(self.world)]
        self._free = [None, None]  # event: the gather that last read buffer b is done
        self.last = 0

    def step(self, i: int, gather: bool = True) -> None:
        """Render frame ``i`` into this rank's buffer; with ``gather`` also assemble it on rank 0."""
        b = i & 1
        if self._free[b] is not None:
            self.stream.wait_event(self._free[b])
        self.scene.render_into(self.cam, self.params, self.bufs[b].data_ptr(), self.nbytes,
                               self.stream.cuda_stream)
        self.last = b
        if self.world > 1 and gather:
            rendered = torch.cuda.Event()
            rendered.record(self.stream)
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(rendered)
                gather_image(self.bufs[b], self.height, self.row_block, group=self.group, dst=0,
                             out=self.full[b] if self.rank == 0 else None,
                             scratch=self.scratch[b] if self.rank == 0 else None, row_index=self.row_index)
                done = torch.cuda.Event()
                done.record(self.comm)
                self._free[b] = done

    def finish(self) -> None:
        torch.cuda.synchronize()

    def image(self) -> Optional[torch.Tensor]:
        """The last assembled frame (rank 0; ``[H, W, 3]`` in HBM)."""
        if self.world == 1:
            return self.bufs[self.last][: self.rows]
        return self.full[self.last] if self.rank == 0 else None
