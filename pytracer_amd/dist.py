"""Multi-GPU: replicate the scene, partition the pixels, gather the image.

One process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over xGMI on ROCm).  The frame's
rows are cut into blocks of ``row_block`` rows; block ``b`` belongs to rank ``b % world_size``
(interleaving balances sky rows against geometry rows).  Every rank renders only its rows into a
compact ``[rows_r, W, 3]`` buffer — there is no exchange inside a frame — and ONE collective step per
frame assembles the ``HdrImage`` on rank 0.  Per-pixel PCG seeds depend only on the global pixel
index, so the assembled image is bit-identical for every world size.

The gather writes straight into the final image: a row block is ``row_block * W * 3`` contiguous values
both in the sender's compact shard (its k-th block) and in the ``[H, W, 3]`` frame (rows
``[b * row_block, (b + 1) * row_block)``), so rank 0 posts one receive per remote block whose destination
IS that slice of the frame, every other rank one send per block, all of them in one batched
point-to-point group (``batch_isend_irecv``: a single ncclGroupStart/End on RCCL).  No padded scratch
copy of the frame, no de-interleave pass; rank 0's own blocks are one strided device copy.

The local renderer is pluggable (``render_local(params) -> tensor``) so the partition/gather logic is
exercised on CPU with the ``gloo`` backend in the tests; in production it is ``DeviceScene.render_into``.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist

from . import abi


def shard_rows(height: int, row_block: int, world_size: int, rank: int) -> List[int]:
    return abi.rows_for_rank(height, row_block, world_size, rank)


def max_shard_rows(height: int, row_block: int, world_size: int) -> int:
    return max(len(shard_rows(height, row_block, world_size, r)) for r in range(world_size))


def shard_blocks(height: int, row_block: int, world_size: int, rank: int) -> List[Tuple[int, int, int]]:
    """``[(first row in the frame, first row in the compact shard, rows)]`` for each block of ``rank``."""
    rb = max(1, int(row_block))
    out, local = [], 0
    for b in range((height + rb - 1) // rb):
        if b % world_size == rank:
            n = min(rb, height - b * rb)
            out.append((b * rb, local, n))
            local += n
    return out


_use_p2p = True  # flips to False (on every rank alike) if the backend refuses batched point-to-point transfers


def _gather_padded(local, height, row_block, world, rank, group, dst, out):
    """The collective of last resort: one ``gather`` of shards padded to a common size, then the block copies."""
    pad = max_shard_rows(height, row_block, world)
    shard = local[:pad] if local.shape[0] >= pad else torch.cat(
        [local, local.new_zeros((pad - local.shape[0],) + tuple(local.shape[1:]))])
    shard = shard.contiguous()
    if rank == dst:
        parts = [torch.empty_like(shard) for _ in range(world)]
        dist.gather(shard, parts, dst=dst, group=group)
        for r in range(world):
            for g0, l0, n in shard_blocks(height, row_block, world, r):
                out[g0:g0 + n].copy_(parts[r][l0:l0 + n], non_blocking=True)
    else:
        dist.gather(shard, None, dst=dst, group=group)


def gather_image(local: torch.Tensor, height: int, row_block: int, group=None, dst: int = 0,
                 out: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """Assemble the frame on ``dst`` from the ranks' compact row shards.

    ``local`` is this rank's ``[>= rows_of_this_rank, W, 3]`` shard (rows beyond its own are ignored).
    Returns the ``[H, W, 3]`` image on ``dst`` (``out`` when given) and ``None`` elsewhere."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1:
        rows = shard_rows(height, row_block, 1, 0)
        return local[: len(rows)]
    # rehearsal on a box with fewer GPUs than ranks (PT_DIST_BACKEND=gloo): gloo moves host memory only
    staged = dist.get_backend(group) == "gloo" and local.is_cuda
    if staged:
        dev_out, dev_local = out, local
        local = dev_local.cpu()
        out = torch.empty((height,) + tuple(local.shape[1:]), dtype=local.dtype) if rank == dst else None
    global _use_p2p
    ops = []
    if rank == dst and out is None:
        out = torch.empty((height,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if not _use_p2p:
        _gather_padded(local, height, row_block, world, rank, group, dst, out)
    elif rank == dst:
        for r in range(world):
            blocks = shard_blocks(height, row_block, world, r)
            if r == dst:
                for g0, l0, n in blocks:  # (a handful of strided copies on the device; no communication)
                    out[g0:g0 + n].copy_(local[l0:l0 + n], non_blocking=True)
            else:
                peer = dist.get_global_rank(group, r) if group is not None else r
                ops += [dist.P2POp(dist.irecv, out[g0:g0 + n], peer, group) for g0, _, n in blocks]
    else:
        peer = dist.get_global_rank(group, dst) if group is not None else dst
        ops = [dist.P2POp(dist.isend, local[l0:l0 + n], peer, group)
               for _, l0, n in shard_blocks(height, row_block, world, rank)]
    if ops:
        try:
            for req in dist.batch_isend_irecv(ops):
                req.wait()  # (RCCL: returns once the group is enqueued on the current stream)
        except (RuntimeError, ValueError, NotImplementedError):
            # the same call fails the same way on every rank (it is refused before anything is sent): all of them
            # switch to the padded gather, for this frame and the following ones
            _use_p2p = False
            _gather_padded(local, height, row_block, world, rank, group, dst, out)
    if staged and rank == dst:
        if dev_out is None:
            dev_out = torch.empty(out.shape, dtype=out.dtype, device=dev_local.device)
        dev_out.copy_(out)
        out = dev_out
    return out if rank == dst else None


def render_sharded(render_local: Callable[[abi.Params], torch.Tensor], params: abi.Params, group=None,
                   row_block: int = 8, dst: int = 0) -> Optional[torch.Tensor]:
    """Render this rank's rows with ``render_local`` and gather the frame to ``dst``.

    ``render_local(p)`` must return this rank's ``[rows_for_rank(p), W, 3]`` tensor for the partition
    written into ``p`` (``row_block``, ``n_ranks``, ``rank``)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    p = abi.copy_params(params, row_block=row_block, n_ranks=world, rank=rank)
    shard = render_local(p)
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
        dt = torch.float32 if params.out_format == abi.OUT_F32 else torch.float64
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.bufs = [torch.zeros((max(self.rows, 1), self.width, 3), dtype=dt, device=self.device) for _ in range(2)]
        self.nbytes = self.rows * self.width * 3 * self.bufs[0].element_size()
        # a dedicated non-blocking stream: launches on the legacy default stream serialise the host with
        # the device (measured: ~51 us/launch on the null stream vs ~4 us on a side stream)
        self.stream = torch.cuda.Stream(device=self.device)
        self.comm = torch.cuda.Stream() if self.world > 1 else None
        self.full = None
        if self.world > 1 and self.rank == 0:
            self.full = [torch.empty((self.height, self.width, 3), dtype=dt, device=self.device) for _ in range(2)]
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
                             out=self.full[b] if self.rank == 0 else None)
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
