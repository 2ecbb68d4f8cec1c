"""Multi-GPU: replicate the scene, partition the pixels, gather the image.

One process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over xGMI on ROCm).  The frame's
rows are cut into blocks of ``row_block`` rows; block ``b`` belongs to rank ``b % world_size``
(interleaving balances sky rows against geometry rows).  Every rank renders only its rows into a
compact ``[rows_r, W, 3]`` buffer — there is no exchange inside a frame — and ONE collective step per
frame assembles the ``HdrImage`` on rank 0.  Per-pixel PCG seeds depend only on the global pixel
index, so the assembled image is bit-identical for every world size.

The gather moves ONE message per remote rank (round 3; round 2 posted one receive per remote 8-row block: 236
transfers of 368 KB per 4K frame at 8 ranks): a rank's compact shard is contiguous, so it is sent whole into a
staging buffer on rank 0, and a strided device copy per rank then drops its blocks into row-block order
(``frame.view(groups, world, row_block, W, 3)[:, r] = shard.view(groups, row_block, W, 3)`` plus at most one ragged
tail block).  The render granularity (8-row blocks, for load balance) and the gather granularity (a rank's whole
share) are thereby independent: ``world - 1`` transfers per frame, all in one batched point-to-point group
(``batch_isend_irecv``: a single ncclGroupStart/End on RCCL), and ``gather_plan()`` reports their number and bytes.

Most of a frame is usually the same colour to the bit -- the sky: 99 % of the pixels of BASELINE's 4K configuration --
and at 12 bytes per pixel the full shards of a 4K frame (87 MB into rank 0 at 8 ranks, 50 MB over ONE link at 2) take
longer to move than to render.  So a shard travels SPARSE by default (``encode_sparse`` / ``decode_sparse``; lossless):
cut into runs of ``SPARSE_TILE`` consecutive pixels, a run whose pixels are all bitwise equal to its first is sent as
that one pixel, the others whole.  Two messages per remote rank -- a fixed-size part (count, one flag and one pixel per
run) and the runs that are not constant -- posted in ONE batched point-to-point group, with NO host round trip in between
(round 5): how many runs the second message carries is not read from the first but agreed on beforehand
WITHOUT communication -- a capacity both sides derive by the same rule from the count of the frame before
(``SparseGatherState``: count + 25 % + 64 runs), which the sender knows from its own encode and rank 0 from the fixed part it
received; the counts travel to the host asynchronously and are looked at one frame LATER (``confirm``), when they have long
arrived.  A frame whose count exceeds its capacity (a scene that suddenly changes) is repaired then: the missing runs follow
in one more message and that rank's shard is decoded again -- before the frame counts as assembled.  Only the very first
frame of a loop, which has no history, reads its counts back before the payload is sized (round 4's protocol).
``PT_GATHER_SPARSE=0`` (or ``sparse=False``) sends the shards whole in one message, as in round 2.

Which transport a process group uses is decided ONCE and COLLECTIVELY (``choose_transport``): every rank, also one
that owns no rows, runs the same probe and the verdicts are all-reduced, so no rank can end up in a different
collective from the others; an error inside a frame's gather is raised, never retried on another transport.

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


P2P, PADDED = "p2p", "padded"
_transport = {}  # process group (None = the default one) -> P2P | PADDED, agreed on by all its ranks


def choose_transport(group=None, force: Optional[str] = None) -> str:
    """Agree, once per process group, on how shards travel: batched point-to-point transfers (one per remote rank),
    or -- for a backend without them -- one ``gather`` of shards padded to a common size.

    Decided from CAPABILITY, not from a trial transfer (a probe that one rank is refused while its neighbours
    have already posted theirs leaves those waiting for ever): ``nccl`` (= RCCL) and ``gloo`` both implement
    ``batch_isend_irecv``, anything else takes the padded ``gather``; ``force`` (or the environment variable
    ``PT_GATHER``) pins the choice.  A collective all the same: EVERY rank of the group calls it (``gather_image`` does,
    on first use), whether or not it owns rows, and the ranks' choices meet in ONE all-reduce -- if they differ (one rank
    started with another ``PT_GATHER``) every rank raises instead of entering different collectives later."""
    import os

    if group in _transport and force is None:
        return _transport[group]
    force = force or os.environ.get("PT_GATHER")
    if force in (P2P, PADDED):
        choice = force
    else:
        choice = P2P if dist.get_backend(group) in ("nccl", "gloo") else PADDED
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    code = 1 if choice == P2P else 0
    lo = torch.tensor([code], dtype=torch.int32, device=dev)
    hi = lo.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    if int(lo.item()) != int(hi.item()):
        raise RuntimeError("the ranks chose different gather transports (PT_GATHER differs between ranks?)")
    _transport[group] = choice
    return choice


def gather_plan(height: int, width: int, row_block: int, world: int, itemsize: int = 4, transport: str = P2P) -> dict:
    """What one frame's gather moves: transfers into rank 0 and their bytes (for the bench line)."""
    rows = [len(shard_rows(height, row_block, world, r)) for r in range(world)]
    if transport == P2P:
        ops = sum(1 for r in range(1, world) if rows[r] > 0)
        nbytes = sum(rows[1:]) * width * 3 * itemsize
    else:
        ops = 1
        nbytes = max(rows) * (world - 1) * width * 3 * itemsize
    return {"transport": transport, "gather_ops_per_frame": ops, "gather_bytes_per_frame": nbytes,
            "rows_per_rank": rows}


def place_shard(out: torch.Tensor, shard: torch.Tensor, height: int, row_block: int, world: int, r: int) -> None:
    """Drop rank ``r``'s compact shard into row-block order: one strided copy for the complete groups of
    ``world`` blocks, one more for a ragged tail block."""
    rb = max(1, int(row_block))
    G = rb * world
    ng = height // G
    tail_shape = tuple(out.shape[1:])
    if ng:
        out[: ng * G].view((ng, world, rb) + tail_shape)[:, r].copy_(shard[: ng * rb].view((ng, rb) + tail_shape),
                                                                     non_blocking=True)
    g0 = ng * G + r * rb  # first row of r's block in the incomplete last group
    n = min(rb, height - g0)
    if n > 0:
        out[g0:g0 + n].copy_(shard[ng * rb: ng * rb + n], non_blocking=True)


class PhaseClock:
    """Wall time per phase of ONE gather, for the bench's per-phase breakdown (``ShardedFrameLoop.phase_probe``): ``mark``
    drains the device and books the time since the previous mark under a name.  Every mark is a host-device
    synchronisation, so a clocked gather runs its phases one after the other -- a measurement mode; the frame loops
    pass no clock and synchronise nowhere."""

    def __init__(self):
        from time import perf_counter

        self._now = perf_counter
        self.ms = {}
        self.restart()

    def restart(self) -> None:
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        self._last = self._now()

    def mark(self, name: str) -> None:
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        t = self._now()
        self.ms[name] = self.ms.get(name, 0.0) + (t - self._last) * 1e3
        self._last = t


def _mark(clock: Optional["PhaseClock"], name: str) -> None:
    if clock is not None:
        clock.mark(name)


SPARSE_TILE = 128  # pixels per run (csrc/pt_post.h: PT_SPARSE_RUN)
last_gather = {}   # on dst, after a gather: {"bytes": what the remote ranks sent for that frame, "sparse": bool}
_sparse_scratch = {}  # (device, dtype, runs, stream) -> payload buffer at full capacity: reused frame after frame on THAT stream


def release_sparse_scratch(stream=None) -> None:
    """Drop the encoder's payload buffers -- of ``stream`` (a ``torch.cuda.Stream``), or all of them."""
    for key in [k for k in _sparse_scratch if stream is None or k[3] == stream.cuda_stream]:
        del _sparse_scratch[key]


def sparse_default() -> bool:
    import os

    return os.environ.get("PT_GATHER_SPARSE", "1") != "0"


def _fmt_of(dtype: torch.dtype) -> int:
    return abi.OUT_F32 if dtype == torch.float32 else abi.OUT_F64


def _int_view(t: torch.Tensor) -> torch.Tensor:
    return t.view(torch.int32 if t.element_size() == 4 else torch.int64)


def sparse_fixed_bytes(n_pixels: int, itemsize: int, tile: int = SPARSE_TILE) -> int:
    nt = (n_pixels + tile - 1) // tile
    return 8 + (nt + 1) // 2 * 8 + nt * 3 * itemsize


def encode_sparse(shard: torch.Tensor, tile: int = SPARSE_TILE) -> Tuple[torch.Tensor, torch.Tensor]:
    """``[rows, W, 3]`` (contiguous) -> ``(fixed, payload)``: ``fixed`` is a uint8 tensor of ``sparse_fixed_bytes`` bytes
    -- the number n of runs that are not constant (int64), for every run its place among those or -1 if it is constant
    (int32, padded to a multiple of 8 bytes), the first pixel of every run -- and ``payload`` the ``[n, tile, 3]`` runs
    that are not constant, in order.  Equality is BITWISE (-0.0 and 0.0 differ, a NaN equals itself).  Reads n back
    to the host (one synchronisation of the current stream).  On a HIP device: three small kernels
    (``pt_image_sparse_encode``, csrc/pt_post.h); elsewhere the torch restatement below -- same bytes."""
    flat = shard.reshape(-1, 3)
    npx = flat.shape[0]
    nt = (npx + tile - 1) // tile
    if shard.is_cuda and tile == SPARSE_TILE:
        from . import _lib

        dev = shard.device
        fixed = torch.empty((sparse_fixed_bytes(npx, shard.element_size()),), dtype=torch.uint8, device=dev)
        # one buffer per STREAM: two loops encoding same-sized shards on different streams must not share a payload that
        # RCCL may still be sending
        key = (dev, shard.dtype, nt, torch.cuda.current_stream(dev).cuda_stream)
        if key not in _sparse_scratch:
            _sparse_scratch[key] = torch.empty((nt, tile, 3), dtype=shard.dtype, device=dev)
        payload = _sparse_scratch[key]
        _lib.check(_lib.lib().pt_image_sparse_encode(dev.index or 0, shard.contiguous().data_ptr(), npx, _fmt_of(shard.dtype),
                                                     fixed.data_ptr(), payload.data_ptr(),
                                                     torch.cuda.current_stream(dev).cuda_stream))
        return fixed, payload[: sparse_count(fixed)]
    if nt * tile != npx:  # the last run is filled up with its own last pixel: that does not make it less constant
        flat = torch.cat([flat, flat[-1:].expand(nt * tile - npx, 3)])
    tiles = flat.contiguous().view(nt, tile, 3)
    bits = _int_view(tiles)
    const = (bits == bits[:, :1, :]).all(dim=2).all(dim=1)
    idx = (~const).nonzero().squeeze(1)
    payload = tiles.index_select(0, idx)
    count = torch.tensor([idx.numel()], dtype=torch.int64, device=shard.device)
    place = torch.full(((nt + 1) // 2 * 2,), -1, dtype=torch.int32, device=shard.device)
    place[idx] = torch.arange(idx.numel(), dtype=torch.int32, device=shard.device)
    fixed = torch.cat([count.view(torch.uint8), place.view(torch.uint8), tiles[:, 0, :].contiguous().view(torch.uint8).reshape(-1)])
    return fixed, payload


def sparse_count(fixed: torch.Tensor) -> int:
    return int(fixed[:8].clone().view(torch.int64).item())


def encode_sparse_full(shard: torch.Tensor, parity: int = 0, tile: int = SPARSE_TILE) -> Tuple[torch.Tensor, torch.Tensor]:
    """``encode_sparse`` without the count's read-back: -> ``(fixed, payload)`` with ``payload`` at FULL capacity
    (``[runs, tile, 3]``; its first ``count`` runs are the ones that are not constant, the rest is undefined) -- nothing here
    waits for the device.  ``parity`` picks one of two payload buffers per stream, so that a frame's runs stay intact until
    the frame has been confirmed (``SparseGatherState``) while the next frame is encoded into the other one."""
    flat = shard.reshape(-1, 3)
    npx = flat.shape[0]
    nt = (npx + tile - 1) // tile
    if shard.is_cuda and tile == SPARSE_TILE:
        from . import _lib

        dev = shard.device
        stream = torch.cuda.current_stream(dev).cuda_stream
        fixed = torch.empty((sparse_fixed_bytes(npx, shard.element_size()),), dtype=torch.uint8, device=dev)
        key = (dev, shard.dtype, nt, stream, int(parity) & 1)
        if key not in _sparse_scratch:
            _sparse_scratch[key] = torch.empty((nt, tile, 3), dtype=shard.dtype, device=dev)
        payload = _sparse_scratch[key]
        _lib.check(_lib.lib().pt_image_sparse_encode(dev.index or 0, shard.contiguous().data_ptr(), npx, _fmt_of(shard.dtype),
                                                     fixed.data_ptr(), payload.data_ptr(), stream))
        return fixed, payload
    fixed, runs = encode_sparse(shard, tile)
    payload = torch.zeros((nt, tile, 3), dtype=shard.dtype, device=shard.device)
    payload[: runs.shape[0]] = runs
    return fixed, payload


def decode_sparse(fixed: torch.Tensor, payload: Optional[torch.Tensor], n_pixels: int, dtype: torch.dtype,
                  tile: int = SPARSE_TILE, frame: Optional[torch.Tensor] = None, row_block: int = 0, world: int = 1,
                  rank: int = 0) -> torch.Tensor:
    """The inverse of ``encode_sparse``.  Without ``frame``: -> the shard, ``[n_pixels, 3]``.  With ``frame``
    (``[H, W, 3]``): the shard's rows are written where the partition (``row_block``, ``world``, ``rank``) puts them, in
    the same pass on a HIP device; -> ``frame``."""
    nt = (n_pixels + tile - 1) // tile
    pl_bytes = (nt + 1) // 2 * 8
    if fixed.is_cuda and tile == SPARSE_TILE:
        from . import _lib

        dev = fixed.device
        n = 0 if payload is None else payload.shape[0]
        out = frame if frame is not None else torch.empty((n_pixels, 3), dtype=dtype, device=dev)
        W = frame.shape[1] if frame is not None else 0
        _lib.check(_lib.lib().pt_image_sparse_decode(dev.index or 0, fixed.data_ptr(), payload.data_ptr() if n else None, n_pixels,
                                                     _fmt_of(dtype), out.data_ptr(), W, int(row_block), world if frame is not None else 1,
                                                     rank, torch.cuda.current_stream(dev).cuda_stream))
        return out
    place = fixed[8:8 + 4 * nt].view(torch.int32)
    firsts = fixed[8 + pl_bytes:].view(dtype).view(nt, 1, 3)
    tiles = firsts.expand(nt, tile, 3).contiguous()
    idx = (place >= 0).nonzero().squeeze(1)
    n = 0 if payload is None else payload.shape[0]
    if idx.numel() > n:  # (a payload buffer may be larger than the count: a loop's receive buffers hold every run)
        raise RuntimeError(f"sparse shard: {n} runs received, {idx.numel()} flagged")
    if idx.numel():
        tiles.index_copy_(0, idx, payload.index_select(0, place[idx].to(torch.int64)))
    shard = tiles.view(nt * tile, 3)[:n_pixels]
    if frame is None:
        return shard
    W = frame.shape[1]
    place_shard(frame, shard.view(n_pixels // W, W, 3), frame.shape[0], row_block, world, rank)
    return frame


def decode_sparse_many(fixed: List[torch.Tensor], payload: List[torch.Tensor], ranks: List[int], frame: torch.Tensor,
                       row_block: int, world: int) -> None:
    """The shards of ``ranks`` (as ``encode_sparse`` made them) straight into their rows of ``frame``: on a HIP device ONE
    launch for all of them (``pt_image_sparse_decode_many``), elsewhere one ``decode_sparse`` each."""
    W = frame.shape[1]
    npx = [len(shard_rows(frame.shape[0], row_block, world, r)) * W for r in ranks]
    if frame.is_cuda and len(ranks) <= 64:
        import ctypes as C

        from . import _lib

        k = len(ranks)
        fx = (C.c_void_p * k)(*[f.data_ptr() for f in fixed])
        py = (C.c_void_p * k)(*[p.data_ptr() if p is not None and p.shape[0] else None for p in payload])
        _lib.check(_lib.lib().pt_image_sparse_decode_many(frame.device.index or 0, k, fx, py, (C.c_longlong * k)(*npx), (C.c_int * k)(*ranks),
                                                          _fmt_of(frame.dtype), frame.data_ptr(), W, int(row_block), world,
                                                          torch.cuda.current_stream(frame.device).cuda_stream))
        return
    for f, p, r, n in zip(fixed, payload, ranks, npx):
        decode_sparse(f, p, n, frame.dtype, frame=frame, row_block=row_block, world=world, rank=r)


def _gather_padded(local, height, row_block, world, rank, group, dst, out):
    """The collective for a backend without batched point-to-point: one ``gather`` of shards padded to a common size."""
    pad = max_shard_rows(height, row_block, world)
    shard = local[:pad] if local.shape[0] >= pad else torch.cat(
        [local, local.new_zeros((pad - local.shape[0],) + tuple(local.shape[1:]))])
    shard = shard.contiguous()
    if rank == dst:
        parts = [torch.empty_like(shard) for _ in range(world)]
        dist.gather(shard, parts, dst=dst, group=group)
        for r in range(world):
            place_shard(out, parts[r], height, row_block, world, r)
    else:
        dist.gather(shard, None, dst=dst, group=group)


class SparseGatherState:
    """What the sparse gather of a frame LOOP remembers between frames so that no frame waits for the host.

    * ``cap[r]``: how many runs the payload message of rank ``r`` carries in the NEXT frame -- derived by ``next_cap`` from the
      count of the last confirmed frame, on the sender from its own encode and on ``dst`` from the fixed part it received:
      the same number on both sides without a message.  Unknown (no frame yet): that frame uses the blocking protocol.
    * ``pending``: frames whose transfers have been posted but whose counts have not been looked at yet.  ``confirm`` (called
      before the next gather and by ``finish``) waits for the counts' asynchronous copies -- made a frame ago: they have
      arrived -- updates the capacities and, where a count exceeded its capacity, moves the missing runs and decodes that
      shard again (both sides know: the comparison is the same on both).
    """

    SLACK_NUM, SLACK_DEN, EXTRA = 5, 4, 64  # capacity = count * 5/4 + 64 runs, at most every run of the shard

    def __init__(self):
        self.cap = {}        # rank -> runs in the next payload message
        self.pending = []    # [{"counts": {rank: host tensor}, "event", "cap": {rank: int}, ...}]
        self.frames = 0      # frames gathered through this state
        self.blocking = 0    # ... of which with the blocking (first-frame) protocol
        self.overflows = 0   # ... shards repaired at confirmation
        self._recv = {}      # dst: (rank, parity) -> (fixed, payload at full capacity)

    @classmethod
    def next_cap(cls, count: int, runs: int) -> int:
        return min(int(runs), int(count) * cls.SLACK_NUM // cls.SLACK_DEN + cls.EXTRA)

    def recv_buffers(self, r: int, parity: int, fixed_bytes: int, runs: int, like: torch.Tensor):
        key = (r, parity & 1)
        buf = self._recv.get(key)
        if buf is None or buf[0].numel() != fixed_bytes or buf[1].shape[0] != runs or buf[1].dtype != like.dtype:
            buf = (torch.empty((fixed_bytes,), dtype=torch.uint8, device=like.device),
                   torch.zeros((runs, SPARSE_TILE, 3), dtype=like.dtype, device=like.device))
            self._recv[key] = buf
        return buf


def _count_to_host(fixed_heads: torch.Tensor):
    """The int64 counts at the head of fixed part(s) -> (host tensor, event or None), WITHOUT waiting: a non-blocking copy
    into page-locked memory with an event behind it on a HIP device, the values themselves on the CPU."""
    heads = fixed_heads.view(torch.int64).reshape(-1)
    if heads.is_cuda:
        host = torch.empty(heads.shape, dtype=torch.int64).pin_memory()
        host.copy_(heads, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(heads.device))
        return host, ev
    return heads.clone(), None


def _wait_all(ops):
    """One batched point-to-point group.  A rehearsal on a box with fewer GPUs than ranks runs ``gloo``, which moves host memory
    only: device tensors of the group are then staged through the host HERE, message by message (a synchronisation per message:
    a rehearsal of the protocol, not of its timing) -- everything else of the gather, encode / decode kernels included, runs as
    it does under RCCL."""
    if not ops:
        return
    staged = [op for op in ops if op.tensor.is_cuda and dist.get_backend(op.group) == "gloo"]
    if not staged:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        return
    host_ops, back = [], []
    for op in ops:
        if op.tensor.is_cuda:
            h = op.tensor.cpu() if op.op is dist.isend else torch.empty(op.tensor.shape, dtype=op.tensor.dtype)
            if op.op is dist.irecv:
                back.append((op.tensor, h))
            host_ops.append(dist.P2POp(op.op, h, op.peer, op.group))
        else:
            host_ops.append(op)
    for req in dist.batch_isend_irecv(host_ops):
        req.wait()
    for dev_t, h in back:
        dev_t.copy_(h)


def confirm_sparse(state: SparseGatherState, group=None, dst: int = 0) -> None:
    """Look at the counts of the frames gathered so far through ``state`` (their copies were started a frame ago): new
    capacities, and the repair of every shard whose count exceeded the capacity it travelled with.  Collective in the sense
    that every rank of the group calls it at the same point of its frame sequence (before the next gather / in finish)."""
    rank = dist.get_rank(group)

    def peer_of(r):
        return dist.get_global_rank(group, r) if group is not None else r

    while state.pending:
        f = state.pending.pop(0)
        if f["event"] is not None:
            f["event"].synchronize()  # (a copy enqueued a frame ago)
        if rank == dst:
            ops, redo = [], []
            for k, r in enumerate(f["remote"]):
                count, cap = int(f["counts"][k]), f["cap"][r]
                state.cap[r] = state.next_cap(count, f["runs"][r])
                if count > cap:  # the tail of its runs follows now
                    ops.append(dist.P2POp(dist.irecv, f["payload"][r][cap:count], peer_of(r), group))
                    redo.append(r)
                    last_gather["bytes"] = last_gather.get("bytes", 0) + f["payload"][r][cap:count].numel() * f["payload"][r].element_size()
            _wait_all(ops)
            for r in redo:
                state.overflows += 1
                decode_sparse(f["fixed"][r], f["payload"][r], f["npx"][r], f["out"].dtype, frame=f["out"],
                              row_block=f["row_block"], world=f["world"], rank=r)
        else:
            count, cap = int(f["counts"][0]), f["cap"][rank]
            state.cap[rank] = state.next_cap(count, f["runs"][rank])
            if count > cap:
                state.overflows += 1
                _wait_all([dist.P2POp(dist.isend, f["payload"][cap:count], peer_of(dst), group)])
    last_gather["confirmed"] = True


def _gather_sparse_async(local, height, row_block, world, rank, group, dst, out, nrows, state, clock=None):
    """The sparse gather without a host round trip: both messages of every remote rank in ONE batched group, the payload cut at
    the capacity ``state`` holds for that rank.  -> False when a capacity is still unknown (the caller then runs the blocking
    protocol, which also seeds the capacities)."""
    W = local.shape[1]
    esize = local.element_size()
    remote = [r for r in range(world) if r != dst and nrows[r] > 0]
    mine = [r for r in remote if r == rank] if rank != dst else remote
    if any(r not in state.cap for r in mine):
        return False
    parity = state.frames & 1
    runs = {r: (nrows[r] * W + SPARSE_TILE - 1) // SPARSE_TILE for r in remote}

    def peer_of(r):
        return dist.get_global_rank(group, r) if group is not None else r

    if rank == dst:
        fixed, payload, ops = {}, {}, []
        for r in remote:
            fixed[r], payload[r] = state.recv_buffers(r, parity, sparse_fixed_bytes(nrows[r] * W, esize), runs[r], local)
            ops.append(dist.P2POp(dist.irecv, fixed[r], peer_of(r), group))
            if state.cap[r] > 0:
                ops.append(dist.P2POp(dist.irecv, payload[r][: state.cap[r]], peer_of(r), group))
        _wait_all(ops)
        _mark(clock, "transfer_ms")
        if nrows[dst] > 0:
            place_shard(out, local, height, row_block, world, dst)
        if remote:
            decode_sparse_many([fixed[r] for r in remote], [payload[r] for r in remote], remote, out, row_block, world)
        _mark(clock, "decode_ms")
        counts, ev = _count_to_host(torch.stack([fixed[r][:8] for r in remote])) if remote else (torch.zeros(0, dtype=torch.int64), None)
        state.pending.append({"counts": counts, "event": ev, "remote": remote, "cap": {r: state.cap[r] for r in remote}, "runs": runs,
                              "fixed": fixed, "payload": payload, "npx": {r: nrows[r] * W for r in remote}, "out": out,
                              "row_block": row_block, "world": world})
        # (not final until confirm_sparse has looked at the counts: a shard whose count exceeded its capacity is repaired there,
        #  and the repair's bytes are added to `bytes`)
        last_gather.update(bytes=sum(fixed[r].numel() + state.cap[r] * SPARSE_TILE * 3 * esize for r in remote), sparse=True, confirmed=False)
    elif nrows[rank] > 0:
        fixed, payload = encode_sparse_full(local[: nrows[rank]].contiguous(), parity)
        _mark(clock, "encode_ms")  # (classify + scan + pack: no read-back)
        cap = state.cap[rank]
        ops = [dist.P2POp(dist.isend, fixed, peer_of(dst), group)]
        if cap > 0:
            ops.append(dist.P2POp(dist.isend, payload[:cap], peer_of(dst), group))
        _wait_all(ops)
        _mark(clock, "transfer_ms")
        counts, ev = _count_to_host(fixed[:8])
        state.pending.append({"counts": counts, "event": ev, "cap": {rank: cap}, "runs": runs, "payload": payload, "fixed": fixed})
    state.frames += 1
    return True


def _gather_sparse(local, height, row_block, world, rank, group, dst, out, nrows, clock=None, state=None):
    """Point-to-point gather of sparse shards.  With a ``state`` that knows every capacity: both messages in one group, no
    host round trip (``_gather_sparse_async``).  Otherwise (the first frame of a loop, or no loop at all) the blocking
    protocol: the fixed parts first (sizes known from the partition), then -- once ``dst`` has read the counts -- the runs
    that are not constant."""
    W = local.shape[1]
    esize = local.element_size()
    if state is not None:
        confirm_sparse(state, group, dst)  # (the frame before: its counts arrived long ago)
        if _gather_sparse_async(local, height, row_block, world, rank, group, dst, out, nrows, state, clock):
            return
        state.frames += 1
        state.blocking += 1

    def peer_of(r):
        return dist.get_global_rank(group, r) if group is not None else r

    if rank == dst:
        remote = [r for r in range(world) if r != dst and nrows[r] > 0]
        fixed = {r: torch.empty((sparse_fixed_bytes(nrows[r] * W, esize),), dtype=torch.uint8, device=local.device) for r in remote}
        _wait_all([dist.P2POp(dist.irecv, fixed[r], peer_of(r), group) for r in remote])
        counts = {}
        if remote:
            heads = torch.stack([fixed[r][:8] for r in remote]).view(torch.int64).reshape(-1).cpu()  # (one synchronisation)
            counts = {r: int(heads[k]) for k, r in enumerate(remote)}
        _mark(clock, "transfer_ms")  # (fixed parts in, counts read back: includes waiting for the remote ranks' encode)
        payload = {r: torch.empty((counts[r], SPARSE_TILE, 3), dtype=local.dtype, device=local.device) for r in remote}
        _wait_all([dist.P2POp(dist.irecv, payload[r], peer_of(r), group) for r in remote if counts[r] > 0])
        _mark(clock, "transfer_ms")
        if nrows[dst] > 0:
            place_shard(out, local, height, row_block, world, dst)
        if remote:
            decode_sparse_many([fixed[r] for r in remote], [payload[r] for r in remote], remote, out, row_block, world)
        _mark(clock, "decode_ms")
        last_gather.update(bytes=sum(fixed[r].numel() + payload[r].numel() * esize for r in remote), sparse=True)
        if state is not None:  # the capacities the next frames travel with
            for r in remote:
                state.cap[r] = state.next_cap(counts[r], (nrows[r] * W + SPARSE_TILE - 1) // SPARSE_TILE)
    elif nrows[rank] > 0:
        fixed, payload = encode_sparse(local[: nrows[rank]].contiguous())
        _mark(clock, "encode_ms")  # (classify + scan + pack, and the count's read-back)
        if state is not None:
            state.cap[rank] = state.next_cap(payload.shape[0], (nrows[rank] * W + SPARSE_TILE - 1) // SPARSE_TILE)
        _wait_all([dist.P2POp(dist.isend, fixed, peer_of(dst), group)])
        if payload.shape[0] > 0:
            _wait_all([dist.P2POp(dist.isend, payload, peer_of(dst), group)])
        _mark(clock, "transfer_ms")


def gather_image(local: torch.Tensor, height: int, row_block: int, group=None, dst: int = 0,
                 out: Optional[torch.Tensor] = None, staging: Optional[torch.Tensor] = None,
                 sparse: Optional[bool] = None, clock: Optional[PhaseClock] = None,
                 state: Optional[SparseGatherState] = None) -> Optional[torch.Tensor]:
    """Assemble the frame on ``dst`` from the ranks' compact row shards.

    ``local`` is this rank's ``[>= rows_of_this_rank, W, 3]`` shard (rows beyond its own are ignored).
    ``staging`` (``dst`` only, optional): a ``[world, max_shard_rows, W, 3]`` buffer the remote shards land in.
    ``state`` (a frame loop's ``SparseGatherState``): the sparse gather then moves both of a rank's messages in one group
    without reading a count back; the returned image is then PROVISIONAL -- ``last_gather["confirmed"]`` is False -- until
    ``confirm_sparse(state)`` has run (the next gather and ``ShardedFrameLoop.finish`` do; ``ShardedFrameLoop.image()`` only
    hands out confirmed frames): a shard whose run count exceeded the capacity it travelled with is repaired there, and the
    repair's bytes are added to ``last_gather["bytes"]``.  Returns the ``[H, W, 3]`` image on ``dst`` (``out`` when given) and ``None`` elsewhere."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1:
        rows = shard_rows(height, row_block, 1, 0)
        return local[: len(rows)]
    transport = choose_transport(group)  # (collective on first use: every rank gets here, with or without rows)
    # rehearsal on a box with fewer GPUs than ranks (PT_DIST_BACKEND=gloo): gloo moves host memory only
    if sparse is None:
        sparse = sparse_default()  # (the same on every rank: an argument or the environment of the job)
    # (a loop's sparse gather stages its MESSAGES through the host under gloo, see _wait_all; every other path the whole shard)
    staged = dist.get_backend(group) == "gloo" and local.is_cuda and not (state is not None and sparse and transport == P2P)
    if staged:
        dev_out, dev_local = out, local
        local = dev_local.cpu()
        out = torch.empty((height,) + tuple(local.shape[1:]), dtype=local.dtype) if rank == dst else None
        staging = None
    if rank == dst and out is None:
        out = torch.empty((height,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    nrows = [len(shard_rows(height, row_block, world, r)) for r in range(world)]
    if transport == PADDED:
        _gather_padded(local, height, row_block, world, rank, group, dst, out)
        _mark(clock, "transfer_ms")
    elif sparse:
        _gather_sparse(local, height, row_block, world, rank, group, dst, out, nrows, clock, None if staged else state)
    elif rank == dst:
        if staging is None:
            staging = torch.empty((world, max(nrows)) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        ops = []
        for r in range(world):
            if r != dst and nrows[r] > 0:
                peer = dist.get_global_rank(group, r) if group is not None else r
                ops.append(dist.P2POp(dist.irecv, staging[r, : nrows[r]], peer, group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()  # (RCCL: returns once the group is enqueued on the current stream)
        _mark(clock, "transfer_ms")
        for r in range(world):
            if nrows[r] > 0:
                place_shard(out, local if r == dst else staging[r], height, row_block, world, r)
        _mark(clock, "decode_ms")
        last_gather.update(bytes=sum(nrows[r] for r in range(world) if r != dst) * local[0].numel() * local.element_size(), sparse=False)
    elif nrows[rank] > 0:
        peer = dist.get_global_rank(group, dst) if group is not None else dst
        for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, local[: nrows[rank]], peer, group)]):
            req.wait()
        _mark(clock, "transfer_ms")
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

    ``scene`` is a scene handle or a list of n of them (the same scene uploaded n times): frame ``i`` is rendered by
    handle ``i % n`` on that handle's own stream, so up to n frames are in flight on the GPU -- a rank's share of a frame
    is short and latency-bound, and only another frame can fill what it leaves idle (``pytracer_amd/pipeline.py``).  The
    gather runs n frames BEHIND the render: ``step(i)`` enqueues the render of frame ``i`` and then gathers frame
    ``i - n`` on the comm stream -- the sparse gather reads a count back to the host, and the host should wait for that
    while the GPU has work, not before it has been given it.  ``finish()`` gathers the frames still waiting and drains
    the streams."""

    def __init__(self, scene, cam: abi.Camera, params: abi.Params, group=None, row_block: int = 8,
                 device: Optional[torch.device] = None, solo: bool = False, sparse: Optional[bool] = None):
        """``solo``: this process renders the WHOLE frame by itself, whatever process group exists (the one-GPU
        reference loop of the sharded bench)."""
        self.scenes = list(scene) if isinstance(scene, (list, tuple)) else [scene]
        self.scene, self.cam = self.scenes[0], cam
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() and not solo else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() and not solo else 0
        self.row_block = row_block
        self.params = abi.copy_params(params, row_block=row_block, n_ranks=self.world, rank=self.rank)
        self.height, self.width = params.height, params.width
        self.rows = len(shard_rows(self.height, row_block, self.world, self.rank))
        dt = torch.float32 if params.out_format == abi.OUT_F32 else torch.float64
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        n = len(self.scenes)
        self.lag = n if self.world > 1 else 0
        nb = n + 2 if self.world > 1 else max(2, n)
        self.bufs = [torch.zeros((max(self.rows, 1), self.width, 3), dtype=dt, device=self.device) for _ in range(nb)]
        self.nbytes = self.rows * self.width * 3 * self.bufs[0].element_size()
        # dedicated non-blocking streams: launches on the legacy default stream serialise the host with
        # the device (measured: ~51 us/launch on the null stream vs ~4 us on a side stream)
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(n)]
        self.stream = self.streams[0]
        self.comm = torch.cuda.Stream() if self.world > 1 else None
        self.full = None
        self.staging = None
        if self.world > 1 and self.rank == 0:
            self.full = [torch.empty((self.height, self.width, 3), dtype=dt, device=self.device) for _ in range(nb)]
            # (whole shards only) the remote shards of a frame land here; the comm stream runs receive -> placement
            # copies -> next receive in order, so one staging buffer serves all frame buffers
            self.staging = torch.empty((self.world, max_shard_rows(self.height, row_block, self.world), self.width, 3),
                                       dtype=dt, device=self.device)
        if self.world > 1:
            choose_transport(group)  # collective, before the first frame
        self._free = [None] * nb      # event: the gather that last read buffer b is done
        self._rendered = [None] * nb  # event: the render that last wrote buffer b is done ...
        self._rendered_on = [0] * nb  # ... and the stream slot it ran on
        self._assembled = self.world == 1  # image(): the last frame rendered has been gathered
        self._pending = []            # buffers whose frames are rendered (or being rendered) and not gathered yet, oldest first
        self.sparse = sparse_default() if sparse is None else bool(sparse)
        # the sparse gather's memory between frames: capacities agreed without a message, frames awaiting confirmation
        self.gstate = SparseGatherState() if (self.world > 1 and self.sparse) else None
        self.gather_bytes = None      # rank 0: what the remote ranks sent for the last gathered frame
        self.last = 0
        self._count = 0

    def step(self, i: int, gather: bool = True) -> None:
        """Enqueue the render of the next frame into one of this rank's buffers; with ``gather`` also assemble a frame on
        rank 0 -- the one ``lag`` frames before this (the last ones are assembled by ``finish``).  ``i`` is ignored but for
        documentation: frames are numbered by the calls."""
        k = self._count
        self._count += 1
        slot, b = k % len(self.scenes), k % len(self.bufs)
        if self._pending and not gather:  # (frames still waiting to be assembled must not be overwritten)
            self._drain()
        st = self.streams[slot]
        if self._free[b] is not None:
            st.wait_event(self._free[b])
            self._free[b] = None
        # the frame that last wrote this buffer ran on another stream when buffers and streams do not divide evenly
        # (three in flight: five buffers): it must have finished
        if self._rendered[b] is not None and self._rendered_on[b] != slot:
            st.wait_event(self._rendered[b])
        self.scenes[slot].render_into(self.cam, self.params, self.bufs[b].data_ptr(), self.nbytes, st.cuda_stream)
        self.last = b
        if self.world > 1 or len(self.streams) > 1:
            rendered = torch.cuda.Event()
            rendered.record(st)
            self._rendered[b], self._rendered_on[b] = rendered, slot
        self._assembled = self.world == 1
        if self.world > 1 and gather:
            self._pending.append(b)
            if len(self._pending) > self.lag:
                self._gather(self._pending.pop(0))

    def _drain(self) -> None:
        while self._pending:
            self._gather(self._pending.pop(0))

    def _gather(self, b: int) -> None:
        with torch.cuda.stream(self.comm):
            if self._rendered[b] is not None:
                self.comm.wait_event(self._rendered[b])
            gather_image(self.bufs[b], self.height, self.row_block, group=self.group, dst=0,
                         out=self.full[b] if self.rank == 0 else None,
                         staging=self.staging if self.rank == 0 else None, sparse=self.sparse, state=self.gstate)
            if self.rank == 0:
                self.gather_bytes = last_gather.get("bytes")
            done = torch.cuda.Event()
            done.record(self.comm)
            self._free[b] = done
        if not self._pending:
            self._assembled = True

    def finish(self) -> None:
        self._drain()
        if self.gstate is not None and self.gstate.pending:  # the last frames' counts: capacities, and a repair where one overflowed
            with torch.cuda.stream(self.comm):
                confirm_sparse(self.gstate, self.group, 0)
        torch.cuda.synchronize()

    PHASES = ("render_ms", "encode_ms", "transfer_ms", "decode_ms")

    def phase_probe(self, frames: int = 4) -> dict:
        """Where a sharded frame's time goes ON THIS RANK, phases run one after the other (``PhaseClock``: a measurement
        mode, nothing overlaps): ``render_ms`` this rank's rows; then -- behind a barrier, so that no rank's transfer
        waits for another's render -- ``encode_ms`` (remote ranks: the sparse encode with its count read-back),
        ``transfer_ms`` (remote: the sends; rank 0: until every shard is in, i.e. the slowest remote rank's encode plus
        the wire) and ``decode_ms`` (rank 0: its own placement and the one-launch decode of the remote shards).
        -> mean ms per frame and phase.  Collective: every rank of the group calls it."""
        self.finish()
        total = {k: 0.0 for k in self.PHASES}
        st = self.streams[0]
        for _ in range(max(1, frames)):
            if self.world > 1:
                dist.barrier(self.group)
            clock = PhaseClock()
            self.scenes[0].render_into(self.cam, self.params, self.bufs[0].data_ptr(), self.nbytes, st.cuda_stream)
            clock.mark("render_ms")
            if self.world > 1:
                dist.barrier(self.group)
                clock.restart()
                with torch.cuda.stream(self.comm):
                    gather_image(self.bufs[0], self.height, self.row_block, group=self.group, dst=0,
                                 out=self.full[0] if self.rank == 0 else None,
                                 staging=self.staging if self.rank == 0 else None, sparse=self.sparse, clock=clock, state=self.gstate)
                    if self.gstate is not None:
                        confirm_sparse(self.gstate, self.group, 0)
                torch.cuda.synchronize()
            for k in self.PHASES:
                total[k] += clock.ms.get(k, 0.0)
        # the probe rendered into buffer 0 and (in a process group) assembled that frame into full[0]; everything has been
        # drained, so the loop's bookkeeping is reset to exactly that state (image() used to raise here after a
        # step(gather=False), and a later step could wait on events of frames long gone)
        torch.cuda.synchronize()
        self.last = 0
        self._assembled = True
        self._pending = []
        self._free = [None] * len(self.bufs)
        self._rendered = [None] * len(self.bufs)
        return {k: v / max(1, frames) for k, v in total.items()}

    def image(self) -> Optional[torch.Tensor]:
        """The last frame, assembled (rank 0; ``[H, W, 3]`` in HBM; ``None`` on the other ranks).  Raises when the last
        frame was rendered without a gather (``step(gather=False)`` in a process group): there is no assembled frame
        then, and handing out an older buffer would be a stale image."""
        if self.world == 1:
            return self.bufs[self.last][: self.rows]
        if not self._assembled:
            raise RuntimeError("ShardedFrameLoop.image(): the last frame was not gathered (step(gather=False), or finish() not called)")
        return self.full[self.last] if self.rank == 0 else None

    def close(self) -> None:
        """Drain the streams and release the sparse encoder's payload buffers of this loop's comm stream."""
        self.finish()
        if self.comm is not None:
            release_sparse_scratch(self.comm)
