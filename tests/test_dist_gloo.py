"""Multi-process tests of the pixel partition + gather on CPU (gloo, world_size 2 and 3).

The GPU kernel cannot run here, so each rank's local renderer is the CPU oracle (allowed: tests may
use the oracle as a stand-in); what is under test is pytracer_amd.dist — the interleaved row-block
partition and the gather (one point-to-point transfer per remote rank, then a strided placement into row-block order) — and the invariant that the assembled frame is
bit-identical to the single-process frame (per-pixel seeds depend only on the global pixel index)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, renderer, S, height, row_block, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as orc
        from pytracer_amd import abi, dist as ptdist, flatten, scenes

        flat = flatten.flatten_world(scenes.synthetic_world(8, with_plane=True))
        cam = flatten.flatten_camera(scenes.synthetic_camera(48, height))
        par = abi.make_params(48, height, renderer, samples_per_side=S, num_of_rays=2, max_depth=2,
                              path_state=45, path_seq=54)

        def render_local(p):
            out, _ = orc.render(flat, cam, p, n_threads=2, sqr_mode=orc.SQR_MUL)
            return torch.from_numpy(out)

        full = ptdist.render_sharded(render_local, par, row_block=row_block)
        if rank == 0:
            ref, _ = orc.render(flat, cam, par, n_threads=2, sqr_mode=orc.SQR_MUL)
            ret["ok"] = bool(full.numpy().tobytes() == ref.tobytes())
            ret["shape"] = tuple(full.shape)
        else:
            assert full is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,renderer,S,height,row_block", [
    (2, 1, 0, 27, 8),   # Flat, ragged last block, uneven shards (16 + 11 rows)
    (2, 2, 2, 24, 4),   # PathTracer with jitter: per-pixel seeds make shards reproducible
    (3, 0, 0, 10, 8),   # more ranks than full blocks: rank 2 owns nothing
])
def test_sharded_render_matches_single_process(world, renderer, S, height, row_block):
    from oracle import oracle as orc

    orc.build()
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, renderer, S, height, row_block, ret), nprocs=world, join=True)
    assert ret["ok"] is True
    assert ret["shape"] == (height, 48, 3)


def _gather_worker(rank, world, port, transport, H, ret, sparse=None, W=5, kind="ramp"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pytracer_amd import dist as ptdist

        if transport is not None:
            ptdist.choose_transport(force=transport)  # (else: the collective probe inside the first gather decides)
        if kind == "ramp":  # no two pixels alike
            full = torch.arange(H * W * 3, dtype=torch.float64).reshape(H, W, 3)
        else:  # a sky of one colour (with -0.0 in it) and a few patches of "geometry", fp32 as the frames are
            g = torch.Generator().manual_seed(5)
            full = torch.empty((H, W, 3), dtype=torch.float32)
            full[...] = torch.tensor([0.25, -0.0, 0.75])
            if kind == "patches":
                for _ in range(6):
                    r0, c0 = int(torch.randint(0, H, (1,), generator=g)), int(torch.randint(0, W, (1,), generator=g))
                    full[r0:r0 + 3, c0:c0 + 40] = torch.rand((min(3, H - r0), min(40, W - c0), 3), generator=g)
                full[H // 2, W // 2, 1] = 0.0  # differs from the sky's -0.0 in the sign bit only
        rows = ptdist.shard_rows(H, 8, world, rank)
        shard = full[rows].contiguous() if rows else torch.zeros((1, W, 3), dtype=full.dtype)
        out = ptdist.gather_image(shard, H, 8, sparse=sparse)
        if rank == 0:
            bits = torch.int64 if full.dtype == torch.float64 else torch.int32
            ret["ok"] = bool(torch.equal(out.view(bits), full.view(bits)))
            ret["transport"] = ptdist.choose_transport()
            ret["bytes"] = int(ptdist.last_gather.get("bytes", -1))
            ret["sparse"] = bool(ptdist.last_gather.get("sparse", False))
        else:
            assert out is None
        # a second frame through the same (now settled) transport
        out = ptdist.gather_image(shard + 1.0, H, 8, sparse=sparse)
        if rank == 0:
            ret["ok2"] = bool(torch.equal(out, full + 1.0))
        # a clocked gather (bench.py's per-phase breakdown): the same frame, and every rank books the phases it took part in
        clock = ptdist.PhaseClock()
        out = ptdist.gather_image(shard + 2.0, H, 8, sparse=sparse, clock=clock)
        if rank == 0:
            ret["ok3"] = bool(torch.equal(out, full + 2.0))
        ret[f"phases{rank}"] = sorted(clock.ms)
        assert all(v >= 0.0 for v in clock.ms.values())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,transport,H", [
    (2, "p2p", 27), (3, "p2p", 27),        # ragged: 8 + 8 + 8 + 3 rows over `world` ranks
    (2, "padded", 27), (3, "padded", 27),
    (3, None, 27),                          # nobody forces anything: the collective probe decides (gloo: p2p)
    (3, None, 10), (3, "padded", 10),      # rank 2 owns no rows and must still be in every collective
    (3, None, 50),                          # 2 complete groups of 3 blocks + a ragged tail block of rank 0
])
def test_gather_one_transfer_per_rank_and_the_padded_transport(world, transport, H):
    """Both transports behind gather_image: one batched point-to-point transfer per remote rank followed by the strided
    placement into row-block order, and the padded `gather` for a backend that refuses those; which one is used is
    agreed on collectively (`choose_transport`), also by a rank without rows."""
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_gather_worker, args=(world, port, transport, H, ret), nprocs=world, join=True)
    assert ret["ok"] is True and ret["ok2"] is True and ret["ok3"] is True
    assert ret["transport"] == (transport or "p2p")
    assert "transfer_ms" in ret["phases0"]
    if (transport or "p2p") == "p2p":
        assert "decode_ms" in ret["phases0"] and "transfer_ms" in ret["phases1"]


@pytest.mark.parametrize("world,H,W,kind,sparse", [
    (2, 27, 5, "ramp", False),        # shards whole, one message per rank (round 2's gather)
    (3, 50, 300, "patches", True),    # a sky and patches: runs of 128 pixels, most of them constant
    (3, 50, 300, "patches", False),
    (2, 64, 257, "sky", True),        # nothing but sky: no second message at all
    (3, 10, 300, "patches", True),    # rank 2 owns no rows
    (2, 27, 5, "ramp", True),         # no run is constant: everything travels, plus the fixed part
])
def test_sparse_gather_is_lossless_and_smaller(world, H, W, kind, sparse):
    """dist.encode_sparse / decode_sparse behind gather_image: runs of 128 pixels that are one colour to the bit (-0.0 is
    not 0.0) travel as one pixel, the frame on rank 0 is the frame bit for bit, and what crossed the wire is counted."""
    from pytracer_amd import dist as ptdist

    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_gather_worker, args=(world, port, "p2p", H, ret, sparse, W, kind), nprocs=world, join=True)
    assert ret["ok"] is True and ret["ok2"] is True
    assert ret["sparse"] is sparse
    esize = 8 if kind == "ramp" else 4
    dense = sum(len(ptdist.shard_rows(H, 8, world, r)) for r in range(1, world)) * W * 3 * esize
    if not sparse:
        assert ret["bytes"] == dense
    elif kind == "sky":
        assert ret["bytes"] == sum(ptdist.sparse_fixed_bytes(len(ptdist.shard_rows(H, 8, world, r)) * W, 4) for r in range(1, world))
        assert ret["bytes"] < dense // 20
    elif kind == "patches":
        assert ret["bytes"] < dense // 2
    else:  # everything travels: the runs (the last one of a shard filled up to 128 pixels) plus the fixed part
        worst = 0
        for r in range(1, world):
            npx = len(ptdist.shard_rows(H, 8, world, r)) * W
            worst += ptdist.sparse_fixed_bytes(npx, esize) + (npx + 127) // 128 * 128 * 3 * esize
        assert dense < ret["bytes"] == worst


def test_sparse_codec_round_trip():
    from pytracer_amd import dist as ptdist

    g = torch.Generator().manual_seed(1)
    for dtype, bits in ((torch.float32, torch.int32), (torch.float64, torch.int64)):
        for rows, W in ((1, 1), (3, 129), (8, 128), (5, 37), (2, 1000)):
            x = torch.empty((rows, W, 3), dtype=dtype)
            x[...] = torch.tensor([0.3, 0.5, float("nan")], dtype=dtype)  # (a NaN equals itself bitwise)
            if W > 9:
                x[rows // 2, 5:9] = torch.rand((4, 3), generator=g, dtype=dtype)
            fixed, payload = ptdist.encode_sparse(x)
            assert fixed.numel() == ptdist.sparse_fixed_bytes(rows * W, x.element_size())
            assert ptdist.sparse_count(fixed) == payload.shape[0] <= 2
            y = ptdist.decode_sparse(fixed, payload, rows * W, dtype).view(rows, W, 3)
            assert torch.equal(x.view(bits), y.contiguous().view(bits))
    with pytest.raises(RuntimeError):
        ptdist.decode_sparse(fixed, payload[:0], rows * W, dtype)
    # straight into a frame: the rows land where the partition puts them
    full = torch.rand((27, 40, 3), generator=g)
    full[:, :20] = 0.5
    got = torch.zeros_like(full)
    for r in range(3):
        rws = ptdist.shard_rows(27, 8, 3, r)
        f, p = ptdist.encode_sparse(full[rws].contiguous())
        ptdist.decode_sparse(f, p, len(rws) * 40, torch.float32, frame=got, row_block=8, world=3, rank=r)
    assert torch.equal(got, full)


def test_gather_plan_counts_transfers():
    from pytracer_amd import dist as ptdist

    plan = ptdist.gather_plan(2160, 3840, 8, 8, itemsize=4)
    assert plan["gather_ops_per_frame"] == 7  # <= 40 transfers per 4K frame at 8 ranks (was 236)
    assert plan["gather_bytes_per_frame"] == sum(plan["rows_per_rank"][1:]) * 3840 * 3 * 4
    assert sum(plan["rows_per_rank"]) == 2160
    assert ptdist.gather_plan(10, 48, 8, 3)["gather_ops_per_frame"] == 1  # rank 2 owns nothing: no transfer for it


def test_place_shard_is_the_inverse_of_the_partition():
    from pytracer_amd import dist as ptdist

    for H, rb, world in ((27, 8, 2), (50, 8, 3), (10, 8, 3), (2160, 8, 8), (721, 7, 3), (5, 8, 2), (64, 8, 8)):
        full = torch.arange(H * 2 * 3, dtype=torch.float64).reshape(H, 2, 3)
        out = torch.full_like(full, -1.0)
        for r in range(world):
            rows = ptdist.shard_rows(H, rb, world, r)
            if rows:
                ptdist.place_shard(out, full[rows].contiguous(), H, rb, world, r)
        assert torch.equal(out, full), (H, rb, world)


def test_partition_helpers():
    from pytracer_amd import dist as ptdist

    assert ptdist.shard_rows(20, 8, 2, 0) == list(range(0, 8)) + list(range(16, 20))
    assert ptdist.shard_rows(20, 8, 2, 1) == list(range(8, 16))
    assert ptdist.max_shard_rows(20, 8, 2) == 12
    assert ptdist.max_shard_rows(10, 8, 3) == 8
    # (first row in the frame, first row in the compact shard, rows) per block
    assert ptdist.shard_blocks(20, 8, 2, 0) == [(0, 0, 8), (16, 8, 4)]
    assert ptdist.shard_blocks(20, 8, 2, 1) == [(8, 0, 8)]
    assert ptdist.shard_blocks(10, 8, 3, 2) == []
    for h, rb, w in ((2160, 8, 8), (721, 7, 3), (5, 8, 2)):
        for r in range(w):
            rows = [g0 + k for g0, _, n in ptdist.shard_blocks(h, rb, w, r) for k in range(n)]
            assert rows == ptdist.shard_rows(h, rb, w, r)
            assert [l0 for _, l0, _ in ptdist.shard_blocks(h, rb, w, r)] == \
                [sum(n for _, _, n in ptdist.shard_blocks(h, rb, w, r)[:k]) for k in range(len(ptdist.shard_blocks(h, rb, w, r)))]


# ---- the frame LOOP's sparse gather: capacities agreed without a message, no count read back per frame (round 5) -----------
def _loop_worker(rank, world, port, H, W, ret, device="cpu"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pytracer_amd import dist as ptdist

        ptdist.choose_transport(force="p2p")
        g = torch.Generator().manual_seed(11)
        sky = torch.empty((H, W, 3), dtype=torch.float32)
        sky[...] = torch.tensor([0.25, -0.0, 0.75])
        dev = torch.device(device)
        if dev.type == "cuda":
            torch.cuda.set_device(dev)

        def frame(kind, k):
            f = sky.clone()
            if kind == "few":      # a handful of runs that are not constant: the steady state
                f[(7 * k) % H, :, 0] = torch.rand((W,), generator=g)
                f[H // 3: H // 3 + 2] = torch.rand((2, W, 3), generator=g)
            elif kind == "noise":  # EVERY run differs: far beyond any capacity derived from a `few` frame
                f = torch.rand((H, W, 3), generator=g)
            elif kind == "half":
                f[: H // 2] = torch.rand((H // 2, W, 3), generator=g)
            return f

        kinds = ["few", "few", "few", "noise", "noise", "few", "half", "sky", "few"]
        frames = [frame(kind, k) for k, kind in enumerate(kinds)]  # (same generator on every rank: the same frames)
        rows = ptdist.shard_rows(H, 8, world, rank)
        state = ptdist.SparseGatherState()
        outs = [torch.zeros((H, W, 3), dtype=torch.float32, device=dev) for _ in frames] if rank == 0 else [None] * len(frames)
        sent = []
        for k, f in enumerate(frames):
            shard = (f[rows].contiguous() if rows else torch.zeros((1, W, 3), dtype=f.dtype)).to(dev)
            out = ptdist.gather_image(shard, H, 8, sparse=True, out=outs[k], state=state)
            assert (out is outs[k]) if rank == 0 else out is None
            sent.append(int(ptdist.last_gather.get("bytes", -1)) if rank == 0 else 0)
        ptdist.confirm_sparse(state)  # (what ShardedFrameLoop.finish does: the last frame's counts, and its repair if need be)
        assert not state.pending
        if rank == 0:
            ret["ok"] = [bool(torch.equal(o.cpu().view(torch.int32), f.view(torch.int32))) for o, f in zip(outs, frames)]
            ret["sent"] = sent
        ret[f"stats{rank}"] = (state.frames, state.blocking, state.overflows)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_frame_loop_gather_on_the_device_under_gloo():
    """The same nine frames with the shards in HBM (two ranks sharing cuda:0, gloo carrying the MESSAGES through the host): the
    device encode without its read-back (pt_image_sparse_encode into a full-capacity payload), the asynchronous copy of the
    counts into page-locked memory, the one-launch decode out of full-capacity receive buffers and the single-shard decode of
    the overflow repair -- the pieces of the protocol a CPU run cannot reach."""
    ret = mp.Manager().dict()
    port = _free_port()
    mp.spawn(_loop_worker, args=(2, port, 64, 1280, ret, "cuda:0"), nprocs=2, join=True)
    assert ret["ok"] == [True] * 9, ret["ok"]
    assert ret["stats0"][:2] == (9, 1) and ret["stats0"][2] >= 1 and ret["stats1"][2] >= 1


@pytest.mark.parametrize("world,H,W", [(2, 64, 1280), (3, 50, 1280), (8, 2160, 96)])
def test_frame_loop_gather_needs_no_count_round_trip_and_repairs_overflows(world, H, W):
    """Nine frames through ONE ``SparseGatherState`` (pytracer_amd/dist.py): the first uses the blocking protocol (no history),
    every later one posts both messages of a rank in one group with the payload cut at a capacity derived -- on both sides,
    without a message -- from the frame before; a frame that suddenly is all noise overflows that capacity and is repaired
    when its counts are looked at (one frame later), the frame after it travels with a capacity that fits.  Every frame
    arrives bit for bit.  World 8 is the target machine's: 2 160 rows in 8-row blocks leave ranks 6 and 7 one block short
    (VERDICT r4 next 6)."""
    import math

    ret = mp.Manager().dict()
    port = _free_port()
    mp.spawn(_loop_worker, args=(world, port, H, W, ret), nprocs=world, join=True)
    assert ret["ok"] == [True] * 9, ret["ok"]
    frames, blocking, overflows = ret["stats0"]
    assert (frames, blocking) == (9, 1), "only the first frame may read counts back before sizing the payload"
    n_remote = sum(1 for r in range(1, world) if len(__import__("pytracer_amd.dist", fromlist=["x"]).shard_rows(H, 8, world, r)) > 0)
    # frame 3 (few -> noise) overflows on every remote rank; frame 6 (few -> half) on the ranks whose rows lie in the upper half
    assert overflows >= n_remote, (overflows, n_remote)
    for r in range(1, world):
        f, b, o = ret[f"stats{r}"]
        owns_rows = len(__import__("pytracer_amd.dist", fromlist=["x"]).shard_rows(H, 8, world, r)) > 0
        # exactly ONE blocking frame (the first) on every remote rank that owns rows; a rank without rows has nothing to block on
        assert (f, b) == (9, 1 if owns_rows else 0), (r, f, b, owns_rows)
    sent = ret["sent"]
    whole = sum(len(__import__("pytracer_amd.dist", fromlist=["x"]).shard_rows(H, 8, world, r)) for r in range(1, world)) * W * 12
    assert sent[1] < 0.6 * whole and sent[2] == sent[1], "steady frames travel with a capacity, far below the whole shards"
    assert sent[4] > sent[1], "the capacity follows the scene: the second noise frame is sent in one go"
    assert math.isfinite(sum(sent))
