#!/usr/bin/env python3
"""What the sparse gather's codec costs on the GPU (pytracer_amd/dist.py: encode_sparse on the sending rank,
decode_sparse + placement on rank 0) for a rank's share of the C4 frame, and what it saves in bytes.

    python tools/sparse_cost.py [n_ranks ...]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pytracer_amd import abi, dist as ptdist, flatten, scenes  # noqa: E402
from pytracer_amd.device import DeviceScene  # noqa: E402
from tools.kbench import CONFIGS  # noqa: E402

ns, plane, wide, W, H, kw = CONFIGS["c4"]
flat = flatten.flatten_world(scenes.synthetic_world(ns, with_plane=plane, wide=wide))
cam = flatten.flatten_camera(scenes.synthetic_camera(W, H))
ds = DeviceScene(flat)
for world in [int(a) for a in sys.argv[1:]] or [2, 4, 8]:
    rank = world - 1
    par = abi.make_params(W, H, out_format=abi.OUT_F32, pcg_mode=abi.PCG_SAMPLE, n_ranks=world, rank=rank, row_block=8, **kw)
    rows = len(ptdist.shard_rows(H, 8, world, rank))
    shard = torch.empty((rows, W, 3), dtype=torch.float32, device="cuda")
    ds.render_into(cam, par, shard.data_ptr(), shard.numel() * 4, None)
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        fixed, payload = ptdist.encode_sparse(shard)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        back = ptdist.decode_sparse(fixed, payload, rows * W, torch.float32).view(rows, W, 3)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    assert torch.equal(back.view(torch.int32), shard.view(torch.int32))
    # the kernels alone: 20 encodes / decodes between two events (the count read-back of encode included: it is part of it)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    for _ in range(20):
        ptdist.encode_sparse(shard)
    e[1].record()
    frame = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    for _ in range(20):  # (as rank 0 does it: straight into the frame)
        ptdist.decode_sparse(fixed, payload, rows * W, torch.float32, frame=frame, row_block=8, world=world, rank=rank)
    e[2].record()
    torch.cuda.synchronize()
    enc_us, dec_us = e[0].elapsed_time(e[1]) * 50, e[1].elapsed_time(e[2]) * 50
    dense = shard.numel() * 4
    sent = fixed.numel() + payload.numel() * 4
    print(f"{world} ranks: a shard of {rows} rows = {dense / 1e6:.1f} MB whole, {sent / 1e6:.2f} MB sparse ({payload.shape[0]} of "
          f"{(rows * W + 127) // 128} runs of 128 pixels are not one colour); encode {1e3 * (t1 - t0):.3f} ms, decode {1e3 * (t2 - t1):.3f} ms "
          f"(one call, host wall clock, synchronised); back to back {enc_us:.0f} / {dec_us:.0f} us per call; "
          f"rendering this share of the frame: {ds.stats().kernel_ms:.3f} ms")
