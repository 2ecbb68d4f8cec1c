#!/usr/bin/env python3
"""A rank's share of the C4 frame (rank 3 of 8, PT_PCG_SAMPLE / PT_PCG_PIXEL) one frame after the other and with 2 / 3
frames in flight (pytracer_amd.pipeline.FramePipeline): how much of the share's fixed latency another frame hides.

    python tools/share_in_flight.py [n_ranks]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pytracer_amd import abi, dist as ptdist, flatten, scenes  # noqa: E402
from pytracer_amd.pipeline import FramePipeline  # noqa: E402
from tools.kbench import CONFIGS  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = min(3, world - 1)
ns, plane, wide, W, H, kw = CONFIGS["c4"]
flat = flatten.flatten_world(scenes.synthetic_world(ns, with_plane=plane, wide=wide))
cam = flatten.flatten_camera(scenes.synthetic_camera(W, H))
rows = len(ptdist.shard_rows(H, 8, world, rank))
K = 100
for mode, name in ((abi.PCG_SAMPLE, "SAMPLE"), (abi.PCG_PIXEL, "PIXEL")):
    par = abi.make_params(W, H, out_format=abi.OUT_F32, pcg_mode=mode, n_ranks=world, rank=rank, row_block=8, **kw)
    ref = None
    for n in (1, 2, 3):
        with FramePipeline(flat, n_in_flight=n) as pipe:
            pipe.set_count_rays(False)
            pipe.set_timing(False)
            outs = [torch.empty((rows, W, 3), dtype=torch.float32, device="cuda") for _ in range(n)]
            for i in range(2 * n):
                pipe.submit(cam, par, outs[i % n])
            pipe.wait()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(K):
                pipe.submit(cam, par, outs[i % n])
            pipe.wait()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if ref is None:
                ref = outs[0].clone()
            same = all(torch.equal(ref, o) for o in outs)
        print(f"rank {rank} of {world}, {name}: {n} in flight: {dt / K * 1e3:.4f} ms per frame, identical: {same}")
