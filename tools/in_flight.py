#!/usr/bin/env python3
"""Frames per second of the BASELINE configurations with 1, 2 and 3 frames in flight (pytracer_amd.pipeline.FramePipeline:
one handle per slot on one uploaded scene, one stream each; wall clock between two device synchronisations).

    python tools/in_flight.py [c2 c3 c3:sample c4:sample c5 ...]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pytracer_amd import abi, flatten, scenes  # noqa: E402
from pytracer_amd.pipeline import FramePipeline  # noqa: E402
from tools.kbench import CONFIGS  # noqa: E402

for name in sys.argv[1:] or ["c2", "c3", "c3:sample", "c4", "c4:sample", "c5", "c3n10"]:
    ns, plane, wide, W, H, kw = CONFIGS[name.split(":")[0]]
    kw = dict(kw)
    kw.pop("lights", None)
    if name.endswith(":sample"):
        kw["pcg_mode"] = abi.PCG_SAMPLE
    flat = flatten.flatten_world(scenes.synthetic_world(ns, with_plane=plane, wide=wide))
    cam = flatten.flatten_camera(scenes.synthetic_camera(W, H))
    par = abi.make_params(W, H, out_format=abi.OUT_F32, **kw)
    K = 200 if W * H < 2_000_000 else 60
    ref, line = None, []
    for n in (1, 2, 3):
        with FramePipeline(flat, n_in_flight=n) as pipe:
            pipe.set_count_rays(False)
            pipe.set_timing(False)
            outs = [torch.empty((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(n)]
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
            assert all(torch.equal(ref, o) for o in outs), "frames differ"
        line.append(f"{n} in flight {dt / K * 1e3:.4f} ms")
    print(f"{name:10s} {W}x{H}: " + ", ".join(line) + " per frame (frames identical)")
