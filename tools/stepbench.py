#!/usr/bin/env python3
"""Where does a bench step go?  Enqueue (host) time vs GPU time, default stream vs a side stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytracer_amd import abi, flatten, scenes
from pytracer_amd.device import DeviceScene

W, H = 1280, 720
flat = flatten.flatten_world(scenes.synthetic_world(32, with_plane=True))
cam = flatten.flatten_camera(scenes.synthetic_camera(W, H))
par = abi.make_params(W, H, abi.RENDERER_FLAT, out_format=abi.OUT_F32)
ds = DeviceScene(flat)
out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
nb = out.numel() * 4
ds.set_count_rays(False)
for name, stream in (("default", torch.cuda.current_stream()), ("side", torch.cuda.Stream()), ("lib-own(sync)", None)):
    for timing in (True, False):
        ds.set_timing(timing)
        sp = stream.cuda_stream if stream is not None else None
        ptr = out.data_ptr()
        for _ in range(20):
            ds.render_into(cam, par, ptr, nb, sp)
        torch.cuda.synchronize()
        K = 500
        t0 = time.perf_counter()
        for _ in range(K):
            ds.render_into(cam, par, ptr, nb, sp)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        ds.sync()
        t2 = time.perf_counter()
        print(f"{name:14s} timing={timing}: enqueue {1e6 * (t1 - t0) / K:6.1f} us/step, total {1e6 * (t2 - t0) / K:6.1f} us/step")
