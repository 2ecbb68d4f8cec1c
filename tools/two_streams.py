#!/usr/bin/env python3
"""Frames of C2 (1280x720, 32 spheres + plane, Flat) back to back on ONE stream against alternating on TWO (two scene
handles, two output buffers): how much of a launch's fixed cost (dispatch, ramp, drain) the next frame can hide.

    python tools/two_streams.py [frames]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pytracer_amd import abi, flatten, scenes  # noqa: E402
from pytracer_amd.device import DeviceScene  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 400
W, H = 1280, 720
flat = flatten.flatten_world(scenes.synthetic_world(32, with_plane=True))
cam = flatten.flatten_camera(scenes.synthetic_camera(W, H))
par = abi.make_params(W, H, abi.RENDERER_FLAT, out_format=abi.OUT_F32)
for n_streams in (1, 2, 3, 4):
    dss = [DeviceScene(flat) for _ in range(n_streams)]
    outs = [torch.empty((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(n_streams)]
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    for ds in dss:
        ds.set_count_rays(False)
        ds.set_timing(False)
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            j = i % n_streams
            dss[j].render_into(cam, par, outs[j].data_ptr(), outs[j].numel() * 4, streams[j].cuda_stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    same = all(torch.equal(outs[0], o) for o in outs)
    print(f"{n_streams} stream(s): {dt / K * 1e6:.2f} us per frame, {W * H * K / dt / 1e9:.1f} Gray/s, frames identical: {same}")
    del dss
