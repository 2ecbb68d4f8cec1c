#!/usr/bin/env python3
"""ONE C2 frame rendered as k concurrent launches (bands of rows, one stream and one scene handle each), frames strictly one
after the other (every band of frame i+1 waits for all bands of frame i): does a frame's fixed cost shrink when its
launch is split?

    python tools/split_frame.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pytracer_amd import abi, flatten, scenes  # noqa: E402
from pytracer_amd.device import DeviceScene  # noqa: E402

W, H, K = 1280, 720, 400
flat = flatten.flatten_world(scenes.synthetic_world(32, with_plane=True))
cam = flatten.flatten_camera(scenes.synthetic_camera(W, H))
ref = None
for k in (1, 2, 3, 5):
    rb = ((H + k - 1) // k + 15) // 16 * 16  # (multiples of 16 rows keep the 16x16-tile kernel; the last band is shorter)
    dss = [DeviceScene(flat) for _ in range(k)]
    streams = [torch.cuda.Stream() for _ in range(k)]
    for ds in dss:
        ds.set_count_rays(False)
        ds.set_timing(False)
    out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    pars = [abi.make_params(W, H, abi.RENDERER_FLAT, out_format=abi.OUT_F32, n_ranks=k, rank=r, row_block=rb) for r in range(k)]
    done = [None] * k
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            prev = list(done)
            for r in range(k):
                for e in prev:
                    if e is not None and k > 1:
                        streams[r].wait_event(e)
                band = out[r * rb:min(H, (r + 1) * rb)]
                dss[r].render_into(cam, pars[r], band.data_ptr(), band.numel() * 4, streams[r].cuda_stream)
                if k > 1:
                    done[r] = torch.cuda.Event()
                    done[r].record(streams[r])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    if ref is None:
        ref = out.clone()
    print(f"{k} band(s): {dt / K * 1e6:.2f} us per frame, identical: {torch.equal(ref, out)}")
    for ds in dss:
        ds.close()
