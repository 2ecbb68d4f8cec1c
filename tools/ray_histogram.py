#!/usr/bin/env python3
"""Rays per pixel of a num_of_rays > 1 frame, from the oracle (diagnostics: what the heaviest pixel of a frame costs
against the average one).  usage: ray_histogram.py c2n10|c3n10|demo10"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pytracer_amd import abi, flatten, scenes  # noqa: E402
from oracle import oracle  # noqa: E402
import importlib.util  # noqa: E402

spec = importlib.util.spec_from_file_location("kbench", os.path.join(os.path.dirname(os.path.abspath(__file__)), "kbench.py"))


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c2n10"
    import re
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "kbench.py")).read()
    ns = {}
    exec(src[src.index("CONFIGS = {"):src.index("def main")], {"abi": abi}, ns)
    nsph, plane, wide, W, H, kw = ns["CONFIGS"][name]
    kw = dict(kw)
    if nsph == "demo":
        world, cam_o = scenes.demo_world(clock=150.0)
        cam = flatten.flatten_camera(cam_o)
    else:
        world = scenes.synthetic_world(nsph, with_plane=plane, wide=wide)
        cam = flatten.flatten_camera(scenes.synthetic_camera(W, H))
    flat = flatten.flatten_world(world)
    par = abi.make_params(W, H, out_format=abi.OUT_F32, **kw)
    img = np.zeros((H, W), dtype=np.uint32)
    oracle.lib().pto_set_ray_image.argtypes = [C.c_void_p]
    oracle.lib().pto_set_ray_image(img.ctypes.data_as(C.c_void_p))
    _, n = oracle.render(flat, cam, par, n_threads=8, sqr_mode=oracle.SQR_MUL)
    oracle.lib().pto_set_ray_image(None)
    r = img.reshape(-1).astype(np.int64)
    assert r.sum() == n
    fl = r[r > 1]
    print(f"{name}: {W}x{H}, rays {n}, pixels with more than the primary ray {fl.size}")
    print(f"  rays per such pixel: mean {fl.mean():.1f}  p50 {np.percentile(fl, 50):.0f}  p90 {np.percentile(fl, 90):.0f}  "
          f"p99 {np.percentile(fl, 99):.0f}  p99.9 {np.percentile(fl, 99.9):.0f}  max {fl.max()}")
    edges = [2, 12, 25, 50, 100, 200, 400, 800, 1200]
    for lo, hi in zip(edges[:-1], edges[1:]):
        m = (fl >= lo) & (fl < hi)
        print(f"  {lo:5d} .. {hi:5d}: {m.sum():8d} pixels  {100.0 * fl[m].sum() / fl.sum():5.1f} % of the rays")
    rows = img.max(axis=1)
    print("  heaviest pixel per band of 60 rows:", [int(rows[i:i + 60].max()) for i in range(0, H, 60)])
    np.save(f"/tmp/rays_{name}.npy", img)


if __name__ == "__main__":
    main()
