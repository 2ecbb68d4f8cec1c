#!/usr/bin/env python3
"""C2 frame: survivors of the conservative cull per 16x16 tile and per 8x8 quadrant of it, and how many of them a pixel's ray
really hits (diagnostics for pt_tile4_kernel: would per-quadrant survivor masks pay?).  Needs a GPU (device.cull_probe)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pytracer_amd import abi, flatten, scenes  # noqa: E402
from pytracer_amd.device import DeviceScene  # noqa: E402


def main():
    W, H = 1280, 720
    flat = flatten.flatten_world(scenes.synthetic_world(32, with_plane=True))
    cam = flatten.flatten_camera(scenes.synthetic_camera(W, H))
    ds = DeviceScene(flat)
    rows = []
    for ty in range(H // 16):
        t_s, q_s = [], []
        for tx in range(0, W // 16, 2):  # every other tile column
            keep = ds.cull_probe(cam, W, H, tx * 16, tx * 16 + 16, ty * 16, ty * 16 + 15)
            n = int(keep.sum())
            qs = []
            for qy in range(2):
                for qx in range(2):
                    k = ds.cull_probe(cam, W, H, tx * 16 + 8 * qx, tx * 16 + 8 * qx + 8, ty * 16 + 8 * qy, ty * 16 + 8 * qy + 7)
                    qs.append(int((k & keep).sum()))
            t_s.append(n)
            q_s.append(np.mean(qs))
        rows.append((ty, np.mean(t_s), np.max(t_s), np.mean(q_s)))
        print(f"tile row {ty:2d}: survivors per 16x16 tile mean {np.mean(t_s):5.1f} max {np.max(t_s):3d}; per 8x8 quadrant mean {np.mean(q_s):5.1f}"
              f"  -> ray-shape tests per tile {4 * np.mean(t_s):6.1f} -> {np.sum(4 * 0 + np.mean(q_s) * 4):6.1f}", flush=True)
    ds.close()


if __name__ == "__main__":
    main()
