#!/usr/bin/env python3
"""VERDICT r5 next 6 -- would a PER-REGION list of ball groups bound the first bounce of C4's second pass?

    python tools/region_reach.py            (CPU only: numpy on the synthetic C4 scene, no GPU, no oracle)

A scattered ray has tmax = inf (materials.py:132-152: tmin 1e-3, no far end), so nothing is "out of reach" by distance: the only
geometric bound a start SURFACE gives is the half space its BRDF scatters into -- a ball lying entirely below the tangent plane
at the start point cannot be met.  For every sphere A of the scene the script samples points of its camera-facing side and counts
  per point  : balls entirely below that point's tangent plane (what a per-RAY test could skip: the upper bound of any list),
  per region : balls below the tangent planes of EVERY sampled point within one 8x8-pixel region's footprint on A
               (what a list made once per region could skip),
and the share of 64-ball / 8-ball Morton groups that a region's list could drop whole.  Printed as the frame's averages, weighted by the
number of flagged pixels each sphere covers (its projected area)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytracer_amd import flatten, scenes  # noqa: E402


def main():
    W, H = 3840, 2160
    flat = flatten.flatten_world(scenes.synthetic_world(256, wide=True))
    n = flat.n_shapes
    m = np.asarray(flat.m).reshape(12, n)
    c = np.stack([m[3], m[7], m[11]], axis=1)[1:]   # centres (the sky dome, shape 0, is not a ball of the filter)
    r = m[0][1:].copy()                              # uniform scale = radius
    cam = np.array([-2.0, 0.0, 1.0])                 # PerspectiveCamera(d = 1) at translation(-1, 0, 1): origin (-d, 0, 0) moved
    pix = 2.0 / H                                    # a pixel's extent on the screen plane at distance 1 (v spans 2 over H rows)
    rng = np.random.default_rng(7)
    # Morton-ish groups of 64 as the library cuts them: here simply 4 chunks by x (the exact order does not matter for a share)
    def morton(q):  # 10 bits per axis, interleaved (the library sorts its ball tables this way at upload)
        qi = ((q - q.min(axis=0)) / (np.ptp(q, axis=0) + 1e-12) * 1023).astype(np.int64)
        code = np.zeros(len(q), dtype=np.int64)
        for b in range(10):
            for ax in range(3):
                code |= ((qi[:, ax] >> b) & 1) << (3 * b + ax)
        return code

    order = np.argsort(morton(c))
    group = np.empty(len(c), dtype=int)
    group[order] = np.arange(len(c)) // 64
    group8 = np.empty(len(c), dtype=int)
    group8[order] = np.arange(len(c)) // 8
    w_sum = 0.0
    acc = dict(point=0.0, region=0.0, groups=0.0, groups8=0.0, regions_per_sphere=0.0)
    for a in range(len(c)):
        to_cam = cam - c[a]
        dist = np.linalg.norm(to_cam)
        u = to_cam / dist
        # points of the camera-facing hemisphere (visible side), uniform on the sphere cap n.u > r/dist
        pts = rng.normal(size=(4000, 3))
        pts /= np.linalg.norm(pts, axis=1, keepdims=True)
        pts = pts[pts @ u > r[a] / dist]
        # projected footprint: angular radius r/dist -> pixels
        rad_px = (r[a] / dist) / pix
        weight = np.pi * rad_px ** 2
        others = np.arange(len(c)) != a
        v = c[others] - c[a]                                     # [m, 3]
        rj = r[others]
        # ball j entirely below the tangent plane at normal n:  n.v + rj < r_a   (n.(c_j - P) + r_j < 0 with P = c_a + r_a n)
        below = pts @ v.T + rj[None, :] < r[a]                   # [p, m]
        acc["point"] += weight * below.mean()
        # a region: the points whose projection falls into one 8x8-pixel cell of the sphere's footprint
        e1 = np.cross(u, [0.0, 0.0, 1.0])
        e1 /= np.linalg.norm(e1)
        e2 = np.cross(u, e1)
        px = (pts @ e1) * rad_px
        py = (pts @ e2) * rad_px
        cell = (np.floor(px / 8.0).astype(int), np.floor(py / 8.0).astype(int))
        keys = cell[0] * 10007 + cell[1]
        shares, gshares, g8shares = [], [], []
        for k in np.unique(keys):
            sel = keys == k
            if sel.sum() < 8:
                continue
            all_below = below[sel].all(axis=0)
            shares.append(all_below.mean())
            g = group[others]
            gshares.append(np.mean([all_below[g == q].all() for q in range(4) if (g == q).any()]))
            g8 = group8[others]
            g8shares.append(np.mean([all_below[g8 == q].all() for q in np.unique(g8)]))
        if shares:
            acc["region"] += weight * float(np.mean(shares))
            acc["groups"] += weight * float(np.mean(gshares))
            acc["groups8"] += weight * float(np.mean(g8shares))
            acc["regions_per_sphere"] += weight * len(shares)
        w_sum += weight
    print(f"C4 scene: {len(c)} balls, 3840x2160; averages weighted by a sphere's projected area (its flagged pixels)")
    print(f"  sphere footprint: median radius {np.median((r / np.linalg.norm(cam - c, axis=1)) / pix):.1f} px; 8x8-pixel regions per sphere (area-weighted mean): {acc['regions_per_sphere'] / w_sum:.1f}")
    print(f"  balls below the tangent plane of ONE start point (a per-ray test's upper bound): {100 * acc['point'] / w_sum:.1f} %")
    print(f"  balls below the tangent planes of EVERY start point of one region (a per-region list): {100 * acc['region'] / w_sum:.1f} %")
    print(f"  64-ball groups a per-region list could drop whole: {100 * acc['groups'] / w_sum:.1f} %")
    print(f"  8-ball groups (the filter's unit: one round of scalar loads) a per-region list could drop whole: {100 * acc['groups8'] / w_sum:.1f} %")


if __name__ == "__main__":
    main()
