#!/usr/bin/env python3
"""How many rounds would PT_PCG_PIXEL's second pass need under other ways of guessing where a sample starts?  (A study,
not a test: run by hand, `python tests/sim_speculation.py`; it lives under tests/ because it asks the CPU oracle -- test
infrastructure -- for the number of draws of every sample of the pixels of C3's slowest work units.)

A pixel's samples share one generator (SURVEY.md 8c Mode PIXEL): sample k + 1 starts where sample k stopped, so the lanes
of a pixel guess the draws of the samples before theirs (csrc/pt_path.h: seed_round) and a round commits the samples
whose start state was right.  For every flagged pixel of three 8x8 regions where paths bounce between spheres
(profiles/r03_c3_pixel_tail_units.txt) this prints the rounds under
  chain L   -- lane j assumes the j samples before it each drew what the last committed sample drew (what ships for such pixels),
  tree2 L   -- sample vbase + d from every sum of d of the pixel's two most frequent counts,
  win L     -- sample vbase + d from EVERY integer offset in [d lo, d hi], lo / hi the extreme counts of the last eight samples,
with L lanes per pixel.  Result (profiles/r04_pixel_speculation_study.txt): with the 4 lanes per pixel a full C3 frame has
(29 358 flagged pixels on 131 072 resident lanes) no scheme beats the chain by more than a round or two out of 15; the
window needs 16 - 32 lanes per pixel to halve the rounds, i.e. 4 - 8 times the traced samples."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402
from pytracer_amd import abi, flatten, scenes  # noqa: E402

W, H, S = 1280, 720, 4
flat = flatten.flatten_world(scenes.synthetic_world(32))
cam = flatten.flatten_camera(scenes.synthetic_camera(W, H))
par = abi.make_params(W, H, abi.RENDERER_PATHTRACER, samples_per_side=S, num_of_rays=1, max_depth=3, rr_limit=3, path_state=45, path_seq=54)


def counts(col, row):
    """draws of each of the pixel's S*S samples (jitter included), in order"""
    g = orc.Pcg(45, 54 + row * W + col)
    out = []
    for sr in range(S):
        for sc in range(S):
            s0, inc = g.state, g.inc
            up, vp = (sc + g.random_float()) / S, (sr + g.random_float()) / S
            orc.radiance(flat, par, g, orc.tracer_fire_ray(cam, W, H, col, row, up, vp), 0)
            st, n = s0, 0
            while st != g.state:
                st = (st * 6364136223846793005 + inc) & ((1 << 64) - 1)
                n += 1
            out.append(n)
    return out


def chain(c, L):
    rounds, v, last = 0, 0, 4
    while v < len(c):
        rounds += 1
        k = 0
        while k < L and v + k < len(c) and all(x == last for x in c[v:v + k]):
            k += 1
        v += k
        last = c[v - 1]
    return rounds


def tree2(c, L):
    rounds, v, freq = 0, 0, {4: 1}
    while v < len(c):
        rounds += 1
        al = [k for k, _ in sorted(freq.items(), key=lambda kv: -kv[1])]
        if len(al) < 2:
            al = al + [al[0] + 2]
        lanes, d = [], 0
        while len(lanes) < L:
            for r in range(d + 1):
                if len(lanes) < L:
                    lanes.append((d, (d - r) * al[0] + r * al[1]))
            d += 1
        k = off = 0
        while v + k < len(c) and (k, off) in lanes:
            off += c[v + k]
            k += 1
        for x in c[v:v + k]:
            freq[x] = freq.get(x, 0) + 1
        v += k
    return rounds


def window(c, L):
    rounds, v, hist = 0, 0, [4] * 8
    while v < len(c):
        rounds += 1
        lo, hi = min(hist), max(hist)
        lanes, n, d = {(0, 0)}, 1, 1
        while n < L and d <= 16:
            for o in range(d * lo, d * hi + 1):
                if n < L:
                    lanes.add((d, o))
                    n += 1
            d += 1
        k = off = 0
        while v + k < len(c) and (k, off) in lanes:
            off += c[v + k]
            k += 1
        for x in c[v:v + k]:
            hist = [x] + hist[:7]
        v += k
    return rounds


if __name__ == "__main__":
    orc.build()
    orc.set_sqr_mode(orc.SQR_MUL)
    pix = []
    for ry, rx in ((42, 63), (42, 62), (41, 63)):  # regions of the slowest units of profiles/r03_c3_pixel_tail_units.txt
        for r in range(8):
            for cc in range(8):
                c = counts(rx * 8 + cc, ry * 8 + r)
                if len(set(c)) > 1:
                    pix.append(c)
    print(f"{len(pix)} flagged pixels of three 8x8 regions of C3 (1280x720, D = 3, spp 16); e.g. draws per sample {pix[1]}")
    print(f"{'scheme':10s} {'lanes':>5s} | rounds per pixel: mean   p90   max")
    for name, f in (("chain", chain), ("tree2", tree2), ("window", window)):
        for L in (4, 8, 16, 32, 64):
            r = [f(c, L) for c in pix]
            print(f"{name:10s} {L:5d} | {np.mean(r):22.2f} {np.percentile(r, 90):5.1f} {max(r):5d}")
    orc.set_sqr_mode(orc.SQR_POW)
