#!/usr/bin/env python3
"""pt_tile4_kernel on C2 (a -DPT_DEBUG_TIME build): cycles of every wave, by workgroup -- how uneven are a workgroup's four
16x16 tiles, and what would sharing work inside a workgroup buy?  

    PTRACE_LIB=build_variants/libptrace_dbg.so python tools/dbgtile4_waves.py
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

from pytracer_amd import _lib, abi, flatten, scenes  # noqa: E402
from pytracer_amd.device import DeviceScene  # noqa: E402

W, H = 1280, 720
flat = flatten.flatten_world(scenes.synthetic_world(32, with_plane=True))
cam = flatten.flatten_camera(scenes.synthetic_camera(W, H))
par = abi.make_params(W, H, abi.RENDERER_FLAT, out_format=abi.OUT_F32)
ds = DeviceScene(flat)
out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
for _ in range(3):
    ds.render_into(cam, par, out.data_ptr(), out.numel() * 4, None)
st = ds.stats()
n = st.grid * 4
buf = (C.c_ulonglong * (8 * n))()
_lib.lib().pt_debug_read_unitlog(buf, n)
a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 8).astype(np.int64)
tot, per_tile, end = a[:, 0].copy(), a[:, 1].copy(), a[:, 2]
# (every 16th workgroup also reports section sums with atomics in this build: its waves are slow for THAT reason: left out)
sampled = np.repeat(np.arange(n // 4) % 16 == 0, 4)
tot[sampled] = 0
per_tile[sampled] = 0
live = tot > 0
wg = tot.reshape(-1, 4)
print(f"kernel {st.kernel_ms * 1e3:.1f} us (debug build), {n} waves, {live.sum()} with a tile")
print(f"wave cycles: mean {tot[live].mean():.0f}  p50 {np.median(tot[live]):.0f}  p90 {np.percentile(tot[live], 90):.0f}  p99 {np.percentile(tot[live], 99):.0f}  max {tot.max()}")
print(f"  of which per tile (prologue, cone, cull, dome check): mean {per_tile[live].mean():.0f}")
# the histogram of wave cycles, and per tile ROW of the frame the longest wave: where the heavy tiles are
edges = np.arange(0, int(tot.max()) + 2048, 2048)
hist, _ = np.histogram(tot[live], bins=edges)
print("wave-cycle histogram (2 048-cycle bins: waves | share of the waves' summed cycles):")
for k, h in enumerate(hist):
    if h:
        sel = live & (tot >= edges[k]) & (tot < edges[k + 1])
        print(f"  {edges[k]:6d} .. {edges[k + 1]:6d}  {h:5d}  {'#' * int(round(60 * h / hist.max()))}  {100 * tot[sel].sum() / tot[live].sum():4.1f} %")
rows = {}
for i in range(n // 4):
    for wv in range(4):
        t = int(wg[i, wv])
        if t:
            ty = i // 40 * 2 + (wv >> 1)
            rows.setdefault(ty, []).append(t)
print("per tile row (16 pixel rows each, top = 0): waves, mean, max cycles")
for ty in sorted(rows):
    r = np.array(rows[ty])
    print(f"  row {ty:2d}: {len(r):3d} waves  mean {r.mean():7.0f}  max {r.max():6d}")
wmax = wg.max(axis=1)
wmean = wg.sum(axis=1) / np.maximum(1, (wg > 0).sum(axis=1))
print(f"workgroups: max of the four waves: mean {wmax.mean():.0f}  max {wmax.max()};  mean of the four: max over workgroups {wmean.max():.0f}")
heavy = np.argsort(-wmax)[:8]
for i in heavy:
    print(f"   workgroup {i} (tiles {i % 40 * 2}..{i % 40 * 2 + 1} x {i // 40 * 2}..{i // 40 * 2 + 1}): waves {wg[i].tolist()}  -> perfectly shared inside the workgroup: {wmean[i]:.0f}")
# what the frame waits for now (its longest wave) against what it would wait for if a workgroup's waves shared their work perfectly
print(f"longest wave {tot.max()} cycles; longest workgroup mean {wmean.max():.0f} cycles ({100 * (1 - wmean.max() / tot.max()):.0f} % shorter); "
      f"with half of each tile's per-pixel work movable: {np.max(np.maximum(wmean, (wg.max(axis=1) + per_tile.reshape(-1, 4).max(axis=1)) / 2)):.0f}")
t0 = end[live].min() - tot[live].max()
print(f"last wave ends {end.max() - t0} ticks after the earliest possible start (100 MHz ticks x 24 = cycles)")
ds.close()
