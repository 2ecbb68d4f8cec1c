#!/usr/bin/env python3
"""Per-unit log of the path tracer's second pass (a -DPT_DEBUG_TIME build): when each work unit started and
ended, its rounds and loop iterations, its size.

    PTRACE_LIB=build_variants/libptrace_dbg.so python tools/dbgunits.py c3 c4rank:sample ...
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

from pytracer_amd import _lib, abi, flatten, scenes  # noqa: E402
from pytracer_amd.device import DeviceScene  # noqa: E402
from tools.kbench import CONFIGS  # noqa: E402

for name in sys.argv[1:]:
    ns, plane, wide, W, H, kw = CONFIGS[name.split(":")[0]]
    kw = dict(kw)
    if name.endswith(":sample"):
        kw["pcg_mode"] = abi.PCG_SAMPLE
    flat = flatten.flatten_world(scenes.synthetic_world(ns, with_plane=plane, wide=wide))
    cam = flatten.flatten_camera(scenes.synthetic_camera(W, H))
    par = abi.make_params(W, H, out_format=abi.OUT_F32, **kw)
    ds = DeviceScene(flat)
    out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    for _ in range(2):
        ds.render_into(cam, par, out.data_ptr(), out.numel() * 4, None)
    st = ds.stats()
    q = (C.c_ulonglong * 16)()
    _lib.lib().pt_debug_read_queue(ds._h, q)
    n = min(int(q[9]), 16384)
    buf = (C.c_ulonglong * (8 * n))()
    _lib.lib().pt_debug_read_unitlog(buf, n)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 8).astype(np.int64)
    t0 = a[:, 0].min()
    start, end = (a[:, 0] - t0) / 1e3, (a[:, 1] - t0) / 1e3
    dur = end - start
    rounds, iters = a[:, 2] & 0xffffffff, a[:, 2] >> 32
    count, L = a[:, 3] & 0xff, (a[:, 3] >> 8) & 0xff
    print(f"{name}: kernel {st.kernel_ms:.3f} ms, {n} units (ppu {q[10]}), unit duration kcycles: mean {dur.mean():.0f} p50 {np.median(dur):.0f} "
          f"p90 {np.percentile(dur, 90):.0f} max {dur.max():.0f}; last end {end.max():.0f} kcycles; late starters (start > 10 kcyc): {(start > 10).sum()}")
    order = np.argsort(-dur)[:8]
    for i in order:
        print(f"   unit {i}: count {count[i]} L {L[i]} rounds {rounds[i]} iterations {iters[i]} start {start[i]:.0f} dur {dur[i]:.0f} kcyc -> {dur[i] / max(1, rounds[i]):.1f} per round, {dur[i] / max(1, iters[i]):.1f} per iteration")
    for lo, hi in ((0, 25), (25, 50), (50, 75), (75, 100)):
        sel = (dur >= np.percentile(dur, lo)) & (dur <= np.percentile(dur, hi))
        print(f"   duration quartile {lo}-{hi}: rounds {rounds[sel].mean():.1f} iterations {iters[sel].mean():.1f} count {count[sel].mean():.1f} per-iteration {(dur[sel] / np.maximum(1, iters[sel])).mean():.1f} kcyc"
              f" | kcyc in: scattered queries {a[sel, 4].mean() / 1e3:.0f}, shade {a[sel, 5].mean() / 1e3:.0f}, start+primary {a[sel, 6].mean() / 1e3:.0f}, commit+fetch {a[sel, 7].mean() / 1e3:.0f}")
    ds.close()
