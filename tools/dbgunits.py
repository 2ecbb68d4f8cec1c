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
    first, region = (a[:, 3] >> 26) & 0x3f, a[:, 3] >> 32
    print(f"{name}: kernel {st.kernel_ms:.3f} ms, {n} units (ppu {q[10]}), unit duration kcycles: mean {dur.mean():.0f} p50 {np.median(dur):.0f} "
          f"p90 {np.percentile(dur, 90):.0f} max {dur.max():.0f}; last end {end.max():.0f} kcycles; late starters (start > 10 kcyc): {(start > 10).sum()}")
    order = np.argsort(-dur)[:8]
    for i in order:
        print(f"   unit {i} (PTRACE_TRACE_UNIT={-2 - (int(region[i]) * 64 + int(first[i]))}): count {count[i]} L {L[i]} rounds {rounds[i]} iterations {iters[i]} start {start[i]:.0f} dur {dur[i]:.0f} kcyc -> {dur[i] / max(1, rounds[i]):.1f} per round, {dur[i] / max(1, iters[i]):.1f} per iteration")
    for lo, hi in ((0, 25), (25, 50), (50, 75), (75, 100)):
        sel = (dur >= np.percentile(dur, lo)) & (dur <= np.percentile(dur, hi))
        print(f"   duration quartile {lo}-{hi}: rounds {rounds[sel].mean():.1f} iterations {iters[sel].mean():.1f} count {count[sel].mean():.1f} per-iteration {(dur[sel] / np.maximum(1, iters[sel])).mean():.1f} kcyc"
              f" | kcyc in: scattered queries {a[sel, 4].mean() / 1e3:.0f}, shade {a[sel, 5].mean() / 1e3:.0f}, start+primary {a[sel, 6].mean() / 1e3:.0f}, commit+fetch {a[sel, 7].mean() / 1e3:.0f}")
    if os.environ.get("DBG_LAT"):  # latency histograms of single vector-memory operations (cycles, log2 buckets)
        hb = (C.c_ulonglong * 160)()
        _lib.lib().pt_debug_read_lat_hist(hb, 1)
        h = np.array(list(hb), dtype=np.int64).reshape(5, 32)
        for k, what in enumerate(("load of a unit descriptor", "  outstanding before it", "atomic on a shard head", "sparse path: one ball per lane")):
            tot = max(1, h[k].sum())
            print(f"   {what:32s} n={h[k].sum():6d} " + " ".join(f"2^{b}:{h[k, b]}" for b in range(32) if h[k, b]))
        print(f"   scattered-ray queries by live rays (bins of 2, the last: 62-64): {h[4].tolist()}; <= 16: {h[4, :9].sum()}, 17-32: {h[4, 9:17].sum()}, > 32: {h[4, 17:].sum()}")
    if os.environ.get("DBG_LAT"):  # the slow operations one by one: do they cluster in time (per XCD, per CU)?
        ne = 4096
        eb = (C.c_ulonglong * (ne * 3 + 1))()
        _lib.lib().pt_debug_read_lat_events(eb, 1)
        nev = min(ne, int(eb[ne * 3]))
        ev = np.array(list(eb)[: nev * 3], dtype=np.int64).reshape(nev, 3)
        if nev:
            if os.path.isdir("gpurun_out"):
                np.save("gpurun_out/lat_events.npy", ev)
            t = (ev[:, 0] - ev[:, 0].min()) / 100.0  # us (100 MHz)
            which, xcc, hwid = ev[:, 2] & 0xff, (ev[:, 2] >> 8) & 0xff, ev[:, 2] >> 16
            cu, se = (hwid >> 8) & 0xf, (hwid >> 13) & 0x7
            print(f"   {nev} operations of >= 8192 cycles; end time (us) histogram in 20-us bins: {np.histogram(t, bins=np.arange(0, t.max() + 20, 20))[0].tolist()}")
            for x in range(8):
                sel = xcc == x
                print(f"     xcc {x}: {sel.sum()} slow ops, by 20-us bin {np.histogram(t[sel], bins=np.arange(0, t.max() + 20, 20))[0].tolist()}, distinct (se,cu) {len(set(zip(se[sel].tolist(), cu[sel].tolist())))}")
            o = np.argsort(t)
            print("     first 40 by time: " + " ".join(f"{t[i]:.1f}us/x{xcc[i]}s{se[i]}c{cu[i]}/k{which[i]}/{ev[i, 1] // 1000}k" for i in o[:40]))
    if os.environ.get("DBG_TRACE"):  # the per-step trace of unit PTRACE_TRACE_UNIT: (cycles since the previous stamp, lanes holding a ray, section)
        nt = 8192
        tb = (C.c_ulonglong * nt)()
        _lib.lib().pt_debug_read_trace(tb, nt)
        rows = [(tb[i] >> 16, (tb[i] >> 8) & 0xff, tb[i] & 0xff) for i in range(nt) if tb[i]]
        print(f"   trace of unit {os.environ.get('PTRACE_TRACE_UNIT')}: {len(rows)} stamps; section:cycles(lanes with a ray)")
        print("   " + " ".join(f"{k}:{t}({npth})" for t, npth, k in rows[:400]))
    ds.close()
