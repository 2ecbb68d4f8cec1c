#!/usr/bin/env python3
"""Speculation statistics of the path tracer's second pass (a -DPT_DEBUG_TIME build of the library):
units, pixel-rounds, samples traced per round, samples validated per round, ppu chosen by pt_unit_sort.

    PTRACE_LIB=build_variants/libptrace_dbg.so python tools/dbgspec.py c3 c4rank c4rank:sample ...
"""
import ctypes as C
import os
import sys

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
    d = (C.c_ulonglong * 8)()
    ds.render_into(cam, par, out.data_ptr(), out.numel() * 4, None)
    _lib.lib().pt_debug_read_dbg(d, 1)
    ds.render_into(cam, par, out.data_ptr(), out.numel() * 4, None)
    st = ds.stats()
    _lib.lib().pt_debug_read_dbg(d, 1)
    q = (C.c_ulonglong * 16)()
    _lib.lib().pt_debug_read_queue(ds._h, q)
    rounds, traced, kept, units = d[4], d[5], d[6], d[7]
    nsamp = max(1, par.samples_per_side) ** 2
    print(f"{name:14s} kernel {st.kernel_ms:.3f} ms  units {units} (ppu {q[10]}, {q[9]} listed)  flagged pixels {kept // nsamp}  "
          f"pixel-rounds {rounds} = {rounds / max(1, kept // nsamp):.2f}/pixel (of {nsamp} samples)  "
          f"traced {traced} kept {kept} = {kept / max(1, traced):.2f}  kept/round {kept / max(1, rounds):.2f}", flush=True)
    t = [q[i] for i in range(1, 9)]
    names = ["0 unit/round handout", "1 start_sample", "2 tile query (P)", "3 -", "4 lane query (S)", "5 shade+unwind+scatter", "6 -", "7 loop top"]
    tot = sum(t) or 1
    print("   cycles per wave: " + "  ".join(f"{n}: {v / (st.grid * 4):.0f} ({100 * v / tot:.0f}%)" for n, v in zip(names, t) if v), flush=True)
    ds.close()
