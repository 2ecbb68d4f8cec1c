#!/usr/bin/env python3
"""Draw counts of the validated samples of ONE unit's pixels (a -DPT_DEBUG_TIME build; PTRACE_TRACE_UNIT=<unit>):
what the speculation on the pixel's shared generator has to guess.

    PTRACE_LIB=pytracer_amd/libptrace_dbg.so PTRACE_TRACE_UNIT=1306 python tools/dbgdraws.py c3
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

name = sys.argv[1]
ns, plane, wide, W, H, kw = CONFIGS[name]
flat = flatten.flatten_world(scenes.synthetic_world(ns, with_plane=plane, wide=wide))
cam = flatten.flatten_camera(scenes.synthetic_camera(W, H))
par = abi.make_params(W, H, out_format=abi.OUT_F32, **kw)
ds = DeviceScene(flat)
out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
ds.render_into(cam, par, out.data_ptr(), out.numel() * 4, None)
n = 8192 + 64 * 80
buf = (C.c_ulonglong * n)()
_lib.lib().pt_debug_read_trace(buf, n)
a = np.frombuffer(buf, dtype=np.uint64)[8192:].reshape(64, 80)
key = int(os.environ.get("PTRACE_TRACE_UNIT", "0"))  # a unit number, or -2 - (region * 64 + first flagged pixel) as tools/dbgunits.py prints it
q = (C.c_ulonglong * 16)()
_lib.lib().pt_debug_read_queue(ds._h, q)
nu = min(int(q[9]), 16384)
ub = (C.c_ulonglong * (8 * nu))()
_lib.lib().pt_debug_read_unitlog(ub, nu)
ula = np.frombuffer(ub, dtype=np.uint64).reshape(-1, 8).astype(np.int64)
for unit in range(nu):
    ul = ula[unit]
    if unit == key or (key <= -2 and (ul[3] >> 32) * 64 + ((ul[3] >> 26) & 0x3f) == -2 - key):
        print(f"unit {unit}: {ul[3] & 0xff} pixels, L {(ul[3] >> 8) & 0xff}, {ul[2] & 0xffffffff} rounds, {ul[2] >> 32} iterations, {(ul[1] - ul[0]) / 1e3:.0f} kcycles")
for p in range(64):
    row = a[p]
    seq = [int(x >> 16) & 0xffff for x in row if (int(x) & 0xff) == 0xEE]
    rnd = [int(x >> 32) for x in row if (int(x) & 0xff) == 0xEE]
    if seq:
        runs = sum(1 for i in range(1, len(seq)) if seq[i] != seq[i - 1])
        vals, cnt = np.unique(seq, return_counts=True)
        top2 = np.sort(cnt)[::-1][:2].sum() / len(seq)
        print(f"pixel {p:2d}: {len(seq)} samples, {runs} changes, top-2 values cover {top2:.2f}: {' '.join(map(str, seq))}")
        print(f"          validated in round: {' '.join(map(str, rnd))}")
