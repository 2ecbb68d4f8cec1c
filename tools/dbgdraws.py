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
for p in range(64):
    row = a[p]
    seq = [int(x >> 16) for x in row if (int(x) & 0xff) == 0xEE]
    if seq:
        runs = sum(1 for i in range(1, len(seq)) if seq[i] != seq[i - 1])
        vals, cnt = np.unique(seq, return_counts=True)
        top2 = np.sort(cnt)[::-1][:2].sum() / len(seq)
        print(f"pixel {p:2d}: {len(seq)} samples, {runs} changes, top-2 values cover {top2:.2f}: {' '.join(map(str, seq))}")
