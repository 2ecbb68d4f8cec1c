#!/usr/bin/env python3
"""Section cycles and round statistics of pt_path_tree_kernel (a -DPT_DEBUG_TIME build).

    PTRACE_LIB=build_variants/libptrace_dbg.so python tools/dbgtree.py [c3n10 ...]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

from pytracer_amd import _lib, abi, flatten, scenes  # noqa: E402
from pytracer_amd.device import DeviceScene  # noqa: E402
from tools.kbench import CONFIGS  # noqa: E402

for name in sys.argv[1:] or ["c3n10"]:
    ns, plane, wide, W, H, kw = CONFIGS[name]
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
    names = ["fetch + cull", "primary ray", "state jump + scatter", "scattered-ray query", "shade", "commit", "node returns"]
    rounds = max(1, q[8])
    tot = sum(q[1:8])
    fused, fused_hit, q12 = (q[12] >> 24) & 0xfffff, q[12] >> 44, q[12] & 0xffffff
    print(f"  leaf rounds carrying the parent's next child: {fused}, of which it was there when the family returned: {fused_hit}")
    print(f"{name}: kernel {st.kernel_ms:.3f} ms, rays {st.n_rays}, units {q[9]}, rounds {q[8]} (leaf {q12}), "
          f"children traced {q[14] & ((1 << 40) - 1)} (leaf rounds with a fused ray whose family did not complete: {q[14] >> 40}), rays committed {q[13]}, most rounds in one pixel {q[15]}")
    for n, v in zip(names, list(q)[1:8]):
        print(f"  {n:22s} {v / 1e6:10.2f} Mcycles  {100 * v / tot:5.1f} %   {v / rounds:8.0f} cycles / round")
    ds.close()

if os.environ.get("DBG_LANES"):  # the scattered-ray query's own split (world_query_lanes; counters of the whole process)
    d = (C.c_ulonglong * 8)()
    _lib.lib().pt_debug_read_dbg(d, 1)
    calls = max(1, d[3])
    print(f"world_query_lanes: calls {d[3]}, prefilter {d[0] / calls:.0f} cycles/call, walk {d[1] / calls:.0f} cycles/call, "
          f"iterations {d[2] / calls:.2f}/call -> {d[1] / max(1, d[2]):.0f} cycles/iteration")
