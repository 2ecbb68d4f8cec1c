import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pytracer_amd import abi, flatten, scenes
from pytracer_amd.device import DeviceScene
import ctypes as C
from pytracer_amd import _lib
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1280, 720)
nsph = int(sys.argv[3]) if len(sys.argv) > 3 else 32
world = scenes.synthetic_world(nsph, wide=nsph > 64, with_plane=bool(int(os.environ.get('DBG_PLANE', 0))))
for l in range(int(os.environ.get('DBG_LIGHTS', 0))):  # (as tools/kbench.py's "pl")
    from pytracer_amd import hostmodel as hm
    world.add_light(hm.PointLight(hm.Vec(-3.0 + 4.0 * l, 6.0 - 9.0 * l, 8.0), hm.Color(1.0, 0.9, 0.8), 0.0))
flat = flatten.flatten_world(world)
cam = flatten.flatten_camera(scenes.synthetic_camera(W, H))
REND = {'path': abi.RENDERER_PATHTRACER, 'flat': abi.RENDERER_FLAT, 'onoff': abi.RENDERER_ONOFF, 'pointlight': abi.RENDERER_POINTLIGHT}[os.environ.get('DBG_RENDERER', 'path')]
par = abi.make_params(W, H, REND, samples_per_side=int(os.environ.get('DBG_S', 4)), num_of_rays=int(os.environ.get('DBG_N', 1)), max_depth=int(os.environ.get('DBG_D', 3)), rr_limit=3, path_state=45, path_seq=54,
                      pcg_mode=int(os.environ.get('DBG_MODE', 1)), n_ranks=int(os.environ.get('DBG_RANKS', 1)), rank=int(os.environ.get('DBG_RANK', 0)), row_block=8)
ds = DeviceScene(flat)
for _ in range(1):
    out = ds.render(cam, par)
q = (C.c_ulonglong * 16)()
_lib.lib().pt_debug_read_queue(ds._h, q)
t = np.array([q[i] for i in range(1, 9)], dtype=np.float64)
names = ["0 unit setup", "1 start_sample", "2 tile query (P)", "3 round commit", "4 full query (S)", "5 shade+unwind+scatter", "6 unit fetch (atomic)", "7 loop top"]
if REND != abi.RENDERER_PATHTRACER:
    names = ["0 tile setup", "1 tile cone", "2 cull", "3 dome check + ray generation", "4 tile query", "5 shade + store", "6 after the pixel loop", "7 strip / loop top"]
print("kernel ms", ds.stats().kernel_ms, "rays", ds.stats().n_rays)
for n, v in zip(names, t):
    print(f"{n:22s} {v:12.0f} cycles")
print("sum", t.sum(), "= per wave", t.sum() / (ds.stats().grid * 4))

if os.environ.get("DBG_TRACE"):
    n = 8192
    buf = (C.c_ulonglong * n)()
    _lib.lib().pt_debug_read_trace(buf, n)
    rows = [(buf[i] >> 16, (buf[i] >> 8) & 0xff, buf[i] & 0xff) for i in range(n) if buf[i]]
    print("trace entries", len(rows))
    import collections
    agg = collections.defaultdict(lambda: [0, 0])
    for t, npath, k in rows:
        agg[k][0] += 1
        agg[k][1] += t
    for k in sorted(agg):
        print(f"stamp {k}: n={agg[k][0]} total={agg[k][1]} ticks avg={agg[k][1]/agg[k][0]:.1f}")
    print("first 160:", " ".join(f"{k}:{t}({npth})" for t, npth, k in rows[:160]))

if os.environ.get("DBG_LANES"):
    d = (C.c_ulonglong * 8)()
    _lib.lib().pt_debug_read_dbg(d, 1)
    calls = max(1, d[3])
    print(f"world_query_lanes: calls {d[3]}, prefilter {d[0]/calls:.0f} cycles/call, walk {d[1]/calls:.0f} cycles/call, "
          f"iterations {d[2]/calls:.2f}/call -> {d[1]/max(1,d[2]):.0f} cycles/iteration")
