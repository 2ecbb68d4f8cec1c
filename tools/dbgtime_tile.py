import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pytracer_amd import abi, flatten, scenes, _lib
from pytracer_amd.device import DeviceScene
W, H = 1280, 720
S = int(sys.argv[1]) if len(sys.argv) > 1 else 0
flat = flatten.flatten_world(scenes.synthetic_world(32, with_plane=True))
cam = flatten.flatten_camera(scenes.synthetic_camera(W, H))
par = abi.make_params(W, H, abi.RENDERER_FLAT, samples_per_side=S, out_format=abi.OUT_F32)
ds = DeviceScene(flat)
for _ in range(3):
    out = ds.render(cam, par)
q = (C.c_ulonglong * 16)()
_lib.lib().pt_debug_read_queue(ds._h, q)
t = np.array([q[i] for i in range(1, 9)], dtype=np.float64)
names = ["coords", "cone", "cull+LDS", "sample setup+primary ray", "tile query", "shade+accum", "store", "loop top"]
nw = ((ds.stats().grid + 63) // 64) * 4  # sampled workgroups only
if ds.stats().kernel == abi.KERNEL_TILE4:  # 16x16 tiles, one per wave; every 16th workgroup reports
    names = ["prologue", "cone", "cull", "dome tile", "four rays", "query", "shade", "store"]
    nw = ((ds.stats().grid + 15) // 16) * 4
print("kernel ms", ds.stats().kernel_ms, "sampled waves", nw, "tiles/wave", 14400 / (ds.stats().grid * 4))
for n, v in zip(names, t):
    print(f"{n:26s} {v / nw:10.0f} cycles/wave  {100 * v / t.sum():5.1f} %")
print("sum per wave", t.sum() / nw)
if ds.stats().kernel == abi.KERNEL_TILE4:
    print(f"longest wave {q[12]} cycles; waves above 16 / 24 / 32 kcycles: {q[13]} / {q[14]} / {q[15]} of the {nw} sampled")
