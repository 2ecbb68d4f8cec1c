import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pytracer_amd import abi, flatten, scenes
from pytracer_amd.device import DeviceScene
import ctypes as C
from pytracer_amd import _lib
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1280, 720)
nsph = int(sys.argv[3]) if len(sys.argv) > 3 else 32
flat = flatten.flatten_world(scenes.synthetic_world(nsph, wide=nsph > 64))
cam = flatten.flatten_camera(scenes.synthetic_camera(W, H))
par = abi.make_params(W, H, abi.RENDERER_PATHTRACER, samples_per_side=4, num_of_rays=1, max_depth=3, rr_limit=3, path_state=45, path_seq=54)
ds = DeviceScene(flat)
for _ in range(2):
    out = ds.render(cam, par)
q = (C.c_ulonglong * 16)()
_lib.lib().pt_debug_read_queue(ds._h, q)
t = np.array([q[i] for i in range(1, 9)], dtype=np.float64)
names = ["handout", "start_sample", "tile query", "shade+finish(P)", "unwind/scatter(S)", "full query+shade(S)", "-", "loop top"]
print("kernel ms", ds.stats().kernel_ms, "rays", ds.stats().n_rays)
for n, v in zip(names, t):
    print(f"{n:22s} {v:12.0f} cycles")
print("sum", t.sum(), "= per wave", t.sum() / (ds.stats().grid * 4))
