import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
from pytracer_amd import abi, flatten, scenes, _lib
from pytracer_amd.device import DeviceScene
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1280, 720)
which = sys.argv[3] if len(sys.argv) > 3 else "plane"
if which == "demo":
    world, camera = scenes.demo_world(clock=150.0)
    flat, cam = flatten.flatten_world(world), flatten.flatten_camera(camera)
else:
    flat = flatten.flatten_world(scenes.synthetic_world(32, with_plane=(which == "plane")))
    cam = flatten.flatten_camera(scenes.synthetic_camera(W, H))
par = abi.make_params(W, H, abi.RENDERER_PATHTRACER, samples_per_side=1, num_of_rays=10, max_depth=3, path_state=45, path_seq=54)
with DeviceScene(flat) as ds:
    ms = []
    for r in range(4):
        ds.render(cam, par)
        ms.append(ds.stats().kernel_ms)
    q = (C.c_ulonglong * 16)()
    _lib.lib().pt_debug_read_queue(ds._h, q)
    print(which, W, H, "kernel ms", ["%.2f" % m for m in ms], "rays", ds.stats().n_rays, "units", q[9], "ppu", q[10], "F", q[11], "kernel", ds.stats().kernel)
