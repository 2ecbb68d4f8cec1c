import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from pytracer_amd import abi, flatten, scenes
from pytracer_amd.device import DeviceScene
flat = flatten.flatten_world(scenes.synthetic_world(32, with_plane=True))
for W, H in ((160, 90), (320, 180), (640, 360), (1280, 720)):
    cam = flatten.flatten_camera(scenes.synthetic_camera(W, H))
    par = abi.make_params(W, H, abi.RENDERER_PATHTRACER, samples_per_side=1, num_of_rays=10, max_depth=3, path_state=45, path_seq=54)
    with DeviceScene(flat) as ds:
        ms = []
        for r in range(5):
            ds.render(cam, par)
            ms.append(ds.stats().kernel_ms)
        print(W, H, "plane scene", ["%.3f" % m for m in ms], ds.stats().n_rays, ds.stats().grid)
flat = flatten.flatten_world(scenes.synthetic_world(32, with_plane=False))
cam = flatten.flatten_camera(scenes.synthetic_camera(160, 90))
with DeviceScene(flat) as ds:
    ms = []
    for r in range(5):
        ds.render(cam, par if False else abi.make_params(160, 90, abi.RENDERER_PATHTRACER, samples_per_side=1, num_of_rays=10, max_depth=3, path_state=45, path_seq=54))
        ms.append(ds.stats().kernel_ms)
    print("no plane 160x90", ["%.3f" % m for m in ms], ds.stats().n_rays)
