import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import util
from pytracer_amd import abi
from pytracer_amd.device import DeviceScene
name = sys.argv[1] if len(sys.argv) > 1 else "g5_c2_flat_160x90"
scene, cam, par, pixels = util.load_frame(name)
from pytracer_amd import device
if os.environ.get("PROBE"):
    print(device.probe(0, np.linspace(0, 10, 100000))[:3])
ds = DeviceScene(scene)
for rep in range(3):
    out = ds.render(cam, par)
    bad = np.argwhere((out != pixels).any(axis=-1))
    print("rep", rep, "differing pixels vs golden:", len(bad), "stats grid", ds.stats().grid, ds.stats().lds_bytes)
    for y, x in bad[:10]:
        print("  ", y, x, out[y, x], pixels[y, x])
    if len(bad):
        print("  rows", sorted(set(bad[:, 0]))[:40]); print("  cols", sorted(set(bad[:, 1]))[:60])
