import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import util
from pytracer_amd import abi, device
from pytracer_amd.device import DeviceScene
scene, cam, par, pixels = util.load_frame("g5_c2_flat_160x90")
rng = np.random.default_rng(2)
a, b = rng.normal(size=100000), rng.normal(size=100000)
device.probe(7, a, b)
ds = DeviceScene(scene)
out = ds.render(cam, par)
np.set_printoptions(linewidth=250)
print("popcount per tile (row 44):", out[44, ::8, 0])
print("mask (row 44):", [hex(int(v)) for v in out[44, ::8, 1]])
print("cos (row 44):", out[44, ::8, 2])
print("popcount per tile (col 84):", out[::8, 84, 0])
