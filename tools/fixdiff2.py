import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import util
from pytracer_amd import abi, device
from pytracer_amd.device import DeviceScene
from oracle import oracle
name = "g5_c2_flat_160x90"
scene, cam, par, pixels = util.load_frame(name)
if os.environ.get("PROBE"):
    rng = np.random.default_rng(2)
    a, b = rng.normal(size=100000), rng.normal(size=100000)
    got = device.probe(7, a, b)
ds = DeviceScene(scene)
out = ds.render(cam, par)
par_o = abi.copy_params(par, pcg_mode=abi.PCG_PIXEL)
for nt in (0, 1, 4):
    ora, n = oracle.render(scene, cam, par_o, n_threads=nt, sqr_mode=oracle.SQR_MUL)
    bad = np.argwhere((ora != pixels).any(axis=-1))
    print("threads", nt, "oracle vs golden differing:", len(bad), "rays", n)
    for y, x in bad[:5]:
        print("  ", y, x, ora[y, x], pixels[y, x])
    if len(bad): print("  rows", sorted(set(bad[:, 0]))[:40])
bad = np.argwhere((out != pixels).any(axis=-1))
print("device vs golden:", len(bad))
for y, x in bad[:12]:
    print("  ", y, x, out[y, x], pixels[y, x])
if len(bad):
    print("  rows", sorted(set(bad[:, 0]))); print("  cols", sorted(set(bad[:, 1])))
out2 = ds.render(cam, par)
print("second render vs golden:", int((out2 != pixels).any(axis=-1).sum()))
