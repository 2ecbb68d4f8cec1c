import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from pytracer_amd import abi, flatten, scenes
    from pytracer_amd.device import DeviceScene
    W, H = 160, 90
    flat = flatten.flatten_world(scenes.synthetic_world(32, with_plane=True))
    cam = flatten.flatten_camera(scenes.synthetic_camera(W, H))
    par = abi.make_params(W, H, abi.RENDERER_FLAT)
    ds = DeviceScene(flat)
    np.save(sys.argv[2], ds.render(cam, par))
else:
    for c in ("0", "1"):
        env = dict(os.environ, PTRACE_CULL=c)
        subprocess.run([sys.executable, __file__, "child", f"/tmp/cull{c}.npy"], env=env, check=True)
    a, b = np.load("/tmp/cull0.npy"), np.load("/tmp/cull1.npy")
    bad = np.argwhere((a != b).any(axis=-1))
    print("differing pixels:", len(bad))
    for y, x in bad[:40]:
        print(y, x, a[y, x], b[y, x])
    if len(bad):
        print("rows", sorted(set(bad[:, 0]))[:50]); print("cols", sorted(set(bad[:, 1]))[:80])
