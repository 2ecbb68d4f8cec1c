#!/usr/bin/env python3
"""num_of_rays > 1: the tree kernel (one pixel per wave) against the one-queue kernel (a lane per flagged pixel), and what
the device chooses by itself (PT_Q_CHOICE; csrc/ptrace.hip: q_min).  Kernel ms of a frame, three frames each.

    python tools/tree_vs_queue.py            # the table: every case under PTRACE_QCHOICE = 0 (tree), 2 (queue), 1 (device's choice)
    python tools/tree_vs_queue.py --one ...  # (internal: one case in this process; the switches are read once per process)
"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CASES = [  # scene, W, H, N, D, S
    ("demo", 1280, 960, 10, 3, 1), ("demo", 640, 480, 10, 3, 1), ("demo", 1920, 1440, 10, 3, 1), ("demo", 320, 240, 10, 3, 1),
    ("plane", 1280, 720, 10, 3, 1), ("plane", 640, 360, 10, 3, 1), ("plane", 1920, 1080, 10, 3, 1),
    ("c3", 1280, 720, 10, 3, 1), ("c3", 1920, 1080, 10, 3, 1),
    ("demo", 1280, 960, 4, 3, 1), ("plane", 1280, 720, 4, 3, 1), ("demo", 1280, 960, 10, 2, 1), ("plane", 1280, 720, 10, 2, 1),
    ("demo", 1280, 960, 3, 3, 2), ("plane", 1280, 720, 20, 2, 1), ("demo", 1280, 960, 2, 3, 1), ("plane", 1280, 720, 2, 3, 1),
    # stacks deeper than the LDS holds (D > 3): the one-queue kernel keeps them in HBM
    ("plane", 1280, 720, 2, 5, 1), ("plane", 1280, 720, 3, 5, 1), ("plane", 1280, 720, 2, 8, 1), ("plane", 640, 360, 3, 5, 1),
    ("plane", 1280, 720, 4, 4, 1), ("demo", 1280, 960, 3, 5, 1), ("c3", 1280, 720, 3, 5, 1), ("plane", 320, 180, 3, 5, 1),
]


def one(which, W, H, N, D, S):
    import ctypes as C

    from pytracer_amd import _lib, abi, flatten, scenes
    from pytracer_amd.device import DeviceScene

    if which == "demo":
        world, camera = scenes.demo_world(clock=150.0)
        flat, cam = flatten.flatten_world(world), flatten.flatten_camera(camera)
    else:
        flat = flatten.flatten_world(scenes.synthetic_world(32, with_plane=(which == "plane")))
        cam = flatten.flatten_camera(scenes.synthetic_camera(W, H))
    par = abi.make_params(W, H, abi.RENDERER_PATHTRACER, samples_per_side=S, num_of_rays=N, max_depth=D, path_state=45, path_seq=54,
                          out_format=abi.OUT_F32)
    with DeviceScene(flat) as ds:
        ms = []
        for _ in range(4):
            out = ds.render(cam, par)
            ms.append(ds.stats().kernel_ms)
        q = (C.c_ulonglong * 16)()
        _lib.lib().pt_debug_read_queue(ds._h, q)
        st = ds.stats()
        import zlib
        print(f"{min(ms[1:]):.3f} {int(q[11])} {st.kernel} {int(st.n_rays)} {zlib.crc32(out.tobytes())}")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        which, W, H, N, D, S = sys.argv[2], *map(int, sys.argv[3:8])
        one(which, W, H, N, D, S)
        return
    print(f"{'scene':6s} {'frame':>10s} {'N':>3s} {'D':>2s} {'S':>2s} {'flagged':>9s} | {'tree ms':>8s} {'queue ms':>9s} | {'chosen':>7s} {'ms':>8s}  frames equal")
    for case in CASES:
        res = {}
        for mode in ("0", "2", "1"):
            env = dict(os.environ, PTRACE_QCHOICE=mode)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"] + [str(c) for c in case], env=env,
                               capture_output=True, text=True, timeout=600)
            if r.returncode != 0:
                res[mode] = None
                print("  failed:", case, mode, r.stderr[-400:])
                continue
            ms, F, kernel, rays, crc = r.stdout.split()[-5:]
            res[mode] = (float(ms), int(F), int(kernel), int(rays), int(crc))
        if all(res.values()):
            t, q, c = res["0"], res["2"], res["1"]
            same = t[4] == q[4] == c[4] and t[3] == q[3] == c[3]
            print(f"{case[0]:6s} {case[1]:5d}x{case[2]:<4d} {case[3]:3d} {case[4]:2d} {case[5]:2d} {t[1]:9d} | {t[0]:8.3f} {q[0]:9.3f} | "
                  f"{'queue' if c[2] == 4 else 'tree':>7s} {c[0]:8.3f}  {same}", flush=True)


if __name__ == "__main__":
    main()
