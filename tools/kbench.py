#!/usr/bin/env python3
"""Kernel micro-benchmark: times the render kernel alone (hipEvents inside the library) for the
BASELINE.json configurations, several rounds in ONE process (cdna_hip_programming.md rule 24).

    python tools/kbench.py [c2 c3 c5 c2onoff c3n10 ...] [--rounds 20]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401  (device memory for the output buffer)

from pytracer_amd import abi, flatten, scenes  # noqa: E402
from pytracer_amd.device import DeviceScene  # noqa: E402

CONFIGS = {
    # name: (n_spheres, plane, wide, W, H, params kwargs)
    "c2": (32, True, False, 1280, 720, dict(renderer=abi.RENDERER_FLAT)),
    "c2onoff": (32, True, False, 1280, 720, dict(renderer=abi.RENDERER_ONOFF)),
    "c2s2": (32, True, False, 1280, 720, dict(renderer=abi.RENDERER_FLAT, samples_per_side=2)),
    "c3": (32, False, False, 1280, 720, dict(renderer=abi.RENDERER_PATHTRACER, samples_per_side=4, num_of_rays=1,
                                             max_depth=3, rr_limit=3, path_state=45, path_seq=54)),
    "c3q": (32, False, False, 640, 360, dict(renderer=abi.RENDERER_PATHTRACER, samples_per_side=4, num_of_rays=1,
                                             max_depth=3, rr_limit=3, path_state=45, path_seq=54)),
    "c3s1": (32, False, False, 1280, 720, dict(renderer=abi.RENDERER_PATHTRACER, samples_per_side=1, num_of_rays=1,
                                              max_depth=3, rr_limit=3, path_state=45, path_seq=54)),
    "c3s2": (32, False, False, 1280, 720, dict(renderer=abi.RENDERER_PATHTRACER, samples_per_side=2, num_of_rays=1,
                                              max_depth=3, rr_limit=3, path_state=45, path_seq=54)),
    "c3sample": (32, False, False, 1280, 720, dict(renderer=abi.RENDERER_PATHTRACER, samples_per_side=4, num_of_rays=1,
                                                  max_depth=3, rr_limit=3, path_state=45, path_seq=54, pcg_mode=abi.PCG_SAMPLE)),
    "t8s4": (32, False, False, 8, 8, dict(renderer=abi.RENDERER_PATHTRACER, samples_per_side=4, num_of_rays=1,
                                          max_depth=3, rr_limit=3, path_state=45, path_seq=54)),
    "t8s1": (32, False, False, 8, 8, dict(renderer=abi.RENDERER_PATHTRACER, samples_per_side=1, num_of_rays=1,
                                          max_depth=3, rr_limit=3, path_state=45, path_seq=54)),
    "t8s8": (32, False, False, 8, 8, dict(renderer=abi.RENDERER_PATHTRACER, samples_per_side=8, num_of_rays=1,
                                          max_depth=3, rr_limit=3, path_state=45, path_seq=54)),
    "f8s4": (32, False, False, 8, 8, dict(renderer=abi.RENDERER_FLAT, samples_per_side=4)),
    "f8s8": (32, False, False, 8, 8, dict(renderer=abi.RENDERER_FLAT, samples_per_side=8)),
    "c3n10": (32, False, False, 1280, 720, dict(renderer=abi.RENDERER_PATHTRACER, samples_per_side=1, num_of_rays=10,
                                                max_depth=3, rr_limit=3, path_state=45, path_seq=54)),
    "c2n10": (32, True, False, 1280, 720, dict(renderer=abi.RENDERER_PATHTRACER, samples_per_side=1, num_of_rays=10,
                                               max_depth=3, rr_limit=3, path_state=45, path_seq=54)),
    "c4crop": (256, False, True, 960, 540, dict(renderer=abi.RENDERER_PATHTRACER, samples_per_side=4, num_of_rays=1,
                                                max_depth=5, rr_limit=3, path_state=45, path_seq=54)),
    "c4": (256, False, True, 3840, 2160, dict(renderer=abi.RENDERER_PATHTRACER, samples_per_side=8, num_of_rays=1,
                                              max_depth=5, rr_limit=3, path_state=45, path_seq=54)),
    "c4rank": (256, False, True, 3840, 2160, dict(renderer=abi.RENDERER_PATHTRACER, samples_per_side=8, num_of_rays=1,
                                                  max_depth=5, rr_limit=3, path_state=45, path_seq=54, n_ranks=8, rank=3, row_block=8)),
    "c5": (10000, False, True, 1280, 720, dict(renderer=abi.RENDERER_FLAT)),
    "c5pt": (10000, False, True, 1280, 720, dict(renderer=abi.RENDERER_PATHTRACER, samples_per_side=2, num_of_rays=1,
                                                 max_depth=3, rr_limit=3, path_state=45, path_seq=54)),
    "c2ortho": (32, True, False, 1280, 720, dict(renderer=abi.RENDERER_FLAT, ortho=True)),
    "c3ortho": (32, False, False, 1280, 720, dict(renderer=abi.RENDERER_PATHTRACER, samples_per_side=4, num_of_rays=1,
                                                  max_depth=3, rr_limit=3, path_state=45, path_seq=54, ortho=True)),
    "pl": (32, True, False, 1280, 720, dict(renderer=abi.RENDERER_POINTLIGHT, lights=2)),
    "pl5": (10000, False, True, 1280, 720, dict(renderer=abi.RENDERER_POINTLIGHT, lights=1)),
    "c5small": (10000, False, True, 320, 180, dict(renderer=abi.RENDERER_FLAT)),
    # the reference's demo scene (examples/demo.txt, clock = 150) at 1280x960 with the CLI's path-tracer defaults
    "demo10": ("demo", False, False, 1280, 960, dict(renderer=abi.RENDERER_PATHTRACER, samples_per_side=1, num_of_rays=10,
                                                     max_depth=3, rr_limit=3, path_state=45, path_seq=54)),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="*", default=["c2", "c3", "c5"])
    ap.add_argument("--rounds", type=int, default=20)
    ap.add_argument("--f64", action="store_true")
    args = ap.parse_args()
    for name in args.names:
        sample = name.endswith(":sample")  # e.g. c4:sample = the same configuration under PT_PCG_SAMPLE
        tree_only = name.endswith(":tree")  # num_of_rays > 1: the tree kernel renders the frame alone (qchoice 0)
        from pytracer_amd import device as _dev
        _dev.set_tuning("qchoice", 0 if tree_only else int(os.environ.get("PTRACE_QCHOICE", "1")))
        ns, plane, wide, W, H, kw = CONFIGS[name.split(":")[0]]
        kw = dict(kw)
        if sample:
            kw["pcg_mode"] = abi.PCG_SAMPLE
        if os.environ.get("KB_DEPTH"):  # (experiments: the same scene at another max_depth)
            kw["max_depth"] = int(os.environ["KB_DEPTH"])
        demo_cam = None
        if ns == "demo":
            world, demo_cam = scenes.demo_world(clock=150.0)
        else:
            world = scenes.synthetic_world(ns, with_plane=plane, wide=wide)
        for l in range(kw.pop("lights", 0)):
            from pytracer_amd import hostmodel as hm
            world.add_light(hm.PointLight(hm.Vec(-3.0 + 4.0 * l, 6.0 - 9.0 * l, 8.0), hm.Color(1.0, 0.9, 0.8), 0.0))
        flat = flatten.flatten_world(world)
        if kw.pop("ortho", False):
            from pytracer_amd import hostmodel as hm
            cam = flatten.flatten_camera(hm.OrthogonalCamera(W / H, hm.translation(hm.Vec(-1.0, 0.0, 1.5)) *
                                                             hm.scaling(hm.Vec(1.0, 3.0, 1.7))))
        elif demo_cam is not None:
            cam = flatten.flatten_camera(demo_cam)
        else:
            cam = flatten.flatten_camera(scenes.synthetic_camera(W, H))
        par = abi.make_params(W, H, out_format=abi.OUT_F64 if args.f64 else abi.OUT_F32, **kw)
        ds = DeviceScene(flat)
        out = torch.empty((H, W, 3), dtype=torch.float64 if args.f64 else torch.float32, device="cuda")
        nbytes = out.numel() * out.element_size()
        ms = []
        rays = 0
        for r in range(args.rounds + 2):
            ds.render_into(cam, par, out.data_ptr(), nbytes, None)
            st = ds.stats()
            rays = st.n_rays
            if r >= 2:
                ms.append(st.kernel_ms)
        ms = np.array(ms)
        n_sph = int((flat.kind == 0).sum())
        n_pl = int((flat.kind == 1).sum())
        flop = rays * (n_sph * 54 + n_pl * 36)
        t = float(np.median(ms)) * 1e-3
        print(f"{name:14s} {W}x{H} shapes={flat.n_shapes:5d} rays={rays:9d}  kernel ms: min {ms.min():.4f} "
              f"med {np.median(ms):.4f} max {ms.max():.4f} | {rays / t / 1e6:9.1f} Mray/s "
              f"{rays * flat.n_shapes / t:.3e} tests/s {flop / t / 1e12:6.2f} TFLOP/s(alg) grid={st.grid}", flush=True)
        ds.close()


if __name__ == "__main__":
    main()
