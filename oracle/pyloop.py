"""The per-pixel loop in the INTERPRETER: a pure-Python restatement of ``ImageTracer.fire_all_rays`` driving a
``FlatRenderer`` / ``OnOffRenderer`` with pixel-centre rays, over the flattened scene.

*** TEST INFRASTRUCTURE — NOT PRODUCT CODE ***  Only ``tests/`` and ``bench.py``'s ``cpu_baseline`` leg import it: it is
the "interpreted baseline" of SURVEY.md §8(d)(ii) -- what the same loop costs in CPython on the GPU box, where the
reference itself cannot travel -- and, like ``pt_oracle.c``, a checker that is itself checked (against the C oracle, bit
for bit, ``tests/test_oracle_golden.py``).  Plain floats and tuples, one Python-level operation per reference operation:

  imagetracer.py:48-58, 60-110   fire_ray / fire_all_rays (S = 0: one ray through the pixel centre)
  camera.py:59-78, 103-124       OrthogonalCamera / PerspectiveCamera.fire_ray
  transformations.py:58-86       Transformation * Point / Vec (x*r0 + y*r1 + z*r2 (+ r3), left to right)
  world.py:51-69                 closest hit, strict <, first shape wins ties
  shapes.py:97-131, 163-189      Sphere / Plane.ray_intersection (squared norm as x*x: the device's arithmetic)
  materials.py:58-59, 96-100     Uniform / Checkered pigments (ImagePigment: not needed by the baseline scenes)
  render.py:52-53, 65-74         OnOffRenderer / FlatRenderer
"""
from __future__ import annotations

from math import acos, atan2, floor, pi, sqrt
from typing import Tuple

import numpy as np

from pytracer_amd import abi


def _rows(a12):  # [12, n] -> per shape a tuple of 12 floats
    return [tuple(float(a12[k, i]) for k in range(12)) for i in range(a12.shape[1])]


def render(scene: abi.FlatScene, cam: abi.Camera, params: abi.Params) -> Tuple[np.ndarray, int]:
    """-> ([H, W, 3] float64, rays): OnOff or Flat, S = 0, the whole frame, one interpreter thread."""
    if params.samples_per_side != 0 or params.renderer not in (abi.RENDERER_ONOFF, abi.RENDERER_FLAT):
        raise ValueError("the interpreted baseline renders OnOff / Flat frames with pixel-centre rays")
    W, H = int(params.width), int(params.height)
    n = scene.n_shapes
    kind = [int(k) for k in scene.kind]
    invm = _rows(scene.invm)
    pig = [(int(scene.pig_kind[i]), tuple(map(float, scene.pig_c1[:, i])), tuple(map(float, scene.pig_c2[:, i])), float(scene.pig_steps[i]))
           for i in range(n)]
    emi = [(int(scene.emi_kind[i]), tuple(map(float, scene.emi_c1[:, i])), tuple(map(float, scene.emi_c2[:, i])), float(scene.emi_steps[i]))
           for i in range(n)]
    if any(k == abi.PIGMENT_IMAGE for k, *_ in pig + emi):
        raise ValueError("ImagePigment is outside the interpreted baseline")
    cm = tuple(float(x) for x in cam.m)
    dist, aspect = float(cam.screen_distance), float(cam.aspect_ratio)
    persp = cam.kind == abi.CAMERA_PERSPECTIVE
    bg = tuple(float(x) for x in params.background)
    white = tuple(float(x) for x in params.onoff_color)
    onoff = params.renderer == abi.RENDERER_ONOFF
    out = np.zeros((H, W, 3), dtype=np.float64)
    inf = float("inf")

    def pigment(p, u, v):  # materials.py:58-59, 96-100
        k, c1, c2, steps = p
        if k == abi.PIGMENT_CHECKERED:
            return c1 if (int(floor(u * steps)) % 2) == (int(floor(v * steps)) % 2) else c2
        return c1

    for row in range(H):
        for col in range(W):
            u = (col + 0.5) / W                      # imagetracer.py:56-58
            v = 1.0 - (row + 0.5) / H
            if persp:                                # camera.py:116-124
                ox, oy, oz = -dist, 0.0, 0.0
                dx, dy, dz = dist, (1.0 - 2 * u) * aspect, 2 * v - 1
            else:                                    # camera.py:70-78
                ox, oy, oz = -1.0, (1.0 - 2 * u) * aspect, 2 * v - 1
                dx, dy, dz = 1.0, 0.0, 0.0
            rox = ox * cm[0] + oy * cm[1] + oz * cm[2] + cm[3]
            roy = ox * cm[4] + oy * cm[5] + oz * cm[6] + cm[7]
            roz = ox * cm[8] + oy * cm[9] + oz * cm[10] + cm[11]
            rdx = dx * cm[0] + dy * cm[1] + dz * cm[2]
            rdy = dx * cm[4] + dy * cm[5] + dz * cm[6]
            rdz = dx * cm[8] + dy * cm[9] + dz * cm[10]
            best_t, best, best_uv = inf, -1, (0.0, 0.0)
            for i in range(n):                       # world.py:56-64
                m = invm[i]
                px = rox * m[0] + roy * m[1] + roz * m[2] + m[3]
                py = rox * m[4] + roy * m[5] + roz * m[6] + m[7]
                pz = rox * m[8] + roy * m[9] + roz * m[10] + m[11]
                qx = rdx * m[0] + rdy * m[1] + rdz * m[2]
                qy = rdx * m[4] + rdy * m[5] + rdz * m[6]
                qz = rdx * m[8] + rdy * m[9] + rdz * m[10]
                if kind[i] == abi.SHAPE_SPHERE:      # shapes.py:103-121
                    a = qx * qx + qy * qy + qz * qz
                    b = 2.0 * (px * qx + py * qy + pz * qz)
                    c = (px * px + py * py + pz * pz) - 1.0
                    delta = b * b - 4.0 * a * c
                    if delta <= 0.0:
                        continue
                    sd = sqrt(delta)
                    t = (-b - sd) / (2.0 * a)
                    if not (1e-5 < t < inf):
                        t = (-b + sd) / (2.0 * a)
                        if not (1e-5 < t < inf):
                            continue
                    if t < best_t:
                        hx, hy, hz = px + qx * t, py + qy * t, pz + qz * t
                        uu = atan2(hy, hx) / (2.0 * pi)          # shapes.py:36-42
                        best_t, best = t, i
                        best_uv = (uu if uu >= 0.0 else uu + 1.0, acos(max(-1.0, min(1.0, hz))) / pi)
                else:                                # shapes.py:168-189
                    if abs(qz) < 1e-5:
                        continue
                    t = -pz / qz
                    if t <= 1e-5 or t >= inf:
                        continue
                    if t < best_t:
                        hx, hy = px + qx * t, py + qy * t
                        best_t, best, best_uv = t, i, (hx - floor(hx), hy - floor(hy))
            if best < 0:
                c3 = bg
            elif onoff:
                c3 = white
            else:                                    # render.py:65-74
                p1, p2 = pigment(pig[best], *best_uv), pigment(emi[best], *best_uv)
                c3 = (p1[0] + p2[0], p1[1] + p2[1], p1[2] + p2[2])
            out[row, col] = c3
    return out, W * H
