"""GpuImageTracer — drop-in for pytracer's ``ImageTracer`` (imagetracer.py:23-110).

Same constructor, ``fire_ray`` and ``fire_all_rays`` signatures.  ``fire_all_rays(renderer)``
flattens ``renderer.world`` / the camera (duck-typed, :mod:`pytracer_amd.flatten`), runs the HIP
kernels through the C-ABI and fills ``image`` in place; it returns ``None`` like the reference.

Differences that follow from running on a GPU, all explicit:

* ``func`` must be one of the reference's renderers (``OnOffRenderer``, ``FlatRenderer``,
  ``PathTracer``, ``PointLightRenderer`` — by class name, from pytracer or
  :mod:`pytracer_amd.hostmodel`).  An arbitrary Python callable cannot run on the device and
  raises ``UnsupportedSceneError``; nothing falls back to a CPU loop.
* Random streams: the reference draws jitter and scattering numbers from two global sequential
  generators in row-major pixel order (imagetracer.py:89-92, render.py:118,128), which is
  inherently serial.  The device uses the per-pixel alignment of SURVEY.md §8c: pixel
  ``i = row*W + col`` owns ``PCG(S0, Q0 + i)`` (``pcg_mode="pixel"``) or each sample owns
  ``PCG(S0, Q0 + i*S² + k)`` (``"sample"``), where (S0, Q0) are the seeds of ``PathTracer.pcg``
  (or of the tracer's ``pcg`` for the other renderers).  Images are deterministic and independent
  of grid, tile or rank layout.
* ``callback`` is invoked once before rendering (as the reference does, imagetracer.py:77-78) and
  once after the frame completes; the device renders a frame in well under ``callback_time_s``.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import abi, flatten
from .device import DeviceScene
from .hostmodel import PCG, Color

_PCG_MODES = {"pixel": abi.PCG_PIXEL, "sample": abi.PCG_SAMPLE}


class _RayView:
    """What ``fire_ray`` returns: origin, dir, tmin, tmax, depth (ray.py:29-44) as plain data."""

    def __init__(self, o, d):
        from .hostmodel import Vec

        self.origin = Vec(*o)
        self.dir = Vec(*d)
        self.tmin = 1e-5
        self.tmax = float("inf")
        self.depth = 0

    def at(self, t):
        from .hostmodel import Vec

        return Vec(self.origin.x + self.dir.x * t, self.origin.y + self.dir.y * t,
                   self.origin.z + self.dir.z * t)


class GpuImageTracer:
    def __init__(self, image, camera, samples_per_side: int = 0, pcg=None, device: int = 0,
                 pcg_mode: str = "pixel"):
        self.image = image
        self.camera = camera
        self.samples_per_side = samples_per_side
        self.pcg = pcg if pcg is not None else PCG()
        self.device = device
        if pcg_mode not in _PCG_MODES:
            raise ValueError(f"pcg_mode must be one of {sorted(_PCG_MODES)}")
        self.pcg_mode = pcg_mode
        self._scene: Optional[DeviceScene] = None
        self._scene_world = None
        self.last_stats: Optional[abi.Stats] = None

    # -- imagetracer.py:48-58 (host arithmetic only: one ray, for inspection/tests) ----------------
    def fire_ray(self, col: int, row: int, u_pixel=0.5, v_pixel=0.5):
        cam = flatten.flatten_camera(self.camera)
        u = (col + u_pixel) / self.image.width
        v = 1.0 - (row + v_pixel) / self.image.height
        m = list(cam.m)
        if cam.kind == abi.CAMERA_PERSPECTIVE:
            o = (-cam.screen_distance, 0.0, 0.0)
            d = (cam.screen_distance, (1.0 - 2 * u) * cam.aspect_ratio, 2 * v - 1)
        else:
            o = (-1.0, (1.0 - 2 * u) * cam.aspect_ratio, 2 * v - 1)
            d = (1.0, 0.0, 0.0)
        wo = tuple(o[0] * m[4 * r] + o[1] * m[4 * r + 1] + o[2] * m[4 * r + 2] + m[4 * r + 3] for r in range(3))
        wd = tuple(d[0] * m[4 * r] + d[1] * m[4 * r + 1] + d[2] * m[4 * r + 2] for r in range(3))
        return _RayView(wo, wd)

    # -- imagetracer.py:60-110 ------------------------------------------------------------------------
    def fire_all_rays(self, func, callback=None, callback_time_s: float = 2.0, **callback_kwargs) -> None:
        if callback:
            callback(col=0, row=0, **callback_kwargs)
        w, h = int(self.image.width), int(self.image.height)
        params = flatten.renderer_params(func, w, h, samples_per_side=int(self.samples_per_side),
                                         tracer_pcg=self.pcg, pcg_mode=_PCG_MODES[self.pcg_mode])
        cam = flatten.flatten_camera(self.camera)
        world = func.world
        if self._scene is None or self._scene_world is not world:
            if self._scene is not None:
                self._scene.close()
            self._scene = DeviceScene(flatten.flatten_world(world), self.device)
            self._scene_world = world
        out = self._scene.render(cam, params)  # [H, W, 3] fp64, row 0 = top (hdrimages.py:78-80)
        self.last_stats = self._scene.stats()
        _fill_image(self.image, out)
        if callback:
            callback(col=w - 1, row=h - 1, **callback_kwargs)

    def close(self):
        if self._scene is not None:
            self._scene.close()
            self._scene = None


def _fill_image(image, arr: np.ndarray) -> None:
    """Write ``[H, W, 3]`` into an HdrImage: the stand-in keeps a numpy array; the reference's
    HdrImage holds a list of Color objects (hdrimages.py:70), filled with that class."""
    if hasattr(image, "set_array"):
        image.set_array(arr)
        return
    color_cls = type(image.pixels[0]) if len(image.pixels) else Color
    flat = arr.reshape(-1, 3)  # (map over three lists: ~15 % less interpreter time than unpacking triples)
    image.pixels[:] = list(map(color_cls, flat[:, 0].tolist(), flat[:, 1].tolist(), flat[:, 2].tolist()))
