"""GpuImageTracer — drop-in for pytracer's ``ImageTracer`` (imagetracer.py:23-110).

Same constructor, ``fire_ray`` and ``fire_all_rays`` signatures.  ``fire_all_rays(renderer)``
flattens ``renderer.world`` / the camera (duck-typed, :mod:`pytracer_amd.flatten`), runs the HIP
kernels through the C-ABI and fills ``image`` in place; it returns ``None`` like the reference.

What ``func`` may be:

* one of the reference's renderers (``OnOffRenderer``, ``FlatRenderer``, ``PathTracer``,
  ``PointLightRenderer`` — by class name, from pytracer or :mod:`pytracer_amd.hostmodel`): the frame
  is rendered on the MI355X.  A renderer whose world the device cannot express (unknown shape / BRDF /
  pigment class, non-affine matrix) raises ``UnsupportedSceneError``: a renderer never silently runs
  anywhere else.
* any other callable ``Ray -> Color`` (the lambdas of the reference's own ``TestImageTracer``,
  test_all.py:576-604): there is nothing to put on a GPU, the contract is "call ``func`` once per
  sample, in the reference's order, and store what it returns".  That per-pixel loop runs on the host
  in this class (``_host_loop``), drawing the jitter numbers from ``self.pcg`` exactly as
  imagetracer.py:80-104 does.

Random streams on the device (``pcg_mode``).  The reference draws jitter and scattering numbers from two global
sequential generators in row-major pixel order (imagetracer.py:89-92, render.py:118,128).

* ``"seq"`` -- the reference's own streams.  For ``OnOffRenderer`` / ``FlatRenderer`` / ``PointLightRenderer`` only the
  jitter stream exists and every sample draws exactly two numbers from it, so sample ``k`` of pixel ``i`` starts
  ``2 (i S² + k)`` draws into the tracer's ``pcg``: the device enters the stream there by jump-ahead and the frame is
  the one ``ImageTracer(image, camera, S, pcg).fire_all_rays(renderer)`` computes, bit for bit; afterwards
  ``tracer.pcg`` has advanced by the ``2 W H S²`` draws the reference would have made.  Refused for ``PathTracer``
  (its scattering stream is serial by construction).
* ``"pixel"`` -- pixel ``i = row*W + col`` owns ``PCG(S0, Q0 + i)`` for jitter and scattering in program order;
  ``"sample"`` -- each sample owns ``PCG(S0, Q0 + i*S² + k)``, where (S0, Q0) are the seeds of
  ``PathTracer.pcg`` (or of the tracer's ``pcg`` for the other renderers).
* ``"auto"`` (default) -- ``"seq"`` where it is exact (the three renderers without a scattering stream), ``"sample"`` for
  the path tracer: no device alignment can BE the reference's serial scattering stream, both per-pixel and per-sample
  generators are pinned by frames the reference itself rendered that way, and with a generator per sample a pixel's
  samples are independent -- 3x faster on BASELINE's C3.  With one sample per pixel (the CLI's default)
  the two coincide.

Images are deterministic and independent of grid, tile or rank layout in every mode.

``callback(col=, row=, **kw)`` is invoked once before rendering and then whenever more than
``callback_time_s`` have passed since the last call (imagetracer.py:76-78, 106-110), with the last
pixel traced so far.  On the device a frame is rendered in horizontal bands (a few per frame, halved
while a band takes longer than half of ``callback_time_s``) so that long frames report progress;
without a callback the frame is one launch.  The clock is wall time (the reference uses
``process_time``, which stands still while the host waits for the GPU).

The scene is re-flattened on every call, as the reference re-reads ``World.shapes`` on every call;
the device copy is reused only when the flattened arrays are bit-identical to the ones uploaded.

``resident=True`` (what the ``render`` command uses): the frame is left in HBM as ``tracer.device_image`` (a
:class:`pytracer_amd.postprocess.DeviceImage`, fp64 like the reference's ``HdrImage``) instead of being copied
into ``image`` -- the PFM floats and the tone-mapped bytes are then the only device-to-host traffic of a render
(main.py:203-213 on the device; 14 MB instead of the 22 MB fp64 frame at 1280x720).  ``image`` keeps its size and is
filled on demand by ``tracer.download()``.

A reference-style ``HdrImage`` (a list of ``Color`` objects, hdrimages.py:70) has that list filled IN PLACE, one new
``Color`` of the image's own class per pixel, as the reference's ``set_pixel`` loop does (0.4 - 0.5 s per 720p frame in the
interpreter; a caller that kept ``px = image.pixels`` sees the frame).  ``lazy_pixels=True`` (opt-in; round 4 did this by
default) REPLACES ``image.pixels`` by a :class:`pytracer_amd.pixels.LazyPixels` instead: it indexes, iterates,
assigns and keeps object identity like the list did and makes a ``Color`` when an index is first read -- 0.001 ms per frame,
for callers that only index / iterate / write the image out.

``tracer.pcg`` is advanced behind a ``"seq"`` frame by the ``2 W H S²`` draws the reference's loop makes
(imagetracer.py:84-101), whether the caller supplied the generator or the tracer made its own ``PCG()``: a second frame
through the same tracer continues the stream, as a second ``ImageTracer.fire_all_rays`` does.  What is NOT imitated is the
reference's default ARGUMENT being one object shared by every ``ImageTracer`` of a process (imagetracer.py:34):
each tracer built without ``pcg`` gets a fresh ``PCG()``.

``fallback="host"`` (opt-in, never the default): a reference renderer whose world the device cannot express (an
unknown shape / BRDF / pigment class, a non-affine matrix) is itself a callable ``Ray -> Color``; with this option it
runs through the host loop above -- the reference's own code computing every radiance, exactly as
``ImageTracer.fire_all_rays`` would -- instead of raising ``UnsupportedSceneError``.  Without the option nothing
ever leaves the device path silently.
"""
from __future__ import annotations

from time import perf_counter, process_time
from typing import Optional

import numpy as np

from . import abi, flatten
from .device import DeviceScene
from .hostmodel import PCG, Color, pcg_advance
from .pixels import LazyPixels

_PCG_MODES = {"auto": None, "seq": abi.PCG_SEQ, "pixel": abi.PCG_PIXEL, "sample": abi.PCG_SAMPLE}


class _RayView:
    """What ``fire_ray`` returns when the camera is a plain parameter holder: origin, dir, tmin, tmax,
    depth and ``at`` (ray.py:29-57) as plain data."""

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
                 pcg_mode: str = "auto", resident: bool = False, fallback: Optional[str] = None,
                 lazy_pixels: bool = False, eager_fill: Optional[bool] = None):
        if fallback not in (None, "host"):
            raise ValueError('fallback must be None or "host"')
        if eager_fill is not None:  # (round 4's spelling of the same switch, with the opposite default)
            lazy_pixels = not eager_fill
        self.lazy_pixels = bool(lazy_pixels)
        self.resident = bool(resident)
        self.fallback = fallback
        self.device_image = None   # resident=True: the last frame, in HBM
        self.last_path = None      # "device" | "host": where the last frame was computed
        self.image = image
        self.camera = camera
        self.samples_per_side = samples_per_side
        self.pcg = pcg if pcg is not None else PCG()
        self.device = device
        if pcg_mode not in _PCG_MODES:
            raise ValueError(f"pcg_mode must be one of {sorted(_PCG_MODES)}")
        self.pcg_mode = pcg_mode
        self._scene: Optional[DeviceScene] = None
        self.last_stats: Optional[abi.Stats] = None
        self.last_bands = 0

    # -- imagetracer.py:48-58 ---------------------------------------------------------------------------
    def fire_ray(self, col: int, row: int, u_pixel=0.5, v_pixel=0.5):
        u = (col + u_pixel) / self.image.width
        v = 1.0 - (row + v_pixel) / self.image.height
        if callable(getattr(self.camera, "fire_ray", None)):
            return self.camera.fire_ray(u, v)  # the reference's own camera object: its own Ray
        cam = flatten.flatten_camera(self.camera)
        m = list(cam.m)
        if cam.kind == abi.CAMERA_PERSPECTIVE:  # camera.py:116-124
            o = (-cam.screen_distance, 0.0, 0.0)
            d = (cam.screen_distance, (1.0 - 2 * u) * cam.aspect_ratio, 2 * v - 1)
        else:  # camera.py:70-78
            o = (-1.0, (1.0 - 2 * u) * cam.aspect_ratio, 2 * v - 1)
            d = (1.0, 0.0, 0.0)
        wo = tuple(o[0] * m[4 * r] + o[1] * m[4 * r + 1] + o[2] * m[4 * r + 2] + m[4 * r + 3] for r in range(3))
        wd = tuple(d[0] * m[4 * r] + d[1] * m[4 * r + 1] + d[2] * m[4 * r + 2] for r in range(3))
        return _RayView(wo, wd)

    # -- imagetracer.py:60-110 ------------------------------------------------------------------------
    def fire_all_rays(self, func, callback=None, callback_time_s: float = 2.0, **callback_kwargs) -> None:
        if flatten.is_device_renderer(func):
            try:
                self._device_frame(func, callback, callback_time_s, callback_kwargs)
                self.last_path = "device"
            except flatten.UnsupportedSceneError:
                # (raised while flattening: before anything was rendered or written)
                if self.fallback != "host" or not callable(func):
                    raise
                self.device_image = None
                self._host_loop(func, callback, callback_time_s, callback_kwargs)
                self.last_path = "host"
        elif callable(func):
            self._host_loop(func, callback, callback_time_s, callback_kwargs)
            self.last_path = "host"
        else:
            raise TypeError(f"func must be a renderer or a callable Ray -> Color, not {type(func).__name__}")

    def _host_loop(self, func, callback, callback_time_s, callback_kwargs) -> None:
        """imagetracer.py:76-110 for a ``func`` that is not a renderer: the same calls in the same order."""
        image, S = self.image, int(self.samples_per_side)
        last_call_time = process_time()
        if callback:
            callback(col=0, row=0, **callback_kwargs)
        for row in range(image.height):
            for col in range(image.width):
                if S > 0:
                    r = g = b = 0.0
                    cls = Color
                    for inter_pixel_row in range(S):
                        for inter_pixel_col in range(S):
                            u_pixel = (inter_pixel_col + self.pcg.random_float()) / S  # drawn first
                            v_pixel = (inter_pixel_row + self.pcg.random_float()) / S
                            c = func(self.fire_ray(col=col, row=row, u_pixel=u_pixel, v_pixel=v_pixel))
                            cls = type(c)
                            r, g, b = r + c.r, g + c.g, b + c.b  # Color.__add__ (colors.py:34)
                    k = 1 / S ** 2
                    image.set_pixel(col, row, cls(r * k, g * k, b * k))  # Color.__mul__ by a scalar
                else:
                    image.set_pixel(col, row, func(self.fire_ray(col=col, row=row)))
                current_time = process_time()
                if callback and (current_time - last_call_time > callback_time_s):
                    callback(col=col, row=row, **callback_kwargs)
                    last_call_time = current_time

    def _device_scene(self, world) -> DeviceScene:
        flat = flatten.flatten_world(world)
        if self._scene is not None and not self._scene.flat.same_bits(flat):
            self._scene.close()
            self._scene = None
        if self._scene is None:
            self._scene = DeviceScene(flat, self.device)
        return self._scene

    def _device_frame(self, func, callback, callback_time_s, callback_kwargs) -> None:
        last_call_time = perf_counter()
        if callback:
            callback(col=0, row=0, **callback_kwargs)
        w, h = int(self.image.width), int(self.image.height)
        mode = _PCG_MODES[self.pcg_mode]
        if mode is None:  # "auto": the reference's own stream where the device can enter it anywhere
            mode = abi.PCG_SAMPLE if flatten.renderer_kind(func) == abi.RENDERER_PATHTRACER else abi.PCG_SEQ
        params = flatten.renderer_params(func, w, h, samples_per_side=int(self.samples_per_side),
                                         tracer_pcg=self.pcg, pcg_mode=mode)
        cam = flatten.flatten_camera(self.camera)
        scene = self._device_scene(func.world)
        dev_t = None
        if self.resident:
            from .devmem import DeviceBuffer  # HBM through the C-ABI (pt_device_alloc): no GPU framework needed

            dev_t = DeviceBuffer((h, w, 3), np.float64, self.device)
            params = abi.copy_params(params, out_format=abi.OUT_F64)
        row_bytes = w * 3 * 8

        def render_rows(p, row0):
            """One launch for the partition written into `p`; -> its rows (host array, or None when resident)."""
            if dev_t is None:
                return scene.render(cam, p)
            rows = DeviceScene.output_shape(p)[0]
            scene.render_into(cam, p, dev_t.data_ptr() + row0 * row_bytes, rows * row_bytes, None)
            return rows

        if not callback or h <= 8:
            out = render_rows(params, 0)  # [H, W, 3] fp64, row 0 = top (hdrimages.py:78-80)
            self.last_stats = scene.stats()
            self.last_bands = 1
        else:
            # Bands of 2^k rows starting at multiples of their height (so a band is "block r0/L of L-row
            # blocks", which the partition fields of pt_params express); per-pixel seeds make the frame
            # independent of how it is cut.  A frame starts as TWO bands (the first one tells how long a band takes)
            # and is cut finer only while a band takes more than half of callback_time_s.
            out = np.empty((h, w, 3), dtype=np.float64) if dev_t is None else None
            band = 1
            while band * 2 < h:
                band *= 2
            row0, n_rays, n_res, kernel_ms, total_ms, self.last_bands = 0, 0, 0, 0.0, 0.0, 0
            st = None
            while row0 < h:
                while row0 % band:
                    band //= 2
                p = abi.copy_params(params, row_block=band, n_ranks=(h + band - 1) // band, rank=row0 // band)
                t0 = perf_counter()
                shard = render_rows(p, row0)
                if dev_t is None:
                    out[row0:row0 + shard.shape[0]] = shard
                    row0 += shard.shape[0]
                else:
                    row0 += shard
                st = scene.stats()  # (waits for the band)
                dt = perf_counter() - t0
                n_rays, n_res = n_rays + st.n_rays, n_res + st.n_rays_resolved
                kernel_ms, total_ms = kernel_ms + st.kernel_ms, total_ms + st.total_ms
                self.last_bands += 1
                now = perf_counter()
                if row0 < h and now - last_call_time > callback_time_s:
                    callback(col=w - 1, row=row0 - 1, **callback_kwargs)
                    last_call_time = now
                if dt > 0.5 * callback_time_s and band > 1:
                    band //= 2
            st.n_rays, st.n_rays_resolved, st.kernel_ms, st.total_ms, st.n_pixels = n_rays, n_res, kernel_ms, total_ms, w * h
            self.last_stats = st
        if mode == abi.PCG_SEQ and int(self.samples_per_side) > 0 and hasattr(self.pcg, "state") and hasattr(self.pcg, "inc"):
            # the reference's loop leaves ImageTracer.pcg 2 W H S^2 draws further on (imagetracer.py:84-101)
            self.pcg.state = pcg_advance(int(self.pcg.state), int(self.pcg.inc), 2 * w * h * int(self.samples_per_side) ** 2)
        if dev_t is not None:
            from .postprocess import DeviceImage

            self.device_image = DeviceImage(dev_t)
        else:
            self.device_image = None
            _fill_image(self.image, out, self.lazy_pixels)

    def download(self) -> None:
        """resident=True: copy the frame left in HBM into ``image`` (what ``fire_all_rays`` does by itself otherwise)."""
        if self.device_image is None:
            raise RuntimeError("no resident frame: fire_all_rays(renderer) with resident=True leaves one")
        _fill_image(self.image, self.device_image.numpy(), self.lazy_pixels)

    def close(self):
        if self._scene is not None:
            self._scene.close()
            self._scene = None


def _fill_image(image, arr: np.ndarray, lazy: bool = False) -> None:
    """Write ``[H, W, 3]`` into an HdrImage: the stand-in keeps a numpy array; the reference's HdrImage holds a list
    of Color objects (hdrimages.py:70), which is filled IN PLACE (the same list object, one new Color of the image's own
    class per pixel: what the reference's ``set_pixel`` loop leaves behind; 0.4 - 0.5 s per 720p frame).  ``lazy=True``
    REPLACES the list by a :class:`pytracer_amd.pixels.LazyPixels` over ``arr`` (which the caller hands over): same
    indexing, iteration, assignment and identity semantics, a ``Color`` made when an index is first read."""
    if hasattr(image, "set_array"):
        image.set_array(arr)
        return
    old = image.pixels
    color_cls = old.color_cls if isinstance(old, LazyPixels) else (type(old[0]) if len(old) else Color)
    if lazy:
        image.pixels = LazyPixels(np.ascontiguousarray(arr, dtype=np.float64), color_cls)
        return
    flat = arr.reshape(-1, 3)  # (map over three lists: ~15 % less interpreter time than unpacking triples)
    colors = list(map(color_cls, flat[:, 0].tolist(), flat[:, 1].tolist(), flat[:, 2].tolist()))
    if isinstance(old, LazyPixels):
        image.pixels = colors
    else:
        old[:] = colors
