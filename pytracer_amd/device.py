"""DeviceScene: a scene resident in the HBM of one MI355X, rendered through the C-ABI."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _lib, abi


class _PinnedBuf:
    """One page-locked host buffer (``pt_host_alloc``) exposed through the array interface; the numpy arrays
    viewing it keep it alive, and it goes back to the pool when the last one is collected."""

    def __init__(self, ptr: int, nbytes: int):
        self.ptr, self.nbytes = ptr, nbytes
        self.__array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 3}

    def __del__(self):
        try:
            _release_pinned(self.ptr, self.nbytes)
        except Exception:
            pass


_free_pinned = {}  # nbytes -> [ptr, ...]; at most _POOL_KEEP idle buffers per size stay allocated
_POOL_KEEP = 2


def _release_pinned(ptr: int, nbytes: int) -> None:
    idle = _free_pinned.setdefault(nbytes, [])
    if len(idle) < _POOL_KEEP:
        idle.append(ptr)
    else:
        _lib.lib().pt_host_free(C.c_void_p(ptr))


def free_pinned_pool() -> None:
    """Release every idle page-locked buffer of the pool (also registered to run at interpreter exit)."""
    for idle in _free_pinned.values():
        while idle:
            try:
                _lib.lib().pt_host_free(C.c_void_p(idle.pop()))
            except Exception:
                pass


import atexit  # noqa: E402

atexit.register(free_pinned_pool)


def pinned_empty(shape, dtype) -> np.ndarray:
    """``np.empty(shape, dtype)`` in page-locked host memory: the destination ``pt_render`` copies into at link
    speed (a pageable array costs about half again as much per frame)."""
    dtype = np.dtype(dtype)
    nbytes = int(np.prod(shape)) * dtype.itemsize
    if nbytes == 0:
        return np.empty(shape, dtype=dtype)
    idle = _free_pinned.get(nbytes)
    if idle:
        ptr = idle.pop()
    else:
        p = C.c_void_p()
        _lib.check(_lib.lib().pt_host_alloc(nbytes, C.byref(p)))
        ptr = int(p.value)
    return np.asarray(_PinnedBuf(ptr, nbytes)).view(dtype).reshape(shape)


class DeviceScene:
    """Owns a ``pt_scene`` handle (``pt_scene_upload`` / ``pt_scene_free``)."""

    def __init__(self, scene: abi.FlatScene, device: int = 0):
        self._h = C.c_void_p()
        self.flat = scene
        self.device = device
        d = scene.desc()
        _lib.check(_lib.lib().pt_scene_upload(C.byref(d), int(device), C.byref(self._h)))

    def clone(self) -> "DeviceScene":
        """A further handle on the same scene (``pt_scene_clone``): shares the uploaded tables, has its own per-camera
        constants, queues and workspace -- so that frames rendered through different handles may be in flight at once,
        each on its own stream (``pytracer_amd.pipeline.FramePipeline``)."""
        other = DeviceScene.__new__(DeviceScene)
        other._h = C.c_void_p()
        other.flat = self.flat
        other.device = self.device
        _lib.check(_lib.lib().pt_scene_clone(self._h, C.byref(other._h)))
        return other

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().pt_scene_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def output_shape(params: abi.Params) -> Tuple[int, int, int]:
        return int(_lib.lib().pt_rows_for_rank(C.byref(params))), int(params.width), 3

    def render(self, cam: abi.Camera, params: abi.Params, pinned: bool = False) -> np.ndarray:
        """Kernel + device->host copy; returns ``[rows_for_rank, W, 3]`` (fp64 or fp32) as an ordinary numpy array, or
        with ``pinned=True`` in page-locked memory from a small pool (for callers that consume the frame at once: the
        array's memory goes back to the pool when it is collected; ``free_pinned_pool()`` releases the idle buffers)."""
        dt = np.float64 if params.out_format == abi.OUT_F64 else np.float32
        out = pinned_empty(self.output_shape(params), dt) if pinned else np.empty(self.output_shape(params), dtype=dt)
        _lib.check(_lib.lib().pt_render(self._h, C.byref(cam), C.byref(params),
                                        out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def render_into(self, cam: abi.Camera, params: abi.Params, dev_ptr: int, nbytes: int,
                    stream: Optional[int] = None) -> None:
        """Render into caller-owned device memory (e.g. a torch tensor's ``data_ptr()``) on ``stream``
        (a ``hipStream_t`` value; ``None`` = the library's stream, synchronous)."""
        _lib.check(_lib.lib().pt_render_device(self._h, C.byref(cam), C.byref(params), C.c_void_p(dev_ptr),
                                               nbytes, C.c_void_p(stream) if stream else None))

    def cull_probe(self, cam: abi.Camera, width: int, height: int, x0: int, x1: int, row0: int, row1: int,
                   pixel=None) -> np.ndarray:
        """Diagnostics: which shapes (by ``World.shapes`` index) the conservative cull of the primary rays through
        the image rectangle ``[x0, x1] x [row0, row1 + 1]`` keeps -- or, with ``pixel=(x, row)``, the cone of that one
        pixel inside the rectangle -- evaluated on the device exactly as the render kernels evaluate it."""
        keep = np.zeros(max(1, self.flat.n_shapes), dtype=np.int32)
        px, py = pixel if pixel is not None else (-1, -1)
        _lib.check(_lib.lib().pt_debug_cull_probe(self._h, C.byref(cam), int(width), int(height), int(x0), int(x1), int(row0),
                                                  int(row1), int(px), int(py), keep.ctypes.data_as(C.c_void_p)))
        return keep[: self.flat.n_shapes].astype(bool)

    def hit_probe(self, rays, shape_index: int = -1) -> np.ndarray:
        """Diagnostics (include/ptrace_debug.h): ``Shape.ray_intersection`` of shape ``shape_index`` of ``World.shapes``
        (or, with -1, ``World.ray_intersection``) for ``[n, 8]`` rays (origin, dir, tmin, tmax), evaluated by the
        kernels' own query and hit-record code.  -> ``[n, 12]``: hit, t, world point, normalised normal, u, v, index."""
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 8)
        out = np.zeros((rays.shape[0], 12), dtype=np.float64)
        _lib.check(_lib.lib().pt_debug_hit_probe(self._h, int(shape_index), rays.ctypes.data_as(C.c_void_p), rays.shape[0],
                                                 out.ctypes.data_as(C.c_void_p)))
        return out

    def lanes_probe(self, rays, anyhit: bool = False) -> np.ndarray:
        """Diagnostics (include/ptrace_debug.h): the same rays through the query the scattered and shadow rays use
        (conservative fp32 filter or grid walk, then exact visits).  -> ``[n, 4]``: hit (or blocked), t, index, 0."""
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 8)
        out = np.zeros((rays.shape[0], 4), dtype=np.float64)
        _lib.check(_lib.lib().pt_debug_lanes_probe(self._h, int(bool(anyhit)), rays.ctypes.data_as(C.c_void_p), rays.shape[0],
                                                   out.ctypes.data_as(C.c_void_p)))
        return out

    def sync(self) -> None:
        _lib.check(_lib.lib().pt_sync(self._h))

    def set_count_rays(self, enable: bool) -> None:
        _lib.check(_lib.lib().pt_set_count_rays(self._h, int(bool(enable))))

    def set_dome_shortcut(self, enable: bool) -> None:
        """Measurement switch: off = every primary ray is generated and traced (same image)."""
        _lib.check(_lib.lib().pt_set_dome_shortcut(self._h, int(bool(enable))))

    def set_timing(self, enable: bool) -> None:
        """hipEvent pair around each render kernel on/off (off: frames run back to back)."""
        _lib.check(_lib.lib().pt_set_timing(self._h, int(bool(enable))))

    def profile_begin(self, capacity: int) -> None:
        """Bracket every following render kernel with its own hipEvent pair (up to ``capacity``)."""
        _lib.check(_lib.lib().pt_profile_begin(self._h, int(capacity)))

    def profile_end(self) -> Tuple[float, int]:
        """-> (summed kernel time in ms, launches) since ``profile_begin``; synchronises."""
        total, n = C.c_double(0.0), C.c_int(0)
        _lib.check(_lib.lib().pt_profile_end(self._h, C.byref(total), C.byref(n)))
        return float(total.value), int(n.value)

    def stats(self) -> abi.Stats:
        st = abi.Stats()
        _lib.check(_lib.lib().pt_get_stats(self._h, C.byref(st)))
        return st

    def handed_over(self) -> Tuple[int, int]:
        """Diagnostics, ``num_of_rays > 1``: (pixels the one-queue kernel handed to the tree kernel in the last frame, the ray
        budget the device derived from the frame's flagged pixels)."""
        n, b = C.c_ulonglong(0), C.c_ulonglong(0)
        _lib.check(_lib.lib().pt_debug_handed_over(self._h, C.byref(n), C.byref(b)))
        return int(n.value), int(b.value)


def device_info(device: int = 0) -> Tuple[int, int]:
    """-> (compute units, peak shader clock in kHz)."""
    cu, khz = C.c_int(0), C.c_int(0)
    _lib.check(_lib.lib().pt_device_info(int(device), C.byref(cu), C.byref(khz)))
    return int(cu.value), int(khz.value)


def device_count() -> int:
    return int(_lib.lib().pt_device_count())


def plan(flat, cam: abi.Camera, params: abi.Params, n_cu: int = 256, dome_shortcut: bool = True) -> abi.PlanInfo:
    """What ``pt_render`` would launch for this scene, camera and parameters -- kernels, grids, LDS, frame-stack home,
    thresholds -- from the library's host-side planner (``pt_debug_plan``, csrc/pt_plan.h).  Touches no device."""
    info = abi.PlanInfo()
    desc = flat.desc()
    _lib.check(_lib.lib().pt_debug_plan(C.byref(desc), C.byref(cam), C.byref(params), int(n_cu), 1 if dome_shortcut else 0, C.byref(info)))
    return info


def set_tuning(name: str, value: int) -> None:
    """One of the library's debug / measurement switches (csrc/pt_plan.h: PT_TUNING_TABLE), by field or ``PTRACE_*`` name.
    Takes effect for the next frame of every scene of the process; none changes a pixel."""
    _lib.check(_lib.lib().pt_debug_set_tuning(name.encode(), int(value)))


def get_tuning(name: str) -> int:
    v = C.c_longlong(0)
    _lib.check(_lib.lib().pt_debug_get_tuning(name.encode(), C.byref(v)))
    return int(v.value)


def device_kernargs() -> bool:
    """True iff ``HIP_FORCE_DEV_KERNARG`` asks the HIP runtime for kernel arguments in device memory
    (``pytracer_amd.prefer_device_kernargs()`` before the first HIP call; ``pt_device_kernargs``)."""
    return bool(_lib.lib().pt_device_kernargs())


def camera_probe(cam: abi.Camera, width: int, height: int, pix) -> np.ndarray:
    """Diagnostics: ``ImageTracer.fire_ray`` for ``[n, 4]`` (col, row, u_pixel, v_pixel) -> ``[n, 7]`` (origin, dir, tmin)."""
    pix = np.ascontiguousarray(pix, dtype=np.float64).reshape(-1, 4)
    out = np.zeros((pix.shape[0], 7), dtype=np.float64)
    _lib.check(_lib.lib().pt_debug_camera_probe(C.byref(cam), int(width), int(height), pix.ctypes.data_as(C.c_void_p),
                                                pix.shape[0], out.ctypes.data_as(C.c_void_p)))
    return out


def scatter_probe(rows) -> Tuple[np.ndarray, np.ndarray]:
    """Diagnostics: ``BRDF.scatter_ray`` for ``[n, 12]`` (brdf kind, PCG init_state, init_seq, normal, incoming, point)
    -> (``[n, 7]`` origin, dir, tmin; ``[n]`` generator states after the call)."""
    rows = np.ascontiguousarray(rows, dtype=np.float64).reshape(-1, 12)
    out = np.zeros((rows.shape[0], 7), dtype=np.float64)
    st = np.zeros(rows.shape[0], dtype=np.uint64)
    _lib.check(_lib.lib().pt_debug_scatter_probe(rows.ctypes.data_as(C.c_void_p), rows.shape[0], out.ctypes.data_as(C.c_void_p),
                                                 st.ctypes.data_as(C.c_void_p)))
    return out, st


def probe(op: int, x, y=None) -> np.ndarray:
    """Evaluate a device primitive elementwise (tests: IEEE exactness / ulp distance to libm)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = x if y is None else np.ascontiguousarray(y, dtype=np.float64)
    out = np.empty_like(x)
    _lib.check(_lib.lib().pt_debug_probe(op, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p),
                                         out.ctypes.data_as(C.c_void_p), x.size))
    return out
