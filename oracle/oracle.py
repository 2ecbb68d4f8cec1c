"""ctypes loader for the CPU oracle (``libpt_oracle.so``).

*** TEST INFRASTRUCTURE — NOT PRODUCT CODE ***  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module.  ``pytracer_amd`` never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

from pytracer_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# PT_ORACLE_LIB: load another build of the oracle (tests/test_oracle_sanitized.py points it at the
# ASan/UBSan build, `make -C oracle asan`, in a child process that preloads libasan)
_LIB_PATH = os.environ.get("PT_ORACLE_LIB") or os.path.join(_HERE, "libpt_oracle.so")
_lib = None

_pd = C.POINTER(C.c_double)
_pu64 = C.POINTER(C.c_uint64)


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (``make -C oracle``)."""
    src = os.path.join(_HERE, "pt_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.pto_render.restype = C.c_int
        L.pto_render.argtypes = [C.POINTER(abi.SceneDesc), C.POINTER(abi.Camera), C.POINTER(abi.Params),
                                 C.c_void_p, C.c_size_t, C.c_int, _pu64]
        L.pto_pcg_random.restype = C.c_uint32
        L.pto_pcg_random_float.restype = C.c_double
        L.pto_pcg_init.argtypes = [_pu64, C.c_uint64, C.c_uint64]
        L.pto_pcg_random.argtypes = [_pu64]
        L.pto_pcg_random_float.argtypes = [_pu64]
        L.pto_xform.argtypes = [_pd, C.c_int, _pd, _pd]
        L.pto_shape_intersect.argtypes = [C.POINTER(abi.SceneDesc), C.c_int, _pd, _pd]
        L.pto_shape_quick_intersect.argtypes = [C.POINTER(abi.SceneDesc), C.c_int, _pd]
        L.pto_world_intersect.argtypes = [C.POINTER(abi.SceneDesc), _pd, _pd]
        L.pto_is_point_visible.argtypes = [C.POINTER(abi.SceneDesc), _pd, _pd]
        L.pto_camera_fire_ray.argtypes = [C.POINTER(abi.Camera), C.c_double, C.c_double, _pd]
        L.pto_tracer_fire_ray.argtypes = [C.POINTER(abi.Camera), C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_double, C.c_double, _pd]
        L.pto_onb.argtypes = [_pd, _pd]
        L.pto_scatter.argtypes = [C.c_int, _pu64, _pd, _pd, _pd, C.c_int, _pd]
        L.pto_pigment.argtypes = [C.POINTER(abi.SceneDesc), C.c_int, C.c_int, C.c_double, C.c_double, _pd]
        L.pto_radiance.argtypes = [C.POINTER(abi.SceneDesc), C.POINTER(abi.Params), _pu64, _pd, C.c_int,
                                   _pd, _pu64]
        L.pto_set_sqr_mode.argtypes = [C.c_int]
        _lib = L
    return _lib


SQR_POW, SQR_MUL = 0, 1


def set_sqr_mode(mode: int) -> None:
    """0: ``x**2`` through libm pow (bit-exact with the reference); 1: ``x*x`` (device arithmetic)."""
    lib().pto_set_sqr_mode(int(mode))


def max_threads() -> int:
    return int(lib().pto_max_threads())


def _arr(a, n=None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1)
    if n is not None:
        assert a.shape[0] == n
    return a


def _p(a: np.ndarray):
    return a.ctypes.data_as(_pd)


def render(scene: abi.FlatScene, cam: abi.Camera, params: abi.Params, n_threads: int = 0,
           sqr_mode: int = SQR_POW) -> Tuple[np.ndarray, int]:
    """Run the oracle's ``fire_all_rays``; returns (``[rows, W, 3]`` array, rays counted)."""
    set_sqr_mode(sqr_mode)
    rows = len(abi.rows_for_rank(params.height, params.row_block, params.n_ranks, params.rank))
    dt = np.float64 if params.out_format == abi.OUT_F64 else np.float32
    out = np.zeros((rows, params.width, 3), dtype=dt)
    n_rays = C.c_uint64(0)
    d = scene.desc()
    rc = lib().pto_render(C.byref(d), C.byref(cam), C.byref(params), out.ctypes.data_as(C.c_void_p),
                          out.nbytes, int(n_threads), C.byref(n_rays))
    if rc != 0:
        raise RuntimeError(f"pto_render failed: {abi.ERROR_NAMES.get(rc, rc)}")
    return out, int(n_rays.value)


class Pcg:
    def __init__(self, init_state=42, init_seq=54):
        self.st = (C.c_uint64 * 2)()
        lib().pto_pcg_init(self.st, init_state, init_seq)

    @property
    def state(self):
        return int(self.st[0])

    @property
    def inc(self):
        return int(self.st[1])

    def random(self) -> int:
        return int(lib().pto_pcg_random(self.st))

    def random_float(self) -> float:
        return float(lib().pto_pcg_random_float(self.st))


def xform(m12, what: int, v3) -> np.ndarray:
    m, v, o = _arr(m12, 12), _arr(v3, 3), np.zeros(3)
    lib().pto_xform(_p(m), what, _p(v), _p(o))
    return o


def ray8(origin, direction, tmin=1e-5, tmax=float("inf")) -> np.ndarray:
    return np.array(list(origin) + list(direction) + [tmin, tmax], dtype=np.float64)


def shape_intersect(scene: abi.FlatScene, i: int, ray) -> Optional[np.ndarray]:
    r, o = _arr(ray, 8), np.zeros(10)
    d = scene.desc()
    return o if lib().pto_shape_intersect(C.byref(d), i, _p(r), _p(o)) else None


def shape_quick_intersect(scene: abi.FlatScene, i: int, ray) -> bool:
    r = _arr(ray, 8)
    d = scene.desc()
    return bool(lib().pto_shape_quick_intersect(C.byref(d), i, _p(r)))


def world_intersect(scene: abi.FlatScene, ray) -> Optional[np.ndarray]:
    r, o = _arr(ray, 8), np.zeros(10)
    d = scene.desc()
    return o if lib().pto_world_intersect(C.byref(d), _p(r), _p(o)) else None


def is_point_visible(scene: abi.FlatScene, point, observer) -> bool:
    a, b = _arr(point, 3), _arr(observer, 3)
    d = scene.desc()
    return bool(lib().pto_is_point_visible(C.byref(d), _p(a), _p(b)))


def camera_fire_ray(cam: abi.Camera, u: float, v: float) -> np.ndarray:
    o = np.zeros(8)
    lib().pto_camera_fire_ray(C.byref(cam), u, v, _p(o))
    return o


def tracer_fire_ray(cam: abi.Camera, W, H, col, row, up=0.5, vp=0.5) -> np.ndarray:
    o = np.zeros(8)
    lib().pto_tracer_fire_ray(C.byref(cam), W, H, col, row, up, vp, _p(o))
    return o


def onb(normal) -> np.ndarray:
    n, o = _arr(normal, 3), np.zeros(9)
    lib().pto_onb(_p(n), _p(o))
    return o.reshape(3, 3)


def scatter(brdf_kind: int, pcg: Pcg, in_dir, point, normal, depth: int) -> np.ndarray:
    a, b, c, o = _arr(in_dir, 3), _arr(point, 3), _arr(normal, 3), np.zeros(8)
    lib().pto_scatter(brdf_kind, pcg.st, _p(a), _p(b), _p(c), depth, _p(o))
    return o


def pigment(scene: abi.FlatScene, i: int, emitted: bool, u: float, v: float) -> np.ndarray:
    o = np.zeros(3)
    d = scene.desc()
    lib().pto_pigment(C.byref(d), i, int(emitted), u, v, _p(o))
    return o


def radiance(scene: abi.FlatScene, params: abi.Params, pcg: Pcg, ray, depth: int = 0):
    r, o = _arr(ray, 8), np.zeros(3)
    n = C.c_uint64(0)
    d = scene.desc()
    lib().pto_radiance(C.byref(d), C.byref(params), pcg.st, _p(r), depth, _p(o), C.byref(n))
    return o, int(n.value)


# ---- HdrImage post-processing (next-3) ------------------------------------------------------------------
def pack_pfm(img: np.ndarray, big_endian: bool = False) -> bytes:
    a = np.ascontiguousarray(img, dtype=np.float64)
    h, w = a.shape[:2]
    out = np.zeros(w * h * 12, dtype=np.uint8)
    L = lib()
    L.pto_pack_pfm.argtypes = [_pd, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.pto_pack_pfm(_p(a.reshape(-1)), w, h, int(big_endian), out.ctypes.data_as(C.c_void_p))
    return out.tobytes()


def average_luminosity(img: np.ndarray, delta: float = 1e-10) -> float:
    a = np.ascontiguousarray(img, dtype=np.float64).reshape(-1)
    L = lib()
    L.pto_average_luminosity.restype = C.c_double
    L.pto_average_luminosity.argtypes = [_pd, C.c_longlong, C.c_double]
    return float(L.pto_average_luminosity(_p(a), a.size // 3, delta))


def tonemap(img: np.ndarray, scale: float, clamp: bool = True, gamma: float = 1.0):
    """-> (normalized [+clamped] image, LDR bytes [H, W, 3] uint8)"""
    a = np.array(img, dtype=np.float64, order="C")
    rgb = np.zeros(a.shape, dtype=np.uint8)
    L = lib()
    L.pto_tonemap.argtypes = [_pd, C.c_longlong, C.c_double, C.c_int, C.c_double, C.c_void_p, C.c_int]
    L.pto_tonemap(_p(a.reshape(-1)), a.size, scale, int(clamp), gamma, rgb.ctypes.data_as(C.c_void_p), 1)
    return a, rgb
