"""ctypes mirror of ``include/ptrace.h`` plus the numpy-backed flattened scene.

Nothing here computes anything: it only describes memory.  The same structures feed the
product library (``libptrace.so``, HIP) and — in tests only — the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

# ---- enumerations (ptrace.h) ---------------------------------------------------------------
SHAPE_SPHERE, SHAPE_PLANE = 0, 1
BRDF_DIFFUSE, BRDF_SPECULAR = 0, 1
PIGMENT_UNIFORM, PIGMENT_CHECKERED, PIGMENT_IMAGE = 0, 1, 2
CAMERA_ORTHOGONAL, CAMERA_PERSPECTIVE = 0, 1
RENDERER_ONOFF, RENDERER_FLAT, RENDERER_PATHTRACER, RENDERER_POINTLIGHT = 0, 1, 2, 3
PCG_SEQ, PCG_PIXEL, PCG_SAMPLE = 0, 1, 2
OUT_F64, OUT_F32 = 0, 1

PT_OK = 0
KERNEL_NONE, KERNEL_SIMPLE, KERNEL_TILE, KERNEL_TILE4, KERNEL_PATH, KERNEL_PATH_REGIONS, KERNEL_PATH_TREE = range(7)
ERROR_NAMES = {
    -1: "PT_ERR_INVALID",
    -2: "PT_ERR_HIP",
    -3: "PT_ERR_UNSUPPORTED",
    -4: "PT_ERR_NOMEM",
    -5: "PT_ERR_SIZE",
    -6: "PT_ERR_NODEVICE",
}

_pd = C.POINTER(C.c_double)
_pi32 = C.POINTER(C.c_int32)
_pi64 = C.POINTER(C.c_int64)


class SceneDesc(C.Structure):
    """``pt_scene_desc``"""

    _fields_ = [
        ("n_shapes", C.c_int32),
        ("kind", _pi32),
        ("invm", _pd),
        ("m", _pd),
        ("brdf_kind", _pi32),
        ("brdf_param", _pd),
        ("pig_kind", _pi32),
        ("pig_c1", _pd),
        ("pig_c2", _pd),
        ("pig_steps", _pd),
        ("pig_tex", _pi32),
        ("emi_kind", _pi32),
        ("emi_c1", _pd),
        ("emi_c2", _pd),
        ("emi_steps", _pd),
        ("emi_tex", _pi32),
        ("n_lights", C.c_int32),
        ("light_pos", _pd),
        ("light_color", _pd),
        ("light_radius", _pd),
        ("n_textures", C.c_int32),
        ("tex_w", _pi32),
        ("tex_h", _pi32),
        ("tex_offset", _pi64),
        ("tex_data", _pd),
    ]


class Camera(C.Structure):
    """``pt_camera``"""

    _fields_ = [
        ("kind", C.c_int32),
        ("_pad", C.c_int32),
        ("m", C.c_double * 12),
        ("screen_distance", C.c_double),
        ("aspect_ratio", C.c_double),
    ]


class Params(C.Structure):
    """``pt_params``"""

    _fields_ = [
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("samples_per_side", C.c_int32),
        ("renderer", C.c_int32),
        ("background", C.c_double * 3),
        ("onoff_color", C.c_double * 3),
        ("ambient", C.c_double * 3),
        ("num_of_rays", C.c_int32),
        ("max_depth", C.c_int32),
        ("rr_limit", C.c_int32),
        ("pcg_mode", C.c_int32),
        ("jitter_state", C.c_uint64),
        ("jitter_seq", C.c_uint64),
        ("path_state", C.c_uint64),
        ("path_seq", C.c_uint64),
        ("row_block", C.c_int32),
        ("n_ranks", C.c_int32),
        ("rank", C.c_int32),
        ("out_format", C.c_int32),
    ]


class Stats(C.Structure):
    """``pt_stats``"""

    _fields_ = [
        ("n_rays", C.c_uint64),
        ("n_pixels", C.c_uint64),
        ("kernel_ms", C.c_double),
        ("total_ms", C.c_double),
        ("vgprs", C.c_int32),
        ("lds_bytes", C.c_int32),
        ("grid", C.c_int32),
        ("block", C.c_int32),
        ("n_rays_resolved", C.c_uint64),
        ("kernel", C.c_int32),
        ("_reserved", C.c_int32),
    ]


class PlanInfo(C.Structure):
    """``pt_plan_info`` (include/ptrace_debug.h): what a frame will launch, from the host-side planner (csrc/pt_plan.h)."""

    _fields_ = [
        ("kernel", C.c_int32),
        ("rows", C.c_int32),
        ("npix", C.c_int64),
        ("_pre_kernel", C.c_char * 48),
        ("_first_kernel", C.c_char * 64),
        ("_main_kernel", C.c_char * 64),
        ("_alt_kernel", C.c_char * 64),
        ("grid", C.c_int32),
        ("grid_first", C.c_int32),
        ("grid_alt", C.c_int32),
        ("grid4_x", C.c_int32),
        ("grid4_y", C.c_int32),
        ("npx", C.c_int32),
        ("lds_first", C.c_int64),
        ("lds_main", C.c_int64),
        ("lds_alt", C.c_int64),
        ("frame_stack_home", C.c_int32),
        ("alt_frame_stack_home", C.c_int32),
        ("frame_doubles", C.c_int32),
        ("workspace_bytes", C.c_int64),
        ("q_min_flagged", C.c_int64),
        ("wg_per_cu", C.c_int32),
        ("block_h", C.c_int32),
        ("hier", C.c_int32),
        ("ortho", C.c_int32),
        ("hoist", C.c_int32),
        ("tile4_lds", C.c_int32),
        ("n_spheres", C.c_int32),
        ("n_diag", C.c_int32),
        ("has_grid", C.c_int32),
        ("ball_levels", C.c_int32),
        ("units_need", C.c_int32),
        ("nregions", C.c_int32),
        ("min_rounds", C.c_int32),
        ("spec_draws", C.c_int32),
        ("alt_budget", C.c_int32),
        ("_reserved", C.c_int32 * 7),
    ]

    STACK_HOMES = {0: None, 1: "LDS", 2: "HBM", 3: "SPLIT"}

    @property
    def pre_kernel(self) -> str:
        return self._pre_kernel.decode()

    @property
    def first_kernel(self) -> str:
        return self._first_kernel.decode()

    @property
    def main_kernel(self) -> str:
        return self._main_kernel.decode()

    @property
    def alt_kernel(self) -> str:
        return self._alt_kernel.decode()

    @property
    def kernels(self):
        """The kernels of the frame in launch order (names as csrc/pt_plan.h spells them)."""
        return [k for k in (self.pre_kernel, self.first_kernel, self.main_kernel, self.alt_kernel) if k]

    @property
    def frame_stack(self):
        return self.STACK_HOMES[self.frame_stack_home]

    @property
    def alt_frame_stack(self):
        return self.STACK_HOMES[self.alt_frame_stack_home]


def _f64(a, shape=None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


@dataclass
class FlatScene:
    """Structure-of-arrays scene, fp64, in the layout ``pt_scene_desc`` points at.

    ``invm``/``m`` are ``[12, n]`` (element (r, c) of the 3x4 affine block at row ``r*4+c``),
    colours are ``[3, n]``.
    """

    kind: np.ndarray
    invm: np.ndarray
    m: np.ndarray
    brdf_kind: np.ndarray
    brdf_param: np.ndarray
    pig_kind: np.ndarray
    pig_c1: np.ndarray
    pig_c2: np.ndarray
    pig_steps: np.ndarray
    pig_tex: np.ndarray
    emi_kind: np.ndarray
    emi_c1: np.ndarray
    emi_c2: np.ndarray
    emi_steps: np.ndarray
    emi_tex: np.ndarray
    light_pos: np.ndarray = field(default_factory=lambda: np.zeros((3, 0)))
    light_color: np.ndarray = field(default_factory=lambda: np.zeros((3, 0)))
    light_radius: np.ndarray = field(default_factory=lambda: np.zeros((0,)))
    tex_w: np.ndarray = field(default_factory=lambda: np.zeros((0,), np.int32))
    tex_h: np.ndarray = field(default_factory=lambda: np.zeros((0,), np.int32))
    tex_offset: np.ndarray = field(default_factory=lambda: np.zeros((0,), np.int64))
    tex_data: np.ndarray = field(default_factory=lambda: np.zeros((0,)))

    _F64 = ("invm", "m", "brdf_param", "pig_c1", "pig_c2", "pig_steps", "emi_c1", "emi_c2",
            "emi_steps", "light_pos", "light_color", "light_radius", "tex_data")
    _I32 = ("kind", "brdf_kind", "pig_kind", "pig_tex", "emi_kind", "emi_tex", "tex_w", "tex_h")

    def __post_init__(self):
        for k in self._F64:
            setattr(self, k, _f64(getattr(self, k)))
        for k in self._I32:
            setattr(self, k, _i32(getattr(self, k)))
        self.tex_offset = np.ascontiguousarray(self.tex_offset, dtype=np.int64)
        n = self.n_shapes
        assert self.invm.shape == (12, n) and self.m.shape == (12, n), "matrices must be [12, n]"
        for k in ("pig_c1", "pig_c2", "emi_c1", "emi_c2"):
            assert getattr(self, k).shape == (3, n), f"{k} must be [3, n]"
        for k in ("brdf_kind", "brdf_param", "pig_kind", "pig_steps", "pig_tex", "emi_kind",
                  "emi_steps", "emi_tex"):
            assert getattr(self, k).shape == (n,), f"{k} must be [n]"
        nl = self.n_lights
        assert self.light_pos.shape == (3, nl) and self.light_color.shape == (3, nl)

    @property
    def n_shapes(self) -> int:
        return int(self.kind.shape[0])

    @property
    def n_lights(self) -> int:
        return int(self.light_radius.shape[0])

    @property
    def n_textures(self) -> int:
        return int(self.tex_w.shape[0])

    def desc(self) -> SceneDesc:
        """A ``pt_scene_desc`` pointing into this object's arrays (keep ``self`` alive)."""
        d = SceneDesc()
        d.n_shapes = self.n_shapes
        d.n_lights = self.n_lights
        d.n_textures = self.n_textures
        for k in self._F64:
            setattr(d, k, getattr(self, k).ctypes.data_as(_pd))
        for k in self._I32:
            setattr(d, k, getattr(self, k).ctypes.data_as(_pi32))
        d.tex_offset = self.tex_offset.ctypes.data_as(_pi64)
        return d

    # -- (de)serialisation used by the golden fixtures -----------------------------------------
    _ALL = _F64 + _I32 + ("tex_offset",)

    def to_dict(self, prefix: str = "scene_") -> dict:
        return {prefix + k: getattr(self, k) for k in self._ALL}

    @classmethod
    def from_dict(cls, d, prefix: str = "scene_") -> "FlatScene":
        return cls(**{k: np.array(d[prefix + k]) for k in cls._ALL})

    def same_bits(self, other: "FlatScene") -> bool:
        """True when every array is bit-identical."""
        for k in self._ALL:
            a, b = getattr(self, k), getattr(other, k)
            if a.shape != b.shape or a.tobytes() != b.tobytes():
                return False
        return True


def make_camera(kind: int, m12, screen_distance: float, aspect_ratio: float) -> Camera:
    c = Camera()
    c.kind = int(kind)
    mm = _f64(m12).reshape(-1)
    assert mm.shape[0] == 12
    for i in range(12):
        c.m[i] = float(mm[i])
    c.screen_distance = float(screen_distance)
    c.aspect_ratio = float(aspect_ratio)
    return c


def camera_to_dict(c: Camera, prefix: str = "cam_") -> dict:
    return {
        prefix + "kind": np.int32(c.kind),
        prefix + "m": np.array(list(c.m), dtype=np.float64),
        prefix + "screen_distance": np.float64(c.screen_distance),
        prefix + "aspect_ratio": np.float64(c.aspect_ratio),
    }


def camera_from_dict(d, prefix: str = "cam_") -> Camera:
    return make_camera(int(d[prefix + "kind"]), d[prefix + "m"], float(d[prefix + "screen_distance"]),
                       float(d[prefix + "aspect_ratio"]))


def make_params(width: int, height: int, renderer: int, samples_per_side: int = 0,
                background=(0.0, 0.0, 0.0), onoff_color=(1.0, 1.0, 1.0), ambient=(0.1, 0.1, 0.1),
                num_of_rays: int = 10, max_depth: int = 10, rr_limit: int = 3,
                pcg_mode: int = PCG_PIXEL, jitter_state: int = 42, jitter_seq: int = 54,
                path_state: int = 42, path_seq: int = 54, row_block: int = 8, n_ranks: int = 1,
                rank: int = 0, out_format: int = OUT_F64) -> Params:
    p = Params()
    p.width, p.height = int(width), int(height)
    p.samples_per_side = int(samples_per_side)
    p.renderer = int(renderer)
    for i in range(3):
        p.background[i] = float(background[i])
        p.onoff_color[i] = float(onoff_color[i])
        p.ambient[i] = float(ambient[i])
    p.num_of_rays, p.max_depth, p.rr_limit = int(num_of_rays), int(max_depth), int(rr_limit)
    p.pcg_mode = int(pcg_mode)
    p.jitter_state, p.jitter_seq = int(jitter_state), int(jitter_seq)
    p.path_state, p.path_seq = int(path_state), int(path_seq)
    p.row_block, p.n_ranks, p.rank = int(row_block), int(n_ranks), int(rank)
    p.out_format = int(out_format)
    return p


def copy_params(p: Params, **changes) -> Params:
    q = Params()
    C.memmove(C.byref(q), C.byref(p), C.sizeof(Params))
    for k, v in changes.items():
        setattr(q, k, v)
    return q


def rows_for_rank(height: int, row_block: int, n_ranks: int, rank: int) -> List[int]:
    """Global row indices rank ``rank`` owns (ascending): block b -> rank b % n_ranks."""
    rb = max(1, int(row_block))
    nr = max(1, int(n_ranks))
    return [r for r in range(height) if (r // rb) % nr == rank]
