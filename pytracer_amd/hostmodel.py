"""Minimal host-side scene objects, attribute-compatible with pytracer's.

The product is a drop-in behind pytracer's ``ImageTracer.fire_all_rays``: users hand it
pytracer's own ``World`` / ``Camera`` / ``Renderer`` objects and :mod:`pytracer_amd.flatten`
reads them by attribute name (duck typing).  pytracer itself cannot travel to the GPU box,
so tests, ``smoke()`` and ``bench.py`` describe their scenes with the tiny stand-ins below,
which expose exactly the attributes the flattener looks at — nothing more.  None of these
classes can trace a ray: all radiance computation happens on the device.

Attribute names follow the reference so both kinds of object flatten through one code path:
``Shape.transformation.m / .invm`` (transformations.py:48-56), ``Shape.material.brdf.pigment``,
``Material.emitted_radiance`` (materials.py:199-204), ``World.shapes / .point_lights``
(world.py:38-49), camera fields (camera.py:48-57, 87-101), renderer fields (render.py:26-97).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Sequence

import numpy as np


@dataclass
class Vec:
    x: float = 0.0
    y: float = 0.0
    z: float = 0.0


Point = Vec  # positions and directions are both plain xyz triples on the host side


@dataclass
class Color:
    r: float = 0.0
    g: float = 0.0
    b: float = 0.0


BLACK = Color(0.0, 0.0, 0.0)
WHITE = Color(1.0, 1.0, 1.0)


def _identity() -> List[List[float]]:
    return [[1.0 if i == j else 0.0 for j in range(4)] for i in range(4)]


def _matmul4(a, b) -> List[List[float]]:
    # Accumulate from 0.0 over k = 0..3 so composed matrices carry the same bits as
    # transformations.py:9-16 produces (the flattened scene must be bit-identical).
    out = []
    for i in range(4):
        row = []
        for j in range(4):
            acc = 0.0
            for k in range(4):
                acc += a[i][k] * b[k][j]
            row.append(acc)
        out.append(row)
    return out


class Transformation:
    """Affine transform with its stored inverse (``m`` and ``invm`` are 4x4 nested lists)."""

    def __init__(self, m=None, invm=None):
        self.m = m if m is not None else _identity()
        self.invm = invm if invm is not None else _identity()

    def __mul__(self, other: "Transformation") -> "Transformation":
        if not isinstance(other, Transformation):
            raise TypeError("host-side Transformation only composes with Transformation")
        # (A·B)⁻¹ = B⁻¹·A⁻¹  (transformations.py:87-92)
        return Transformation(_matmul4(self.m, other.m), _matmul4(other.invm, self.invm))

    def inverse(self) -> "Transformation":
        return Transformation(self.invm, self.m)


def _affine(rot3, tr3):
    m = _identity()
    for i in range(3):
        for j in range(3):
            m[i][j] = rot3[i][j]
        m[i][3] = tr3[i]
    return m


def translation(vec) -> Transformation:
    eye = [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]]
    return Transformation(_affine(eye, (vec.x, vec.y, vec.z)), _affine(eye, (-vec.x, -vec.y, -vec.z)))


def scaling(vec) -> Transformation:
    d = lambda a, b, c: [[a, 0.0, 0.0], [0.0, b, 0.0], [0.0, 0.0, c]]  # noqa: E731
    zero = (0.0, 0.0, 0.0)
    return Transformation(_affine(d(vec.x, vec.y, vec.z), zero),
                          _affine(d(1 / vec.x, 1 / vec.y, 1 / vec.z), zero))


def _rotation(axis: int, angle_deg: float) -> Transformation:
    s, c = math.sin(math.radians(angle_deg)), math.cos(math.radians(angle_deg))
    i, j = [(1, 2), (2, 0), (0, 1)][axis]  # the plane being rotated, right-handed
    fwd = [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]]
    bwd = [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]]
    fwd[i][i] = c; fwd[j][j] = c; fwd[i][j] = -s; fwd[j][i] = s  # noqa: E702
    bwd[i][i] = c; bwd[j][j] = c; bwd[i][j] = s; bwd[j][i] = -s  # noqa: E702
    zero = (0.0, 0.0, 0.0)
    return Transformation(_affine(fwd, zero), _affine(bwd, zero))


def rotation_x(angle_deg: float) -> Transformation:
    return _rotation(0, angle_deg)


def rotation_y(angle_deg: float) -> Transformation:
    return _rotation(1, angle_deg)


def rotation_z(angle_deg: float) -> Transformation:
    return _rotation(2, angle_deg)


# ---- pigments / BRDFs / materials: parameter holders only ---------------------------------------
class UniformPigment:
    def __init__(self, color: Color = BLACK):
        self.color = color


class CheckeredPigment:
    def __init__(self, color1: Color, color2: Color, num_of_steps=10):
        self.color1, self.color2, self.num_of_steps = color1, color2, num_of_steps


class ImagePigment:
    def __init__(self, image: "HdrImage"):
        self.image = image


class DiffuseBRDF:
    def __init__(self, pigment=None):
        self.pigment = pigment if pigment is not None else UniformPigment(WHITE)


class SpecularBRDF:
    def __init__(self, pigment=None, threshold_angle_rad=math.pi / 1800.0):
        self.pigment = pigment if pigment is not None else UniformPigment(WHITE)
        self.threshold_angle_rad = threshold_angle_rad


class Material:
    def __init__(self, brdf=None, emitted_radiance=None):
        self.brdf = brdf if brdf is not None else DiffuseBRDF()
        self.emitted_radiance = emitted_radiance if emitted_radiance is not None else UniformPigment(BLACK)


class Sphere:
    def __init__(self, transformation=None, material=None):
        self.transformation = transformation if transformation is not None else Transformation()
        self.material = material if material is not None else Material()


class Plane:
    def __init__(self, transformation=None, material=None):
        self.transformation = transformation if transformation is not None else Transformation()
        self.material = material if material is not None else Material()


@dataclass
class PointLight:
    position: Vec
    color: Color
    linear_radius: float = 0.0


class World:
    def __init__(self):
        self.shapes: list = []
        self.point_lights: list = []

    def add_shape(self, shape):
        self.shapes.append(shape)

    def add_light(self, light):
        self.point_lights.append(light)


class OrthogonalCamera:
    def __init__(self, aspect_ratio=1.0, transformation=None):
        self.aspect_ratio = aspect_ratio
        self.transformation = transformation if transformation is not None else Transformation()


class PerspectiveCamera:
    def __init__(self, screen_distance=1.0, aspect_ratio=1.0, transformation=None):
        self.screen_distance = screen_distance
        self.aspect_ratio = aspect_ratio
        self.transformation = transformation if transformation is not None else Transformation()


# ---- renderers: parameter holders; calling one is an error (the device computes radiance) ------
class _Renderer:
    def __init__(self, world: World, background_color: Color = BLACK):
        self.world = world
        self.background_color = background_color

    def __call__(self, ray):
        raise NotImplementedError(
            f"{type(self).__name__} is a parameter holder: radiance is evaluated by the HIP kernel "
            "through GpuImageTracer.fire_all_rays")


class OnOffRenderer(_Renderer):
    def __init__(self, world, background_color=BLACK, color=WHITE):
        super().__init__(world, background_color)
        self.color = color


class FlatRenderer(_Renderer):
    pass


class PathTracer(_Renderer):
    def __init__(self, world, background_color=BLACK, pcg=None, num_of_rays=10, max_depth=10,
                 russian_roulette_limit=3):
        super().__init__(world, background_color)
        self.pcg = pcg if pcg is not None else PCG()
        self.num_of_rays = num_of_rays
        self.max_depth = max_depth
        self.russian_roulette_limit = russian_roulette_limit


class PointLightRenderer(_Renderer):
    def __init__(self, world, background_color=BLACK, ambient_color=None):
        super().__init__(world, background_color)
        self.ambient_color = ambient_color if ambient_color is not None else Color(0.1, 0.1, 0.1)


# ---- PCG-XSH-RR 64/32 (pcg.py:23-62): host copy, used for seeds and synthetic scene recipes ------
_M64 = (1 << 64) - 1


class PCG:
    def __init__(self, init_state=42, init_seq=54):
        self.init_state, self.init_seq = init_state, init_seq  # remembered for the flattener
        self.state = 0
        self.inc = ((init_seq << 1) | 1) & _M64
        self.random()
        self.state = (self.state + init_state) & _M64
        self.random()

    def random(self) -> int:
        old = self.state
        self.state = (old * 6364136223846793005 + self.inc) & _M64
        xs = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xs >> rot) | (xs << ((-rot) & 31))) & 0xFFFFFFFF

    def random_float(self) -> float:
        return self.random() / 0xFFFFFFFF


def pcg_advance(state: int, inc: int, delta: int) -> int:
    """The state ``delta`` draws after ``state`` (the generator is a linear congruential one: jump by repeated squaring;
    identical to ``delta`` calls of ``random()``; csrc/pt_math.h: pcg_advance64)."""
    acc_mul, acc_add, cur_mul, cur_add = 1, 0, 6364136223846793005, inc & _M64
    while delta > 0:
        if delta & 1:
            acc_mul = (acc_mul * cur_mul) & _M64
            acc_add = (acc_add * cur_mul + cur_add) & _M64
        cur_add = ((cur_mul + 1) * cur_add) & _M64
        cur_mul = (cur_mul * cur_mul) & _M64
        delta >>= 1
    return (acc_mul * state + acc_add) & _M64


# ---- image container ---------------------------------------------------------------------------
class HdrImage:
    """Row-major RGB image, row 0 at the top (hdrimages.py:59-94).

    ``pixels`` is the reference's list of ``Color``; ``array`` is the same data as a
    ``[H, W, 3]`` float64 numpy array (filled by the device path, the list is built lazily)."""

    def __init__(self, width=0, height=0):
        self.width, self.height = width, height
        self.array = np.zeros((height, width, 3), dtype=np.float64)
        self._pixels = None

    @property
    def pixels(self):
        if self._pixels is None:
            flat = self.array.reshape(-1, 3).tolist()
            self._pixels = [Color(r, g, b) for r, g, b in flat]
        return self._pixels

    @pixels.setter
    def pixels(self, value):
        self._pixels = list(value)
        self.array = np.array([[c.r, c.g, c.b] for c in self._pixels], dtype=np.float64).reshape(
            self.height, self.width, 3)

    def set_array(self, arr):
        arr = np.asarray(arr, dtype=np.float64).reshape(self.height, self.width, 3)
        self.array = np.ascontiguousarray(arr)
        self._pixels = None

    def valid_coordinates(self, x, y):
        return 0 <= x < self.width and 0 <= y < self.height

    def pixel_offset(self, x, y):
        return y * self.width + x

    def get_pixel(self, x, y) -> Color:
        assert self.valid_coordinates(x, y)
        r, g, b = self.array[y, x]
        return Color(float(r), float(g), float(b))

    def set_pixel(self, x, y, new_color):
        assert self.valid_coordinates(x, y)
        self.array[y, x] = (new_color.r, new_color.g, new_color.b)
        self._pixels = None

    # -- post-processing (hdrimages.py:96-171): upload, run the device kernels, download -----------------
    def _device(self):
        from .postprocess import DeviceImage

        return DeviceImage.from_numpy(self.array)

    def write_pfm(self, stream, endianness=1):
        self._device().write_pfm(stream, endianness)

    def average_luminosity(self, delta=1e-10):
        return self._device().average_luminosity(delta)

    def normalize_image(self, factor, luminosity=None):
        d = self._device()
        d.normalize_image(factor, luminosity)
        self.set_array(d.numpy())

    def clamp_image(self):
        d = self._device()
        d.clamp_image()
        self.set_array(d.numpy())

    def write_ldr_image(self, stream, format, gamma=1.0):
        self._device().write_ldr_image(stream, format, gamma)
