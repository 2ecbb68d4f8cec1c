"""Duck-typed flattening of pytracer objects into the C-ABI's structure-of-arrays scene.

Works on the reference's own objects (``pytracer.world.World`` …) and on the stand-ins of
:mod:`pytracer_amd.hostmodel`: objects are recognised by class name and read by attribute name.
Anything the device path cannot represent raises ``UnsupportedSceneError`` — there is no silent
CPU fallback.

Reads: ``World.shapes`` (world.py:38-45, order significant — ties go to the first shape,
world.py:62), ``Shape.transformation.m/.invm`` (transformations.py:48-56),
``Material.brdf/.emitted_radiance`` (materials.py:199-204), pigments (materials.py:50-100),
``World.point_lights`` (lights.py:25-39), cameras (camera.py:48-57, 87-101), renderers
(render.py:26-97, 142-156).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from . import abi


class UnsupportedSceneError(TypeError):
    """The object cannot be expressed in the device path's scene description."""


def _cls(obj) -> str:
    return type(obj).__name__


def _affine12(matrix, what: str) -> list:
    """Rows 0..2 of a 4x4 nested list; row 3 must be exactly (0, 0, 0, 1)."""
    rows = [list(map(float, r)) for r in matrix]
    if len(rows) != 4 or any(len(r) != 4 for r in rows):
        raise UnsupportedSceneError(f"{what}: expected a 4x4 matrix")
    if rows[3] != [0.0, 0.0, 0.0, 1.0]:
        raise UnsupportedSceneError(
            f"{what}: non-affine matrix (row 3 = {rows[3]}); the reference would divide by w "
            "(transformations.py:73-78), the device path does not")
    return rows[0] + rows[1] + rows[2]


def _rgb(c) -> Tuple[float, float, float]:
    return float(c.r), float(c.g), float(c.b)


class _Textures:
    def __init__(self):
        self.images, self.w, self.h, self.offset, self.data = [], [], [], [], []
        self._next = 0

    def add(self, image) -> int:
        for i, known in enumerate(self.images):
            if known is image:
                return i
        w, h = int(image.width), int(image.height)
        if hasattr(image, "array"):
            px = np.asarray(image.array, dtype=np.float64).reshape(h * w, 3)
        else:
            px = np.array([[c.r, c.g, c.b] for c in image.pixels], dtype=np.float64).reshape(h * w, 3)
        self.images.append(image)
        self.w.append(w)
        self.h.append(h)
        self.offset.append(self._next)
        self.data.append(px.reshape(-1))
        self._next += px.size
        return len(self.images) - 1


def _pigment(p, textures: _Textures):
    """-> (kind, c1, c2, steps, tex)"""
    name = _cls(p)
    if name == "UniformPigment":
        return abi.PIGMENT_UNIFORM, _rgb(p.color), (0.0, 0.0, 0.0), 0.0, -1
    if name == "CheckeredPigment":
        return abi.PIGMENT_CHECKERED, _rgb(p.color1), _rgb(p.color2), float(p.num_of_steps), -1
    if name == "ImagePigment":
        return abi.PIGMENT_IMAGE, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), 0.0, textures.add(p.image)
    raise UnsupportedSceneError(f"unknown pigment class {name!r}")


def flatten_world(world) -> abi.FlatScene:
    """``World`` -> :class:`abi.FlatScene` (shape order preserved)."""
    shapes = list(world.shapes)
    n = len(shapes)
    kind = np.zeros(n, np.int32)
    invm = np.zeros((12, n))
    m = np.zeros((12, n))
    brdf_kind = np.zeros(n, np.int32)
    brdf_param = np.zeros(n)
    pig = dict(kind=np.zeros(n, np.int32), c1=np.zeros((3, n)), c2=np.zeros((3, n)),
               steps=np.zeros(n), tex=np.full(n, -1, np.int32))
    emi = dict(kind=np.zeros(n, np.int32), c1=np.zeros((3, n)), c2=np.zeros((3, n)),
               steps=np.zeros(n), tex=np.full(n, -1, np.int32))
    textures = _Textures()
    for i, s in enumerate(shapes):
        name = _cls(s)
        if name == "Sphere":
            kind[i] = abi.SHAPE_SPHERE
        elif name == "Plane":
            kind[i] = abi.SHAPE_PLANE
        else:
            raise UnsupportedSceneError(f"shape {i}: unknown shape class {name!r}")
        m[:, i] = _affine12(s.transformation.m, f"shape {i} transformation.m")
        invm[:, i] = _affine12(s.transformation.invm, f"shape {i} transformation.invm")
        brdf = s.material.brdf
        bname = _cls(brdf)
        if bname == "DiffuseBRDF":
            brdf_kind[i] = abi.BRDF_DIFFUSE
        elif bname == "SpecularBRDF":
            brdf_kind[i] = abi.BRDF_SPECULAR
            brdf_param[i] = float(brdf.threshold_angle_rad)
        else:
            raise UnsupportedSceneError(f"shape {i}: unknown BRDF class {bname!r}")
        for dst, src in ((pig, brdf.pigment), (emi, s.material.emitted_radiance)):
            k, c1, c2, steps, tex = _pigment(src, textures)
            dst["kind"][i], dst["steps"][i], dst["tex"][i] = k, steps, tex
            dst["c1"][:, i], dst["c2"][:, i] = c1, c2
    lights = list(getattr(world, "point_lights", []))
    nl = len(lights)
    lpos, lcol, lrad = np.zeros((3, nl)), np.zeros((3, nl)), np.zeros(nl)
    for j, lt in enumerate(lights):
        lpos[:, j] = (lt.position.x, lt.position.y, lt.position.z)
        lcol[:, j] = _rgb(lt.color)
        lrad[j] = float(lt.linear_radius)
    return abi.FlatScene(
        kind=kind, invm=invm, m=m, brdf_kind=brdf_kind, brdf_param=brdf_param,
        pig_kind=pig["kind"], pig_c1=pig["c1"], pig_c2=pig["c2"], pig_steps=pig["steps"],
        pig_tex=pig["tex"], emi_kind=emi["kind"], emi_c1=emi["c1"], emi_c2=emi["c2"],
        emi_steps=emi["steps"], emi_tex=emi["tex"], light_pos=lpos, light_color=lcol,
        light_radius=lrad, tex_w=np.array(textures.w, np.int32), tex_h=np.array(textures.h, np.int32),
        tex_offset=np.array(textures.offset, np.int64),
        tex_data=np.concatenate(textures.data) if textures.data else np.zeros(0))


def flatten_camera(camera) -> abi.Camera:
    name = _cls(camera)
    m12 = _affine12(camera.transformation.m, "camera transformation.m")
    if name == "PerspectiveCamera":
        return abi.make_camera(abi.CAMERA_PERSPECTIVE, m12, camera.screen_distance, camera.aspect_ratio)
    if name == "OrthogonalCamera":
        return abi.make_camera(abi.CAMERA_ORTHOGONAL, m12, 1.0, camera.aspect_ratio)
    raise UnsupportedSceneError(f"unknown camera class {name!r}")


_MULT = 6364136223846793005
_MULT_INV = pow(_MULT, -1, 1 << 64)
_M64 = (1 << 64) - 1


def recover_seeds(pcg, constructed: bool = False) -> Tuple[int, int]:
    """(init_state, init_seq) such that ``PCG(init_state, init_seq)`` is in the state ``pcg`` is in NOW: the seeds it was
    built with if it has not been drawn from since, otherwise the seeds of the generator that starts where this one has
    got to (a tracer whose ``pcg`` an earlier frame already drew from continues its stream, as the reference does).

    ``constructed=True`` (the per-pixel and per-sample alignments, whose generators are DERIVED from a seed pair and do not
    continue anybody's stream): a generator that remembers what it was built with (``init_state`` / ``init_seq``
    attributes, as :class:`pytracer_amd.hostmodel.PCG` does) gives those, drawn from or not -- round 3's behaviour, which
    ``pcg_mode="pixel"`` keeps reproducing; the reference's own PCG remembers nothing and is solved for.

    The reference's PCG keeps only (state, inc) (pcg.py:25-41).  For a fresh generator
    state = ((inc + init_state) * MULT + inc) mod 2^64 and inc = (init_seq << 1) | 1, so both
    seeds can be solved for exactly (seeds < 2^63)."""
    if constructed and hasattr(pcg, "init_state") and hasattr(pcg, "init_seq"):
        return int(pcg.init_state), int(pcg.init_seq)
    inc = int(pcg.inc) & _M64
    state = int(pcg.state) & _M64
    init_state = ((((state - inc) & _M64) * _MULT_INV) - inc) & _M64
    return init_state, inc >> 1


RENDERER_KINDS = {
    "OnOffRenderer": abi.RENDERER_ONOFF,
    "FlatRenderer": abi.RENDERER_FLAT,
    "PathTracer": abi.RENDERER_PATHTRACER,
    "PointLightRenderer": abi.RENDERER_POINTLIGHT,
}


def is_device_renderer(func) -> bool:
    """True for the reference's renderer objects (by class name, render.py:42-193): what the device path
    renders.  Anything else handed to ``fire_all_rays`` is an opaque callable."""
    return _cls(func) in RENDERER_KINDS and hasattr(func, "world") and hasattr(func, "background_color")


def renderer_kind(renderer) -> int:
    """``PT_RENDERER_*`` of a renderer object (by class name)."""
    return RENDERER_KINDS[_cls(renderer)]


def renderer_params(renderer, width: int, height: int, samples_per_side: int = 0,
                    tracer_pcg=None, pcg_mode: int = abi.PCG_PIXEL,
                    out_format: int = abi.OUT_F64) -> abi.Params:
    """Renderer (+ ImageTracer settings) -> ``pt_params`` (single-rank partition)."""
    name = _cls(renderer)
    if name not in RENDERER_KINDS:
        raise UnsupportedSceneError(
            f"{name!r} is not a renderer the device path implements "
            f"(expected one of {sorted(RENDERER_KINDS)}); arbitrary callables cannot run on the GPU")
    kw = dict(background=_rgb(renderer.background_color))
    derived = pcg_mode != abi.PCG_SEQ  # (per-pixel / per-sample generators are derived from the seeds a PCG was built with)
    j_state, j_seq = recover_seeds(tracer_pcg, derived) if tracer_pcg is not None else (42, 54)
    p_state, p_seq = j_state, j_seq
    if name == "OnOffRenderer":
        kw["onoff_color"] = _rgb(renderer.color)
    elif name == "PathTracer":
        kw.update(num_of_rays=renderer.num_of_rays, max_depth=renderer.max_depth,
                  rr_limit=renderer.russian_roulette_limit)
        p_state, p_seq = recover_seeds(renderer.pcg, derived)
    elif name == "PointLightRenderer":
        kw["ambient"] = _rgb(renderer.ambient_color)
    return abi.make_params(width, height, RENDERER_KINDS[name], samples_per_side=samples_per_side,
                           pcg_mode=pcg_mode, jitter_state=j_state, jitter_seq=j_seq,
                           path_state=p_state, path_seq=p_seq, out_format=out_format, **kw)
