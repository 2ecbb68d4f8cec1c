"""Scene recipes: the synthetic benchmark worlds of SURVEY.md §8(d) and the demo scene.

All randomness comes from the bit-exact PCG restatement (no numpy RNG), drawn in a fixed order,
so the same scene is produced everywhere (build container, GPU box, any rank).
"""
from __future__ import annotations

from typing import Tuple

from .hostmodel import (PCG, CheckeredPigment, Color, DiffuseBRDF, Material, PerspectiveCamera, Plane,
                        PointLight, SpecularBRDF, Sphere, Transformation, UniformPigment, Vec, World,
                        rotation_y, rotation_z, scaling, translation)

BLACK = Color(0.0, 0.0, 0.0)


def synthetic_world(n_spheres: int = 32, with_plane: bool = False, wide: bool = False) -> World:
    """sky sphere + (n_spheres-1) random spheres (+ checkered ground plane).

    ``wide=True`` is the C4/C5 variant (smaller spheres spread over a larger area)."""
    g = PCG(42, 54)
    r = g.random_float
    world = World()
    sky = Material(brdf=DiffuseBRDF(UniformPigment(BLACK)),
                   emitted_radiance=UniformPigment(Color(0.7, 0.5, 1.0)))
    world.add_shape(Sphere(scaling(Vec(50.0, 50.0, 50.0)), sky))
    for _ in range(1, n_spheres):
        if wide:
            rad = 0.02 + 0.08 * r()
            cx = 1 + 30 * r()
            cy = -15 + 30 * r()
            cz = rad + 6 * r()
        else:
            rad = 0.1 + 0.4 * r()
            cx = 1 + 9 * r()
            cy = -5 + 10 * r()
            cz = rad + 2 * r()
        colour = Color(0.1 + 0.8 * r(), 0.1 + 0.8 * r(), 0.1 + 0.8 * r())
        brdf = SpecularBRDF(UniformPigment(colour)) if r() < 0.2 else DiffuseBRDF(UniformPigment(colour))
        world.add_shape(Sphere(translation(Vec(cx, cy, cz)) * scaling(Vec(rad, rad, rad)),
                               Material(brdf=brdf, emitted_radiance=UniformPigment(BLACK))))
    if with_plane:
        ground = Material(
            brdf=DiffuseBRDF(CheckeredPigment(Color(0.3, 0.5, 0.1), Color(0.1, 0.2, 0.5), 4)),
            emitted_radiance=UniformPigment(BLACK))
        world.add_shape(Plane(material=ground))
    return world


def synthetic_camera(width: int, height: int) -> PerspectiveCamera:
    return PerspectiveCamera(screen_distance=1.0, aspect_ratio=width / height,
                             transformation=translation(Vec(-1.0, 0.0, 1.0)))


def _as_parsed(*factors) -> Transformation:
    """Compose factors the way the reference's scene parser does (scene_file.py:505-560): it starts
    from the identity and multiplies every factor in, so even a single ``translation(...)`` has gone
    through one matrix product (which e.g. turns the -0.0 entries of an inverse into +0.0)."""
    result = Transformation()
    for f in factors:
        result = result * f
    return result


def demo_world(clock: float = 150.0) -> Tuple[World, PerspectiveCamera]:
    """The scene described by the reference's examples/demo.txt:1-28 (pure data: two planes, one
    sphere, three materials, one point light, a perspective camera)."""
    sky = Material(DiffuseBRDF(UniformPigment(BLACK)), UniformPigment(Color(0.7, 0.5, 1.0)))
    ground = Material(DiffuseBRDF(CheckeredPigment(Color(0.3, 0.5, 0.1), Color(0.1, 0.2, 0.5), 4)),
                      UniformPigment(BLACK))
    mirror = Material(SpecularBRDF(UniformPigment(Color(0.5, 0.5, 0.5))), UniformPigment(BLACK))
    world = World()
    world.add_light(PointLight(Vec(10.0, 10.0, 10.0), Color(1.0, 1.0, 1.0), 1.0))
    world.add_shape(Plane(_as_parsed(translation(Vec(0.0, 0.0, 100.0)), rotation_y(clock)), sky))
    world.add_shape(Plane(_as_parsed(), ground))
    world.add_shape(Sphere(_as_parsed(translation(Vec(0.0, 0.0, 1.0))), mirror))
    camera = PerspectiveCamera(screen_distance=1.0, aspect_ratio=1.0,
                               transformation=_as_parsed(rotation_z(30.0), translation(Vec(-4.0, 0.0, 1.0))))
    return world, camera
