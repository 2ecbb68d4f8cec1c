"""``GpuImageTracer`` with a ``func`` that is not a renderer: the cases of the reference's
``TestImageTracer`` (test_all.py:556-604) pointed at the drop-in, first with this repository's parameter-holder
classes, then -- where the reference is importable (the build container) -- with the reference's own ``HdrImage``,
cameras, ``PCG`` and ``Color``, pixel for pixel against the reference's ``ImageTracer``.  No GPU, no oracle."""
import os
import sys

import pytest

from pytracer_amd import hostmodel as hm
from pytracer_amd.tracer import GpuImageTracer


def _close(a, b, eps=1e-5):
    return abs(a - b) < eps


def test_orientation_and_uv_sub_mapping():
    # test_all.py:562-574
    tracer = GpuImageTracer(image=hm.HdrImage(4, 2), camera=hm.PerspectiveCamera(aspect_ratio=2))
    p = tracer.fire_ray(0, 0, u_pixel=0.0, v_pixel=0.0).at(1.0)
    assert _close(p.x, 0.0) and _close(p.y, 2.0) and _close(p.z, 1.0)
    p = tracer.fire_ray(3, 1, u_pixel=1.0, v_pixel=1.0).at(1.0)
    assert _close(p.x, 0.0) and _close(p.y, -2.0) and _close(p.z, -1.0)
    r1 = tracer.fire_ray(0, 0, u_pixel=2.5, v_pixel=1.5)
    r2 = tracer.fire_ray(2, 1, u_pixel=0.5, v_pixel=0.5)
    for a, b in ((r1.origin, r2.origin), (r1.dir, r2.dir)):
        assert _close(a.x, b.x) and _close(a.y, b.y) and _close(a.z, b.z)


def test_image_coverage_with_a_lambda():
    # test_all.py:576-580
    image = hm.HdrImage(4, 2)
    GpuImageTracer(image=image, camera=hm.PerspectiveCamera(aspect_ratio=2)).fire_all_rays(
        lambda ray: hm.Color(1.0, 2.0, 3.0))
    for row in range(image.height):
        for col in range(image.width):
            assert image.get_pixel(col, row) == hm.Color(1.0, 2.0, 3.0)


def test_antialiasing_count_and_bounds():
    # test_all.py:582-604
    num_of_rays = 0
    small_image = hm.HdrImage(1, 1)
    tracer = GpuImageTracer(small_image, hm.OrthogonalCamera(aspect_ratio=1), samples_per_side=10, pcg=hm.PCG())

    def trace_ray(ray):
        nonlocal num_of_rays
        point = ray.at(1)
        assert pytest.approx(0.0) == point.x
        assert -1.0 <= point.y <= 1.0
        assert -1.0 <= point.z <= 1.0
        num_of_rays += 1
        return hm.Color(0.0, 0.0, 0.0)

    tracer.fire_all_rays(trace_ray)
    assert num_of_rays == 100
    assert small_image.get_pixel(0, 0) == hm.Color(0.0, 0.0, 0.0)


def test_jitter_order_mean_and_callback():
    """u is drawn before v, sub-row outer / sub-column inner, the pixel is the sum times 1/S**2
    (imagetracer.py:86-101); the callback is called once before the loop and then only after
    ``callback_time_s`` (imagetracer.py:76-78, 106-110)."""
    S, W, H = 3, 2, 2
    image = hm.HdrImage(W, H)
    tracer = GpuImageTracer(image, hm.OrthogonalCamera(aspect_ratio=1.0), samples_per_side=S, pcg=hm.PCG(7, 9))
    seen = []

    def func(ray):  # the ray's origin encodes (u, v): o = (-1, (1 - 2u) * a, 2v - 1)
        seen.append((ray.origin.y, ray.origin.z))
        return hm.Color(ray.origin.y, ray.origin.z, 1.0)

    calls = []
    tracer.fire_all_rays(func, callback=lambda col, row, tag: calls.append((col, row, tag)), tag="x")
    assert calls == [(0, 0, "x")]
    g = hm.PCG(7, 9)
    k = 0
    for row in range(H):
        for col in range(W):
            acc = [0.0, 0.0, 0.0]
            for sr in range(S):
                for sc in range(S):
                    up = (sc + g.random_float()) / S
                    vp = (sr + g.random_float()) / S
                    u, v = (col + up) / W, 1.0 - (row + vp) / H
                    assert seen[k] == ((1.0 - 2 * u) * 1.0, 2 * v - 1)
                    acc = [acc[0] + seen[k][0], acc[1] + seen[k][1], acc[2] + 1.0]
                    k += 1
            want = hm.Color(*(a * (1 / S ** 2) for a in acc))
            assert image.get_pixel(col, row) == want
    # a callback_time_s of zero reports (nearly) every pixel, in raster order
    calls.clear()
    GpuImageTracer(hm.HdrImage(3, 2), hm.OrthogonalCamera()).fire_all_rays(
        lambda ray: hm.Color(sum(i * i for i in range(20000)) * 0.0, 0.0, 0.0),
        callback=lambda col, row: calls.append((col, row)), callback_time_s=0.0)
    assert calls[0] == (0, 0) and len(calls) >= 2 and calls[1:] == sorted(calls[1:], key=lambda c: (c[1], c[0]))


def test_not_callable_is_a_type_error():
    with pytest.raises(TypeError):
        GpuImageTracer(hm.HdrImage(1, 1), hm.OrthogonalCamera()).fire_all_rays(42)


def test_renderer_never_falls_back_to_the_host_loop():
    """A renderer object is never treated as an opaque callable: an unsupported world is an error."""
    from pytracer_amd import flatten

    class Torus:
        transformation = hm.Transformation()
        material = hm.Material()

    w = hm.World()
    w.add_shape(Torus())
    with pytest.raises(flatten.UnsupportedSceneError):
        GpuImageTracer(hm.HdrImage(2, 2), hm.OrthogonalCamera()).fire_all_rays(hm.FlatRenderer(w))


# ---- against the reference's own classes (build container only; the reference never travels) --------------------
REF_SRC = "/root/reference/src"


@pytest.fixture()
def ref():
    if not os.path.isdir(os.path.join(REF_SRC, "pytracer")):
        pytest.skip("the reference is not present here")
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF_SRC)
    try:
        import pytracer.camera
        import pytracer.colors
        import pytracer.hdrimages
        import pytracer.imagetracer
        import pytracer.pcg
        import pytracer.transformations  # noqa: F401
        yield sys.modules["pytracer"]
    finally:
        sys.path.remove(REF_SRC)
        for name in [m for m in sys.modules if m == "pytracer" or m.startswith("pytracer.")]:
            del sys.modules[name]


@pytest.mark.parametrize("S", [0, 2])
def test_host_loop_equals_reference_imagetracer(ref, S):
    from pytracer.camera import OrthogonalCamera, PerspectiveCamera
    from pytracer.colors import Color
    from pytracer.geometry import Vec
    from pytracer.hdrimages import HdrImage
    from pytracer.imagetracer import ImageTracer
    from pytracer.pcg import PCG
    from pytracer.transformations import rotation_z, translation

    for camera in (PerspectiveCamera(aspect_ratio=1.5, transformation=rotation_z(20.0) * translation(Vec(-1.0, 0.5, 0.2))),
                   OrthogonalCamera(aspect_ratio=1.5)):
        def func(ray):  # any function of the ray: the two tracers must call it with the same rays in the same order
            p = ray.at(1.25)
            return Color(p.x * 0.5 + 1.0, abs(p.y), p.z * p.z)

        a, b = HdrImage(6, 4), HdrImage(6, 4)
        ImageTracer(a, camera, samples_per_side=S, pcg=PCG(11, 3)).fire_all_rays(func)
        calls = []
        GpuImageTracer(b, camera, samples_per_side=S, pcg=PCG(11, 3)).fire_all_rays(
            func, callback=lambda col, row: calls.append((col, row)))
        assert calls == [(0, 0)]
        assert all(type(c) is Color for c in b.pixels)
        assert [(c.r, c.g, c.b) for c in a.pixels] == [(c.r, c.g, c.b) for c in b.pixels]


def test_renderer_with_an_inexpressible_world_raises_unless_the_host_fallback_is_asked_for(ref):
    """the reference runs any renderer on any world (imagetracer.py:60-110, render.py:26-39).  A world
    with a shape class the device cannot express raises ``UnsupportedSceneError`` by default -- nothing leaves the device
    path silently -- and with ``fallback="host"`` the renderer, itself a callable Ray -> Color, goes through the host loop:
    the reference's own code computes every radiance, and the frame equals the reference ImageTracer's."""
    from pytracer.camera import PerspectiveCamera
    from pytracer.colors import Color
    from pytracer.geometry import Vec
    from pytracer.hdrimages import HdrImage
    from pytracer.imagetracer import ImageTracer
    from pytracer.materials import DiffuseBRDF, Material, UniformPigment
    from pytracer.pcg import PCG
    from pytracer.render import FlatRenderer, PathTracer
    from pytracer.shapes import Sphere
    from pytracer.transformations import scaling, translation
    from pytracer.world import World

    from pytracer_amd import flatten

    class Blob(Sphere):  # a shape class of the user's own: the flattener knows shapes by class name
        pass

    world = World()
    world.add_shape(Blob(transformation=translation(Vec(2.0, 0.0, 0.0)) * scaling(Vec(0.7, 0.7, 0.7)),
                         material=Material(brdf=DiffuseBRDF(UniformPigment(Color(0.3, 0.6, 0.9))),
                                           emitted_radiance=UniformPigment(Color(0.1, 0.2, 0.3)))))
    camera = PerspectiveCamera(aspect_ratio=1.5)
    for make, S in ((lambda: FlatRenderer(world, background_color=Color(0.0, 0.1, 0.0)), 0),
                    (lambda: PathTracer(world, pcg=PCG(45, 54), num_of_rays=2, max_depth=2), 2)):
        with pytest.raises(flatten.UnsupportedSceneError):
            GpuImageTracer(HdrImage(6, 4), camera, samples_per_side=S, pcg=PCG(11, 3)).fire_all_rays(make())
        a, b = HdrImage(6, 4), HdrImage(6, 4)
        ImageTracer(a, camera, samples_per_side=S, pcg=PCG(11, 3)).fire_all_rays(make())
        t = GpuImageTracer(b, camera, samples_per_side=S, pcg=PCG(11, 3), fallback="host")
        t.fire_all_rays(make())
        assert t.last_path == "host"
        assert [(c.r, c.g, c.b) for c in a.pixels] == [(c.r, c.g, c.b) for c in b.pixels]
    with pytest.raises(ValueError):
        GpuImageTracer(HdrImage(2, 2), camera, fallback="oracle")
