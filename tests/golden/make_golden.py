#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE itself.

Run in the build container only (the reference lives at /root/reference and never travels):

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden.py [--out DIR] [g1 g2 ... g5cli]

Without generator names every generator runs IN ITS OWN PROCESS: the reference's scene parser keeps one default
``World()`` for all parses of a process (scene_file.py:363), so a second ``parse_scene`` in the same
process would see the first one's shapes again -- one parse per process is the recipe that produced the committed
files.  ``--out DIR`` writes somewhere else (tests/test_golden_regen.py regenerates a fixture and compares arrays).

Each fixture is pure data: flattened inputs (through pytracer_amd.flatten's duck-typed reader,
applied to the reference's own objects) and the outputs the reference computed for them, stored
as fp64 / integer numpy arrays in .npz files.  Nothing of the reference's source is stored.
"""
import math
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")

import numpy as np  # noqa: E402

from pytracer.camera import OrthogonalCamera, PerspectiveCamera  # noqa: E402
from pytracer.colors import BLACK, WHITE, Color  # noqa: E402
from pytracer.geometry import Normal, Point, Vec, create_onb_from_z  # noqa: E402
from pytracer.hdrimages import HdrImage  # noqa: E402
from pytracer.imagetracer import ImageTracer  # noqa: E402
from pytracer.lights import PointLight  # noqa: E402
from pytracer.materials import (CheckeredPigment, DiffuseBRDF, ImagePigment, Material,  # noqa: E402
                                SpecularBRDF, UniformPigment)
from pytracer.pcg import PCG  # noqa: E402
from pytracer.ray import Ray  # noqa: E402
from pytracer.render import FlatRenderer, OnOffRenderer, PathTracer, PointLightRenderer  # noqa: E402
from pytracer.scene_file import InputStream, parse_scene  # noqa: E402
from pytracer.shapes import Plane, Sphere  # noqa: E402
from pytracer.transformations import (Transformation, rotation_x, rotation_y, rotation_z,  # noqa: E402
                                      scaling, translation)
from pytracer.world import World  # noqa: E402

from pytracer_amd import abi, flatten  # noqa: E402

INF = float("inf")


OUT_DIR = HERE  # (--out)


def save(name, **arrays):
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def pixels_array(image):
    return np.array([[c.r, c.g, c.b] for c in image.pixels], dtype=np.float64).reshape(
        image.height, image.width, 3)


def params_dict(p: abi.Params):
    d = {}
    for name, _ in abi.Params._fields_:
        v = getattr(p, name)
        d["par_" + name] = np.array(list(v)) if hasattr(v, "__len__") else np.array(v)
    return d


# ---------------------------------------------------------------------------------------------
def ref_synthetic_world(n_spheres=32, with_plane=False, wide=False):
    """SURVEY.md §8(d) recipe, built from the reference's classes."""
    g = PCG(42, 54)
    r = g.random_float
    world = World()
    sky = Material(brdf=DiffuseBRDF(UniformPigment(BLACK)),
                   emitted_radiance=UniformPigment(Color(0.7, 0.5, 1.0)))
    world.add_shape(Sphere(scaling(Vec(50.0, 50.0, 50.0)), sky))
    for _ in range(1, n_spheres):
        if wide:
            rad = 0.02 + 0.08 * r(); cx = 1 + 30 * r(); cy = -15 + 30 * r(); cz = rad + 6 * r()  # noqa: E702
        else:
            rad = 0.1 + 0.4 * r(); cx = 1 + 9 * r(); cy = -5 + 10 * r(); cz = rad + 2 * r()  # noqa: E702
        colour = Color(0.1 + 0.8 * r(), 0.1 + 0.8 * r(), 0.1 + 0.8 * r())
        brdf = SpecularBRDF(UniformPigment(colour)) if r() < 0.2 else DiffuseBRDF(UniformPigment(colour))
        world.add_shape(Sphere(translation(Vec(cx, cy, cz)) * scaling(Vec(rad, rad, rad)),
                               Material(brdf=brdf, emitted_radiance=UniformPigment(BLACK))))
    if with_plane:
        world.add_shape(Plane(material=Material(
            brdf=DiffuseBRDF(CheckeredPigment(Color(0.3, 0.5, 0.1), Color(0.1, 0.2, 0.5), 4)),
            emitted_radiance=UniformPigment(BLACK))))
    return world


def ref_synthetic_camera(w, h):
    return PerspectiveCamera(screen_distance=1.0, aspect_ratio=w / h,
                             transformation=translation(Vec(-1.0, 0.0, 1.0)))


class Seeder:
    """Drives the reference's verbatim fire_all_rays into Mode PIXEL / SAMPLE."""

    def __init__(self, renderer, S, s0, q0, per_sample=False):
        self.renderer, self.S, self.s0, self.q0 = renderer, S, s0, q0
        self.period = 2 if per_sample else 2 * S * S
        self.draws = 0
        self.index = 0
        self.cur = None

    def reseed(self):
        self.cur = PCG(self.s0, self.q0 + self.index)
        self.index += 1
        if hasattr(self.renderer, "pcg"):
            self.renderer.pcg = self.cur

    def random_float(self):  # stands in for ImageTracer.pcg
        if self.draws % self.period == 0:
            self.reseed()
        self.draws += 1
        return self.cur.random_float()

    def func(self, ray):  # S == 0: one call per pixel
        self.reseed()
        return self.renderer(ray)


def render_ref(world, camera, renderer, w, h, S=0, mode=abi.PCG_SEQ, s0=42, q0=54, jitter=(42, 54)):
    image = HdrImage(w, h)
    if mode == abi.PCG_SEQ:
        tracer = ImageTracer(image, camera, samples_per_side=S, pcg=PCG(*jitter))
        tracer.fire_all_rays(renderer)
    else:
        seeder = Seeder(renderer, S, s0, q0, per_sample=(mode == abi.PCG_SAMPLE))
        tracer = ImageTracer(image, camera, samples_per_side=S, pcg=seeder)
        tracer.fire_all_rays(renderer if S > 0 else seeder.func)
    return pixels_array(image)


def frame_fixture(name, world, camera, make_renderer, w, h, S=0, mode=abi.PCG_SEQ, s0=42, q0=54,
                  jitter=(42, 54)):
    scene = flatten.flatten_world(world)
    cam = flatten.flatten_camera(camera)
    renderer = make_renderer()
    tracer_pcg = PCG(*jitter)
    par = flatten.renderer_params(renderer, w, h, samples_per_side=S, tracer_pcg=tracer_pcg, pcg_mode=mode)
    if mode != abi.PCG_SEQ:
        par.path_state, par.path_seq = s0, q0
    px = render_ref(world, camera, make_renderer(), w, h, S, mode, s0, q0, jitter)
    save(name, pixels=px, checksum=np.array(float(sum(c for c in px.reshape(-1).tolist()))),
         **scene.to_dict(), **abi.camera_to_dict(cam), **params_dict(par))
    return px


# ---------------------------------------------------------------------------------------------
def g1_pcg():
    seeds = [(42, 54), (45, 54), (45, 54 + 921599), (0, 0), (123456789012345, 2**62 + 17)]
    st, inc, outs, floats = [], [], [], []
    for s, q in seeds:
        p = PCG(s, q)
        st.append(p.state)
        inc.append(p.inc)
        outs.append([p.random() for _ in range(16)])
        p2 = PCG(s, q)
        floats.append([p2.random_float() for _ in range(16)])
    save("g1_pcg", seeds=np.array(seeds, dtype=np.uint64), state=np.array(st, dtype=np.uint64),
         inc=np.array(inc, dtype=np.uint64), outputs=np.array(outs, dtype=np.uint32),
         floats=np.array(floats, dtype=np.float64))


def rnd_transform(r):
    t = translation(Vec(4 * r() - 2, 4 * r() - 2, 4 * r() - 2))
    rot = rotation_x(360 * r()) * rotation_y(360 * r()) * rotation_z(360 * r())
    sc = scaling(Vec(0.2 + 2 * r(), 0.2 + 2 * r(), 0.2 + 2 * r()))
    return t * rot * sc


def m12(mat):
    return [x for row in mat[:3] for x in row]


def g2_xform():
    g = PCG(7, 11)
    r = g.random_float
    ms, invs, vin, pt, vc, nm = [], [], [], [], [], []
    for _ in range(64):
        T = rnd_transform(r)
        v = (10 * r() - 5, 10 * r() - 5, 10 * r() - 5)
        ms.append(m12(T.m))
        invs.append(m12(T.invm))
        vin.append(v)
        p = T * Point(*v)
        w = T * Vec(*v)
        n = T * Normal(*v)
        pt.append((p.x, p.y, p.z))
        vc.append((w.x, w.y, w.z))
        nm.append((n.x, n.y, n.z))
    save("g2_xform", m=np.array(ms), invm=np.array(invs), vin=np.array(vin), point=np.array(pt),
         vec=np.array(vc), normal=np.array(nm))


def g3_shapes():
    g = PCG(3, 5)
    r = g.random_float
    world = World()
    mat = Material()
    for i in range(12):
        T = rnd_transform(r)
        world.add_shape(Sphere(T, mat) if i % 3 else Plane(T, mat))
    # axis-aligned / untransformed shapes as in the reference's own tests
    world.add_shape(Sphere(material=mat))
    world.add_shape(Plane(material=mat))
    world.add_shape(Sphere(translation(Vec(10.0, 0.0, 0.0)), mat))
    scene = flatten.flatten_world(world)
    rays, per_shape, per_world, quick = [], [], [], []
    n = len(world.shapes)
    fixed = [
        ((0, 0, 2), (0, 0, -1)), ((3, 0, 0), (-1, 0, 0)), ((0, 0, 0), (1, 0, 0)),  # test_all.py:608-650
        ((0, 0, 1), (0, 0, -1)), ((0, 0, 1), (0, 0, 1)), ((0, 0, 1), (1, 0, 0)),  # test_all.py:752-770
        ((10, 0, 2), (0, 0, -1)), ((13, 0, 0), (-1, 0, 0)),
    ]
    for k in range(400):
        if k < len(fixed):
            o, d = fixed[k]
            o, d = tuple(map(float, o)), tuple(map(float, d))
        else:
            o = (8 * r() - 4, 8 * r() - 4, 8 * r() - 4)
            d = (2 * r() - 1, 2 * r() - 1, 2 * r() - 1)
        tmin = 1e-5 if k % 2 == 0 else 1e-3
        tmax = INF if k % 5 else 3.0
        ray = Ray(origin=Point(*o), dir=Vec(*d), tmin=tmin, tmax=tmax)
        rays.append(list(o) + list(d) + [tmin, tmax])
        row, qrow = [], []
        for s in world.shapes:
            h = s.ray_intersection(ray)
            if h is None:
                row.append([0.0] + [0.0] * 9)
            else:
                row.append([1.0, h.t, h.world_point.x, h.world_point.y, h.world_point.z, h.normal.x,
                            h.normal.y, h.normal.z, h.surface_point.u, h.surface_point.v])
            qrow.append(1 if s.quick_ray_intersection(ray) else 0)
        per_shape.append(row)
        quick.append(qrow)
        h = world.ray_intersection(ray)
        if h is None:
            per_world.append([0.0] * 11)
        else:
            idx = [i for i, s in enumerate(world.shapes) if s.material is h.material and
                   s.ray_intersection(ray) is not None and s.ray_intersection(ray).t == h.t][0]
            per_world.append([1.0, h.t, h.world_point.x, h.world_point.y, h.world_point.z, h.normal.x,
                              h.normal.y, h.normal.z, h.surface_point.u, h.surface_point.v, float(idx)])
    # is_point_visible
    vis_in, vis_out = [], []
    for k in range(200):
        a = (8 * r() - 4, 8 * r() - 4, 8 * r() - 4)
        b = (8 * r() - 4, 8 * r() - 4, 8 * r() - 4)
        vis_in.append(list(a) + list(b))
        vis_out.append(1 if world.is_point_visible(Point(*a), Point(*b)) else 0)
    assert n == scene.n_shapes
    save("g3_shapes", rays=np.array(rays), per_shape=np.array(per_shape), per_world=np.array(per_world),
         quick=np.array(quick, dtype=np.int32), vis_in=np.array(vis_in),
         vis_out=np.array(vis_out, dtype=np.int32), **scene.to_dict())


def g4_camera():
    g = PCG(9, 2)
    r = g.random_float
    cams = [
        PerspectiveCamera(screen_distance=1.0, aspect_ratio=2.0),
        PerspectiveCamera(screen_distance=1.0, aspect_ratio=1280 / 720, transformation=translation(Vec(-1.0, 0.0, 1.0))),
        PerspectiveCamera(screen_distance=2.5, aspect_ratio=1.0,
                          transformation=rotation_z(30.0) * translation(Vec(-4.0, 0.0, 1.0))),
        OrthogonalCamera(aspect_ratio=2.0),
        OrthogonalCamera(aspect_ratio=16 / 9, transformation=translation(-Vec(0.0, 1.0, 0.0) * 2.0) * rotation_z(90)),
    ]
    out = {}
    for ci, cam in enumerate(cams):
        fc = flatten.flatten_camera(cam)
        uv = [(0.0, 0.0), (1.0, 0.0), (0.0, 1.0), (1.0, 1.0), (0.5, 0.5)] + [(r(), r()) for _ in range(20)]
        rays = []
        for u, v in uv:
            ray = cam.fire_ray(u, v)
            rays.append([ray.origin.x, ray.origin.y, ray.origin.z, ray.dir.x, ray.dir.y, ray.dir.z,
                         ray.tmin, ray.tmax])
        # through ImageTracer.fire_ray at 1280x720
        img = HdrImage(1280, 720)
        tr = ImageTracer(img, cam)
        pix = [(0, 0, 0.5, 0.5), (1279, 719, 0.5, 0.5), (640, 360, 0.5, 0.5), (0, 719, 0.0, 1.0)]
        pix += [(int(r() * 1279), int(r() * 719), r(), r()) for _ in range(20)]
        prays = []
        for col, row, up, vp in pix:
            ray = tr.fire_ray(col, row, u_pixel=up, v_pixel=vp)
            prays.append([ray.origin.x, ray.origin.y, ray.origin.z, ray.dir.x, ray.dir.y, ray.dir.z,
                          ray.tmin, ray.tmax])
        out.update({f"c{ci}_uv": np.array(uv), f"c{ci}_rays": np.array(rays),
                    f"c{ci}_pix": np.array(pix), f"c{ci}_prays": np.array(prays)})
        out.update(abi.camera_to_dict(fc, prefix=f"c{ci}_cam_"))
    save("g4_camera", n_cams=np.array(len(cams)), **out)


def g6_g7_scatter_onb():
    g = PCG(21, 4)
    r = g.random_float
    normals, onbs = [], []
    pcg = PCG()  # test_all.py:991-1011
    for _ in range(100):
        nv = Vec(pcg.random_float(), pcg.random_float(), pcg.random_float())
        nv.normalize()
        e1, e2, e3 = create_onb_from_z(nv)
        normals.append((nv.x, nv.y, nv.z))
        onbs.append([e1.x, e1.y, e1.z, e2.x, e2.y, e2.z, e3.x, e3.y, e3.z])
    for _ in range(100):  # all octants
        nv = Vec(2 * r() - 1, 2 * r() - 1, 2 * r() - 1)
        nv.normalize()
        e1, e2, e3 = create_onb_from_z(Normal(nv.x, nv.y, nv.z))
        normals.append((nv.x, nv.y, nv.z))
        onbs.append([e1.x, e1.y, e1.z, e2.x, e2.y, e2.z, e3.x, e3.y, e3.z])
    sc_in, sc_out, sc_state = [], [], []
    for k in range(200):
        nv = Vec(2 * r() - 1, 2 * r() - 1, 2 * r() - 1)
        nv.normalize()
        inc = Vec(4 * r() - 2, 4 * r() - 2, 4 * r() - 2)
        pt = Point(4 * r() - 2, 4 * r() - 2, 4 * r() - 2)
        kind = k % 2
        seed = (int(r() * 1e6), int(r() * 1e6))
        p = PCG(*seed)
        brdf = DiffuseBRDF() if kind == abi.BRDF_DIFFUSE else SpecularBRDF()
        ray = brdf.scatter_ray(pcg=p, incoming_dir=inc, interaction_point=pt,
                               normal=Normal(nv.x, nv.y, nv.z), depth=3)
        sc_in.append([kind, seed[0], seed[1], nv.x, nv.y, nv.z, inc.x, inc.y, inc.z, pt.x, pt.y, pt.z])
        sc_out.append([ray.origin.x, ray.origin.y, ray.origin.z, ray.dir.x, ray.dir.y, ray.dir.z, ray.tmin, ray.tmax])
        sc_state.append(p.state)
        assert ray.depth == 3
    save("g6_scatter_onb", normals=np.array(normals), onb=np.array(onbs), sc_in=np.array(sc_in),
         sc_out=np.array(sc_out), sc_state=np.array(sc_state, dtype=np.uint64))


def g8_pigments():
    g = PCG(5, 77)
    r = g.random_float
    img = HdrImage(5, 3)
    for i in range(15):
        img.pixels[i] = Color(r(), r(), r())
    world = World()
    world.add_shape(Sphere(material=Material(
        brdf=DiffuseBRDF(CheckeredPigment(Color(1.0, 2.0, 3.0), Color(10.0, 20.0, 30.0), 2)),
        emitted_radiance=ImagePigment(img))))
    world.add_shape(Plane(material=Material(
        brdf=SpecularBRDF(CheckeredPigment(Color(0.3, 0.5, 0.1), Color(0.1, 0.2, 0.5), 7)),
        emitted_radiance=UniformPigment(Color(0.25, 0.5, 0.75)))))
    scene = flatten.flatten_world(world)
    uv = [(0.25, 0.25), (0.75, 0.25), (0.25, 0.75), (0.75, 0.75),  # test_all.py:913-935
          (0.0, 0.0), (1.0, 0.0), (0.0, 1.0), (1.0, 1.0),  # test_all.py:900-911
          (0.5, 0.5), (0.4999999999999999, 0.5000000000000001), (1.0 / 7, 2.0 / 7), (3.0 / 7, 1.0 - 1e-16)]
    uv += [(r(), r()) for _ in range(100)]
    from pytracer.geometry import Vec2d
    outs = []
    for u, v in uv:
        row = []
        for s in world.shapes:
            for pg in (s.material.brdf.pigment, s.material.emitted_radiance):
                c = pg.get_color(Vec2d(u, v))
                row += [c.r, c.g, c.b]
        outs.append(row)
    save("g8_pigments", uv=np.array(uv), colors=np.array(outs), **scene.to_dict())


def g5_frames():
    # --- C1: examples/demo.txt through the reference's own parser (fresh World: SURVEY H5) ---
    with open("/root/reference/examples/demo.txt", "rt") as f:
        scene = parse_scene(InputStream(f), {})
    demo_world, demo_cam = scene.world, scene.camera
    frame_fixture("g5_demo_onoff_160x120", demo_world, demo_cam, lambda: OnOffRenderer(demo_world), 160, 120)
    frame_fixture("g5_demo_flat_160x120", demo_world, demo_cam, lambda: FlatRenderer(demo_world), 160, 120)
    frame_fixture("g5_demo_pointlight_80x60", demo_world, demo_cam, lambda: PointLightRenderer(demo_world), 80, 60)
    frame_fixture("g5_demo_flat_jitter_40x30_seq", demo_world, demo_cam, lambda: FlatRenderer(demo_world),
                  40, 30, S=2, mode=abi.PCG_SEQ)
    frame_fixture("g5_demo_path_40x30_n2d2_pixel", demo_world, demo_cam,
                  lambda: PathTracer(demo_world, pcg=PCG(45, 54), num_of_rays=2, max_depth=2), 40, 30, S=0,
                  mode=abi.PCG_PIXEL, s0=45, q0=54)
    frame_fixture("g5_demo_path_24x18_n3d4_s2_pixel", demo_world, demo_cam,
                  lambda: PathTracer(demo_world, pcg=PCG(45, 54), num_of_rays=3, max_depth=4,
                                     russian_roulette_limit=2), 24, 18, S=2, mode=abi.PCG_PIXEL, s0=45, q0=54)

    # --- C2 shape: 32 spheres + plane, Flat ---
    w2 = ref_synthetic_world(32, with_plane=True)
    px = frame_fixture("g5_c2_flat_160x90", w2, ref_synthetic_camera(160, 90), lambda: FlatRenderer(w2), 160, 90)
    print("  C2-shape checksum", float(px.sum()), "(SURVEY: 22091.72636048037)")
    frame_fixture("g5_c2_onoff_64x36", w2, ref_synthetic_camera(64, 36), lambda: OnOffRenderer(w2), 64, 36)

    # --- C3 shape: 32 spheres, PathTracer ---
    w3 = ref_synthetic_world(32)
    px = frame_fixture("g5_c3_path_80x45_seq", w3, ref_synthetic_camera(80, 45),
                       lambda: PathTracer(w3, pcg=PCG(45, 54), num_of_rays=1, max_depth=3), 80, 45, S=4,
                       mode=abi.PCG_SEQ, jitter=(42, 54))
    print("  C3-shape checksum", float(px.sum()), "(SURVEY: 7799.597879510197)")
    frame_fixture("g5_c3_path_64x36_n1d3_s2_pixel", w3, ref_synthetic_camera(64, 36),
                  lambda: PathTracer(w3, pcg=PCG(45, 54), num_of_rays=1, max_depth=3), 64, 36, S=2,
                  mode=abi.PCG_PIXEL, s0=45, q0=54)
    frame_fixture("g5_c3_path_64x36_n2d2_s2_pixel", w3, ref_synthetic_camera(64, 36),
                  lambda: PathTracer(w3, pcg=PCG(45, 54), num_of_rays=2, max_depth=2), 64, 36, S=2,
                  mode=abi.PCG_PIXEL, s0=45, q0=54)
    frame_fixture("g5_c3_path_32x18_n1d3_s2_sample", w3, ref_synthetic_camera(32, 18),
                  lambda: PathTracer(w3, pcg=PCG(45, 54), num_of_rays=1, max_depth=3), 32, 18, S=2,
                  mode=abi.PCG_SAMPLE, s0=45, q0=54)

    # --- the reference's own renderer tests: 3x3 orthogonal (test_all.py:939-988) ---
    sphere = Sphere(transformation=translation(Vec(2, 0, 0)) * scaling(Vec(0.2, 0.2, 0.2)),
                    material=Material(brdf=DiffuseBRDF(pigment=UniformPigment(Color(1.0, 2.0, 3.0)))))
    w33 = World()
    w33.add_shape(sphere)
    frame_fixture("g5_test_onoff_3x3", w33, OrthogonalCamera(), lambda: OnOffRenderer(w33), 3, 3)
    frame_fixture("g5_test_flat_3x3", w33, OrthogonalCamera(), lambda: FlatRenderer(w33), 3, 3)

    # --- textured + orthogonal camera + point lights ---
    g = PCG(99, 1)
    r = g.random_float
    tex = HdrImage(8, 4)
    for i in range(32):
        tex.pixels[i] = Color(r(), r(), r())
    wt = World()
    wt.add_shape(Sphere(translation(Vec(1.0, 0.3, 0.2)) * rotation_y(25.0) * scaling(Vec(0.7, 0.5, 0.6)),
                        Material(DiffuseBRDF(ImagePigment(tex)), UniformPigment(Color(0.05, 0.0, 0.1)))))
    wt.add_shape(Plane(translation(Vec(0.0, 0.0, -0.6)),
                       Material(DiffuseBRDF(ImagePigment(tex)), CheckeredPigment(BLACK, Color(0.2, 0.2, 0.2), 3))))
    wt.add_shape(Sphere(translation(Vec(0.5, -0.8, 0.0)) * scaling(Vec(0.3, 0.3, 0.3)),
                        Material(SpecularBRDF(UniformPigment(Color(0.9, 0.8, 0.7))))))
    wt.add_light(PointLight(Point(-3.0, 4.0, 5.0), Color(1.0, 0.9, 0.8), 0.0))
    wt.add_light(PointLight(Point(-2.0, -5.0, 3.0), Color(0.2, 0.3, 0.9), 2.5))
    ocam = OrthogonalCamera(aspect_ratio=4 / 3, transformation=translation(Vec(-1.0, 0.0, 0.0)))
    frame_fixture("g5_tex_flat_48x36_ortho", wt, ocam, lambda: FlatRenderer(wt, background_color=Color(0.1, 0.2, 0.3)), 48, 36)
    frame_fixture("g5_tex_pointlight_48x36_ortho", wt, ocam,
                  lambda: PointLightRenderer(wt, background_color=Color(0.1, 0.2, 0.3),
                                             ambient_color=Color(0.05, 0.05, 0.1)), 48, 36)
    pcam = PerspectiveCamera(screen_distance=1.5, aspect_ratio=4 / 3, transformation=translation(Vec(-2.0, 0.0, 0.3)))
    frame_fixture("g5_tex_path_32x24_n2d3_s2_pixel", wt, pcam,
                  lambda: PathTracer(wt, background_color=Color(0.4, 0.5, 0.6), pcg=PCG(45, 54), num_of_rays=2,
                                     max_depth=3, russian_roulette_limit=1), 32, 24, S=2,
                  mode=abi.PCG_PIXEL, s0=45, q0=54)


def g5_c4():
    """BASELINE.json config 4 as specified -- the §8(d) "wide" 256-sphere scene, PathTracer N=1 D=5 rr=3,
    S=8 (spp 64) -- at a frame the pure-Python reference finishes in minutes, in both per-thread PCG modes."""
    w4 = ref_synthetic_world(256, wide=True)
    for mode, tag in ((abi.PCG_PIXEL, "pixel"), (abi.PCG_SAMPLE, "sample")):
        frame_fixture(f"g5_c4_path_32x18_n1d5_s8_{tag}", w4, ref_synthetic_camera(32, 18),
                      lambda: PathTracer(w4, pcg=PCG(45, 54), num_of_rays=1, max_depth=5, russian_roulette_limit=3),
                      32, 18, S=8, mode=mode, s0=45, q0=54)


def g5_cli():
    """What `python -m pytracer render examples/demo.txt` computes with its defaults (main.py:76-129: S=1, so the
    pixel is jittered; PathTracer N=10 D=3 seeds 45/54), driven into Mode PIXEL: the frames the CLI of this
    repository must reproduce.  Flat takes its seeds from ImageTracer's default PCG (42, 54)."""
    with open("/root/reference/examples/demo.txt", "rt") as f:
        scene = parse_scene(InputStream(f), {})
    world, cam = scene.world, scene.camera
    frame_fixture("g5_cli_demo_flat_s1_64x48", world, cam, lambda: FlatRenderer(world), 64, 48, S=1,
                  mode=abi.PCG_PIXEL, s0=42, q0=54)
    frame_fixture("g5_cli_demo_path_s1_32x24_n10d3", world, cam,
                  lambda: PathTracer(world, pcg=PCG(45, 54), num_of_rays=10, max_depth=3), 32, 24, S=1,
                  mode=abi.PCG_PIXEL, s0=45, q0=54)


def g5_seq():
    """The reference's OWN random streams (Mode SEQ), no Seeder: the verbatim ``ImageTracer(image, camera, S,
    pcg=PCG(...)).fire_all_rays(renderer)`` for the three renderers without a scattering stream -- whose jitter stream
    the device enters by jump-ahead -- including what ``python -m pytracer render --algorithm
    flat|onoff|pointlight examples/demo.txt`` computes with its defaults (main.py:76-129, 168-170: S = 1, ImageTracer's
    default PCG(42, 54))."""
    with open("/root/reference/examples/demo.txt", "rt") as f:
        scene = parse_scene(InputStream(f), {})
    world, cam = scene.world, scene.camera
    frame_fixture("g5_seq_cli_demo_flat_s1_64x48", world, cam, lambda: FlatRenderer(world), 64, 48, S=1, mode=abi.PCG_SEQ)
    frame_fixture("g5_seq_cli_demo_onoff_s1_48x36", world, cam, lambda: OnOffRenderer(world), 48, 36, S=1, mode=abi.PCG_SEQ)
    frame_fixture("g5_seq_cli_demo_pointlight_s1_48x36", world, cam, lambda: PointLightRenderer(world), 48, 36, S=1,
                  mode=abi.PCG_SEQ)
    frame_fixture("g5_seq_demo_onoff_s3_40x30", world, cam, lambda: OnOffRenderer(world), 40, 30, S=3, mode=abi.PCG_SEQ,
                  jitter=(7, 11))
    frame_fixture("g5_seq_demo_pointlight_s2_40x30", world, cam, lambda: PointLightRenderer(world), 40, 30, S=2,
                  mode=abi.PCG_SEQ, jitter=(123456789, 2 ** 40 + 3))
    w2 = ref_synthetic_world(32, with_plane=True)
    frame_fixture("g5_seq_c2_flat_s3_48x27", w2, ref_synthetic_camera(48, 27), lambda: FlatRenderer(w2), 48, 27, S=3,
                  mode=abi.PCG_SEQ, jitter=(45, 54))
    # orthogonal camera, textures, point lights
    g = PCG(99, 1)
    r = g.random_float
    tex = HdrImage(8, 4)
    for i in range(32):
        tex.pixels[i] = Color(r(), r(), r())
    wt = World()
    wt.add_shape(Sphere(translation(Vec(1.0, 0.3, 0.2)) * rotation_y(25.0) * scaling(Vec(0.7, 0.5, 0.6)),
                        Material(DiffuseBRDF(ImagePigment(tex)), UniformPigment(Color(0.05, 0.0, 0.1)))))
    wt.add_shape(Plane(translation(Vec(0.0, 0.0, -0.6)),
                       Material(DiffuseBRDF(ImagePigment(tex)), CheckeredPigment(BLACK, Color(0.2, 0.2, 0.2), 3))))
    wt.add_light(PointLight(Point(-3.0, 4.0, 5.0), Color(1.0, 0.9, 0.8), 0.0))
    ocam = OrthogonalCamera(aspect_ratio=4 / 3, transformation=translation(Vec(-1.0, 0.0, 0.0)))
    frame_fixture("g5_seq_tex_pointlight_s2_32x24_ortho", wt, ocam,
                  lambda: PointLightRenderer(wt, background_color=Color(0.1, 0.2, 0.3), ambient_color=Color(0.05, 0.05, 0.1)),
                  32, 24, S=2, mode=abi.PCG_SEQ)


def g5_sample():
    """Mode SAMPLE (one generator per sample) is what `GpuImageTracer` and the CLI give the path tracer by default
    since round 4: more frames of the verbatim reference driven into it (the re-seeding proxy, `Seeder(per_sample=True)`) --
    the reference's own demo scene with several rays per hit (the tree kernel's case), an odd number of samples per side, and
    the textured world through a perspective camera with roulette from depth 1."""
    with open("/root/reference/examples/demo.txt", "rt") as f:
        scene = parse_scene(InputStream(f), {})
    world, cam = scene.world, scene.camera
    frame_fixture("g5_demo_path_24x18_n3d2_s2_sample", world, cam,
                  lambda: PathTracer(world, pcg=PCG(45, 54), num_of_rays=3, max_depth=2), 24, 18, S=2,
                  mode=abi.PCG_SAMPLE, s0=45, q0=54)
    w3 = ref_synthetic_world(32)
    frame_fixture("g5_c3_path_24x14_n1d3_s3_sample", w3, ref_synthetic_camera(24, 14),
                  lambda: PathTracer(w3, pcg=PCG(45, 54), num_of_rays=1, max_depth=3), 24, 14, S=3,
                  mode=abi.PCG_SAMPLE, s0=45, q0=54)
    w2 = ref_synthetic_world(32, with_plane=True)
    frame_fixture("g5_c2plane_path_32x18_n2d3_s2_sample", w2, ref_synthetic_camera(32, 18),
                  lambda: PathTracer(w2, pcg=PCG(7, 11), num_of_rays=2, max_depth=3, russian_roulette_limit=1), 32, 18, S=2,
                  mode=abi.PCG_SAMPLE, s0=7, q0=11)


def g9_furnace():
    """test_all.py:1015-1051: closed diffuse unit sphere, N=1, D=100, rr_limit=101."""
    pcg = PCG()
    rows = []
    scenes = {}
    for i in range(5):
        world = World()
        emitted = pcg.random_float()
        refl = pcg.random_float() * 0.9
        mat = Material(brdf=DiffuseBRDF(pigment=UniformPigment(Color(1.0, 1.0, 1.0) * refl)),
                       emitted_radiance=UniformPigment(Color(1.0, 1.0, 1.0) * emitted))
        world.add_shape(Sphere(material=mat))
        state_before = pcg.state
        pt = PathTracer(pcg=pcg, num_of_rays=1, world=world, max_depth=100, russian_roulette_limit=101)
        c = pt(Ray(origin=Point(0, 0, 0), dir=Vec(1, 0, 0)))
        rows.append([emitted, refl, c.r, c.g, c.b, emitted / (1.0 - refl)])
        scenes.update(flatten.flatten_world(world).to_dict(prefix=f"f{i}_scene_"))
        scenes[f"f{i}_state_before"] = np.array(state_before, dtype=np.uint64)
        scenes[f"f{i}_state_after"] = np.array(pcg.state, dtype=np.uint64)
    scenes["inc"] = np.array(pcg.inc, dtype=np.uint64)
    save("g9_furnace", rows=np.array(rows), **scenes)


def g10_postprocess():
    """main.py:203-213 on two frames: PFM bytes (both endiannesses), average luminosity, normalize + clamp,
    LDR bytes (gamma 1.0 and 2.2).  Pillow is not needed: the LDR integers are computed as
    write_ldr_image computes them (hdrimages.py:160-166)."""
    from io import BytesIO
    from pytracer.hdrimages import Endianness
    out = {}
    for tag, name in (("a", "g5_c2_flat_160x90"), ("b", "g5_demo_path_40x30_n2d2_pixel")):
        src = os.path.join(OUT_DIR, name + ".npz")
        px = np.load(src if os.path.exists(src) else os.path.join(HERE, name + ".npz"))["pixels"]
        h, w = px.shape[:2]
        img = HdrImage(w, h)
        img.pixels = [Color(*map(float, px[y, x])) for y in range(h) for x in range(w)]
        for endian, key in ((Endianness.LITTLE_ENDIAN, "le"), (Endianness.BIG_ENDIAN, "be")):
            buf = BytesIO()
            img.write_pfm(buf, endianness=endian)
            out[f"{tag}_pfm_{key}"] = np.frombuffer(buf.getvalue(), dtype=np.uint8)
        lum = img.average_luminosity()
        out[f"{tag}_lum"] = np.array(lum)
        out[f"{tag}_lum_delta0"] = np.array(img.average_luminosity(delta=1e-3))
        img.normalize_image(factor=1.0)
        img.clamp_image()
        out[f"{tag}_toned"] = pixels_array(img)
        for gamma, key in ((1.0, "g10"), (2.2, "g22")):
            out[f"{tag}_ldr_{key}"] = np.array(
                [[int(255 * math.pow(c, 1 / gamma)) for c in (p.r, p.g, p.b)] for p in img.pixels],
                dtype=np.int32).reshape(h, w, 3)
        out[f"{tag}_pixels"] = px
    save("g10_postprocess", **out)



if __name__ == "__main__":
    argv = sys.argv[1:]
    if "--out" in argv:
        at = argv.index("--out")
        OUT_DIR = os.path.abspath(argv[at + 1])
        os.makedirs(OUT_DIR, exist_ok=True)
        del argv[at:at + 2]
    table = {"g1": g1_pcg, "g2": g2_xform, "g3": g3_shapes, "g4": g4_camera, "g6": g6_g7_scatter_onb,
             "g8": g8_pigments, "g9": g9_furnace, "g5": g5_frames, "g10": g10_postprocess, "g5c4": g5_c4,
             "g5cli": g5_cli, "g5seq": g5_seq, "g5sample": g5_sample}
    if argv:
        parses = sum(1 for k in argv if k in ("g5", "g5cli", "g5seq", "g5sample"))
        if parses > 1:
            raise SystemExit("g5, g5cli, g5seq and g5sample all parse examples/demo.txt: run them in separate processes (SURVEY.md H5)")
        for k in argv:
            table[k]()
    else:
        import subprocess

        for k in ["g1", "g2", "g3", "g4", "g6", "g8", "g9", "g5", "g10", "g5cli", "g5c4", "g5seq", "g5sample"]:  # (g10 reads g5's frames)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--out", OUT_DIR, k], check=True,
                           env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
