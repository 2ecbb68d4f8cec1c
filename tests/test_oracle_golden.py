"""Pin the CPU oracle to the reference: every golden fixture (outputs of the reference itself,
tests/golden/make_golden.py) and the known-answer vectors of the reference's tests/test_all.py.

All comparisons are BIT-EXACT (sqr mode 0 = libm pow, as CPython evaluates ``x**2``)."""
import math

import numpy as np
import pytest

from pytracer_amd import abi
from tests import util


@pytest.fixture(autouse=True)
def _pow_mode(oracle):
    oracle.set_sqr_mode(oracle.SQR_POW)
    yield


# ---- G1: PCG (reference pin: test_all.py:872-887) ---------------------------------------------
def test_pcg_known_answer(oracle):
    p = oracle.Pcg()
    assert p.state == 1753877967969059832
    assert p.inc == 109
    assert [p.random() for _ in range(6)] == [2707161783, 2068313097, 3122475824, 2211639955,
                                              3215226955, 3421331566]


def test_pcg_golden(oracle):
    g = util.load("g1_pcg")
    for k, (s, q) in enumerate(g["seeds"]):
        p = oracle.Pcg(int(s), int(q))
        assert p.state == int(g["state"][k]) and p.inc == int(g["inc"][k])
        assert [p.random() for _ in range(16)] == [int(x) for x in g["outputs"][k]]
        p = oracle.Pcg(int(s), int(q))
        assert util.bits_equal([p.random_float() for _ in range(16)], g["floats"][k])


def test_host_pcg_matches(oracle):
    from pytracer_amd.hostmodel import PCG

    g = util.load("g1_pcg")
    for k, (s, q) in enumerate(g["seeds"]):
        p = PCG(int(s), int(q))
        assert p.state == int(g["state"][k]) and p.inc == int(g["inc"][k])
        assert [p.random() for _ in range(16)] == [int(x) for x in g["outputs"][k]]


# ---- G2: Transformation * Point / Vec / Normal --------------------------------------------------
def test_xform_golden(oracle):
    g = util.load("g2_xform")
    for k in range(g["m"].shape[0]):
        assert util.bits_equal(oracle.xform(g["m"][k], 0, g["vin"][k]), g["point"][k])
        assert util.bits_equal(oracle.xform(g["m"][k], 1, g["vin"][k]), g["vec"][k])
        assert util.bits_equal(oracle.xform(g["invm"][k], 2, g["vin"][k]), g["normal"][k])


def test_xform_known_answers(oracle):
    # test_all.py:393-417
    m = [1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0, 9.0, 9.0, 8.0, 7.0]
    invm = [-3.75, 2.75, -1, 0, 5.75, -4.75, 2.0, 1.0, -2.25, 2.25, -1.0, -2.0]
    assert np.allclose(oracle.xform(m, 1, [1.0, 2.0, 3.0]), [14.0, 38.0, 51.0])
    assert np.allclose(oracle.xform(m, 0, [1.0, 2.0, 3.0]), [18.0, 46.0, 58.0])
    assert np.allclose(oracle.xform(invm, 2, [3.0, 2.0, 4.0]), [-8.75, 7.75, -3.0])


# ---- G3: shapes and world -----------------------------------------------------------------------
def test_shapes_golden(oracle):
    g = util.load("g3_shapes")
    scene = abi.FlatScene.from_dict(g)
    n_hits = 0
    for k, ray in enumerate(g["rays"]):
        for i in range(scene.n_shapes):
            exp = g["per_shape"][k, i]
            got = oracle.shape_intersect(scene, i, ray)
            assert (got is not None) == bool(exp[0]), (k, i)
            if got is not None:
                n_hits += 1
                assert util.bits_equal(got[:9], exp[1:10]), (k, i, got, exp)
            assert oracle.shape_quick_intersect(scene, i, ray) == bool(g["quick"][k, i])
        expw = g["per_world"][k]
        gotw = oracle.world_intersect(scene, ray)
        assert (gotw is not None) == bool(expw[0])
        if gotw is not None:
            assert util.bits_equal(gotw[:9], expw[1:10]) and int(gotw[9]) == int(expw[10])
    assert n_hits > 300
    for k, io in enumerate(g["vis_in"]):
        assert oracle.is_point_visible(scene, io[:3], io[3:]) == bool(g["vis_out"][k])


def _unit_scene(kind, m12=None, inv12=None):
    eye = [1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0]
    z3 = np.zeros((3, 1))
    return abi.FlatScene(kind=[kind], invm=np.array(inv12 or eye).reshape(12, 1),
                         m=np.array(m12 or eye).reshape(12, 1), brdf_kind=[0], brdf_param=[0.0],
                         pig_kind=[0], pig_c1=np.ones((3, 1)), pig_c2=z3, pig_steps=[0.0], pig_tex=[-1],
                         emi_kind=[0], emi_c1=z3, emi_c2=z3, emi_steps=[0.0], emi_tex=[-1])


def test_sphere_known_answers(oracle):
    # test_all.py:608-650
    s = _unit_scene(abi.SHAPE_SPHERE)
    h = oracle.shape_intersect(s, 0, oracle.ray8([0, 0, 2], [0, 0, -1]))
    assert np.allclose(h[:9], [1.0, 0, 0, 1.0, 0, 0, 1.0, 0.0, 0.0])
    h = oracle.shape_intersect(s, 0, oracle.ray8([3, 0, 0], [-1, 0, 0]))
    assert np.allclose(h[:9], [2.0, 1.0, 0, 0, 1.0, 0, 0, 0.0, 0.5])
    assert oracle.shape_intersect(s, 0, oracle.ray8([0, 10, 2], [0, 0, -1])) is None
    # inner hit, test_all.py:637-650
    h = oracle.shape_intersect(s, 0, oracle.ray8([0, 0, 0], [1, 0, 0]))
    assert np.allclose(h[:9], [1.0, 1.0, 0, 0, -1.0, 0, 0, 0.0, 0.5])


def test_plane_known_answers(oracle):
    # test_all.py:752-770
    s = _unit_scene(abi.SHAPE_PLANE)
    h = oracle.shape_intersect(s, 0, oracle.ray8([0, 0, 1], [0, 0, -1]))
    assert np.allclose(h[:9], [1.0, 0, 0, 0, 0, 0, 1.0, 0.0, 0.0])
    assert oracle.shape_intersect(s, 0, oracle.ray8([0, 0, 1], [0, 0, 1])) is None
    assert oracle.shape_intersect(s, 0, oracle.ray8([0, 0, 1], [1, 0, 0])) is None
    # uv wrap, test_all.py:800-819
    h = oracle.shape_intersect(s, 0, oracle.ray8([0.25, 0.75, 1], [0, 0, -1]))
    assert np.allclose(h[7:9], [0.25, 0.75])
    h = oracle.shape_intersect(s, 0, oracle.ray8([4.25, 7.75, 1], [0, 0, -1]))
    assert np.allclose(h[7:9], [0.25, 0.75])


# ---- G4: cameras and ImageTracer.fire_ray ---------------------------------------------------------
def test_camera_golden(oracle):
    g = util.load("g4_camera")
    for ci in range(int(g["n_cams"])):
        cam = abi.camera_from_dict(g, prefix=f"c{ci}_cam_")
        for (u, v), exp in zip(g[f"c{ci}_uv"], g[f"c{ci}_rays"]):
            assert util.bits_equal(oracle.camera_fire_ray(cam, float(u), float(v)), exp)
        for (col, row, up, vp), exp in zip(g[f"c{ci}_pix"], g[f"c{ci}_prays"]):
            got = oracle.tracer_fire_ray(cam, 1280, 720, int(col), int(row), float(up), float(vp))
            assert util.bits_equal(got, exp)


def test_tracer_orientation(oracle):
    # test_all.py:562-574: 4x2 image, perspective camera, aspect 2
    eye = [1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0]
    cam = abi.make_camera(abi.CAMERA_PERSPECTIVE, eye, 1.0, 2.0)
    r = oracle.tracer_fire_ray(cam, 4, 2, 0, 0, 0.0, 0.0)
    assert np.allclose(r[:3] + r[3:6], [0.0, 2.0, 1.0])
    r = oracle.tracer_fire_ray(cam, 4, 2, 3, 1, 1.0, 1.0)
    assert np.allclose(r[:3] + r[3:6], [0.0, -2.0, -1.0])
    r1 = oracle.tracer_fire_ray(cam, 4, 2, 0, 0, 2.5, 1.5)
    r2 = oracle.tracer_fire_ray(cam, 4, 2, 2, 1, 0.5, 0.5)
    assert np.allclose(r1, r2)


# ---- G6/G7: scattering and the orthonormal basis ---------------------------------------------------
def test_onb_and_scatter_golden(oracle):
    g = util.load("g6_scatter_onb")
    for n, exp in zip(g["normals"], g["onb"]):
        assert util.bits_equal(oracle.onb(n).reshape(-1), exp)
    for row, exp, st in zip(g["sc_in"], g["sc_out"], g["sc_state"]):
        p = oracle.Pcg(int(row[1]), int(row[2]))
        got = oracle.scatter(int(row[0]), p, row[6:9], row[9:12], row[3:6], 3)
        assert util.bits_equal(got, exp), (row, got, exp)
        assert p.state == int(st)


# ---- G8: pigments -----------------------------------------------------------------------------------
def test_pigments_golden(oracle):
    g = util.load("g8_pigments")
    scene = abi.FlatScene.from_dict(g)
    for (u, v), exp in zip(g["uv"], g["colors"]):
        got = []
        for i in range(scene.n_shapes):
            for emitted in (False, True):
                got += list(oracle.pigment(scene, i, emitted, float(u), float(v)))
        assert util.bits_equal(got, exp)


# ---- G9: furnace (test_all.py:1015-1051) --------------------------------------------------------------
def test_furnace_golden(oracle):
    g = util.load("g9_furnace")
    par = abi.make_params(1, 1, abi.RENDERER_PATHTRACER, num_of_rays=1, max_depth=100, rr_limit=101)
    for i, row in enumerate(g["rows"]):
        scene = abi.FlatScene.from_dict(g, prefix=f"f{i}_scene_")
        p = oracle.Pcg()
        p.st[0] = int(g[f"f{i}_state_before"])
        p.st[1] = int(g["inc"])
        col, n_rays = oracle.radiance(scene, par, p, oracle.ray8([0, 0, 0], [1, 0, 0]))
        assert util.bits_equal(col, row[2:5])
        assert p.state == int(g[f"f{i}_state_after"])
        assert n_rays == 101
        assert col[0] == pytest.approx(row[5], rel=1e-3)


# ---- G5: whole frames through fire_all_rays --------------------------------------------------------------
@pytest.mark.parametrize("name", util.FRAME_FIXTURES)
def test_frame_golden(oracle, name):
    scene, cam, par, pixels = util.load_frame(name)
    out, n_rays = oracle.render(scene, cam, par, n_threads=0, sqr_mode=oracle.SQR_POW)
    assert out.shape == pixels.shape
    assert util.bits_equal(out, pixels), f"max rel err {util.rel_err(out, pixels).max()}"
    assert n_rays >= par.width * par.height * max(1, par.samples_per_side) ** 2


def test_checksums_match_baseline_md():
    # BASELINE.md §2 checksums, measured by the survey on the reference
    for name, expect in (("g5_demo_onoff_160x120", 44139.0), ("g5_demo_flat_160x120", 19389.899999998433),
                         ("g5_c2_flat_160x90", 22091.72636048037), ("g5_c3_path_80x45_seq", 7799.597879510197)):
        px = util.load(name)["pixels"]
        assert math.fsum(px.reshape(-1).tolist()) == pytest.approx(expect, rel=1e-12)


def test_interpreted_loop_equals_the_c_oracle(oracle):
    """oracle/pyloop.py (the interpreted baseline bench.py times on the GPU box) is the same arithmetic as pt_oracle.c in
    its x*x mode: bit-identical frames, both renderers, both cameras, checkered planes and uv of spheres included; and it
    agrees with the REFERENCE's own frame (x**2 arithmetic) to 1e-12."""
    from oracle import pyloop
    from pytracer_amd import flatten, hostmodel as hm, scenes

    flat = flatten.flatten_world(scenes.synthetic_world(12, with_plane=True))
    demo_world, demo_cam = scenes.demo_world(clock=150.0)
    cases = [(flat, flatten.flatten_camera(scenes.synthetic_camera(40, 22)), 40, 22),
             (flat, flatten.flatten_camera(hm.OrthogonalCamera(40 / 22, hm.translation(hm.Vec(-1.0, 0.0, 1.0)) * hm.scaling(hm.Vec(1.0, 3.0, 2.0)))), 40, 22),
             (flatten.flatten_world(demo_world), flatten.flatten_camera(demo_cam), 32, 24)]
    for scene, cam, W, H in cases:
        for renderer in (abi.RENDERER_FLAT, abi.RENDERER_ONOFF):
            par = abi.make_params(W, H, renderer)
            got, n = pyloop.render(scene, cam, par)
            want, n_rays = oracle.render(scene, cam, par, sqr_mode=oracle.SQR_MUL)
            oracle.set_sqr_mode(oracle.SQR_POW)
            assert n == n_rays == W * H
            assert util.bits_equal(got, want), (renderer, W, H)
    scene, cam, par, pixels = util.load_frame("g5_c2_flat_160x90")
    small = abi.copy_params(par)
    got, _ = pyloop.render(scene, cam, small)
    assert util.rel_err(got, pixels).max() <= 1e-12


def test_sqr_mode_mul_is_close(oracle):
    """x*x instead of pow(x, 2): the device arithmetic.  Not bit-identical in general, but the
    frames stay far inside the 1e-5 contract."""
    scene, cam, par, pixels = util.load_frame("g5_c2_flat_160x90")
    out, _ = oracle.render(scene, cam, par, sqr_mode=oracle.SQR_MUL)
    assert util.rel_err(out, pixels).max() <= 1e-5
    oracle.set_sqr_mode(oracle.SQR_POW)
