"""GPU parity tests (run with ``-m gpu`` on an MI355X): the HIP path, called through the C-ABI, against

* the golden fixtures (outputs of the reference itself),
* the CPU oracle on the same seeded inputs (sqr mode ``x*x`` = the device arithmetic),
* size-independent properties at BASELINE.json's full sizes.

Bars (BASELINE.json north_star): bit-exact for the on/off renderer; <= 1e-5 relative per channel
elsewhere (fp64 output, so the floor is libm-vs-ocml last-ulp differences).  Where no
libm transcendental is involved the device must equal the oracle's ``x*x`` mode bit for bit.
"""
import math
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from pytracer_amd import abi
from tests import util

pytestmark = pytest.mark.gpu

TOL = 1e-5

# (measurement switches may force the round-2 kernels, e.g. PTRACE_TREE=0 PTRACE_TILE4=0 pytest -m gpu: same images)
def _tuned(name):
    from pytracer_amd import device

    return device.get_tuning(name)


# (num_of_rays > 1: the tree kernel, unless a switch forces the one-queue alternative on every frame or takes the tree kernel away)
# (PTRACE_CULL=0: every frame on the one-lane-per-pixel kernels, the brute-force device path the culled kernels are checked against)
NO_CULL = _tuned("cull") == 0
TREE_KERNEL = abi.KERNEL_PATH if NO_CULL else (
    abi.KERNEL_PATH_REGIONS if _tuned("tree") == 0 else (abi.KERNEL_PATH if _tuned("qchoice") == 2 else abi.KERNEL_PATH_TREE))
REGIONS_KERNEL = abi.KERNEL_PATH if NO_CULL else abi.KERNEL_PATH_REGIONS
TILE4_KERNEL = abi.KERNEL_SIMPLE if NO_CULL else (abi.KERNEL_TILE4 if _tuned("tile4") != 0 else abi.KERNEL_TILE)
TILE_KERNELS = (abi.KERNEL_SIMPLE,) if NO_CULL else (abi.KERNEL_TILE, abi.KERNEL_TILE4)


@pytest.fixture(scope="module")
def dev():
    from pytracer_amd import device

    assert device.device_count() >= 1, "no HIP device visible"
    return device


def _device_ok(par: abi.Params) -> bool:
    """SEQ fixtures (the reference's own streams) run on the device unless they are path-traced: the jitter stream is
    entered by jump-ahead (two draws per sample), the path tracer's scattering stream is serial by construction."""
    return not (par.pcg_mode == abi.PCG_SEQ and par.renderer == abi.RENDERER_PATHTRACER)


def _uses_libm(scene: abi.FlatScene, par: abi.Params) -> bool:
    """True when the frame can depend on sin/cos/atan2/acos (the only non-bit-exact operations)."""
    sphere_uv = bool(np.any((scene.kind == abi.SHAPE_SPHERE) &
                            ((scene.pig_kind != abi.PIGMENT_UNIFORM) | (scene.emi_kind != abi.PIGMENT_UNIFORM))))
    if par.renderer == abi.RENDERER_PATHTRACER:
        return True  # diffuse scattering uses sin/cos
    if par.renderer == abi.RENDERER_POINTLIGHT:
        return sphere_uv or bool(np.any(scene.brdf_kind == abi.BRDF_SPECULAR))  # specular eval: acos
    if par.renderer == abi.RENDERER_FLAT:
        return sphere_uv
    return False


# ---- device primitives --------------------------------------------------------------------------------
def test_sqrt_div_are_ieee_exact(dev):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(0, 1e3, 200000), 10.0 ** rng.uniform(-300, 300, 200000),
                        np.array([0.0, 1.0, 2.0, 4.0, 1e-310, 5e-324])])
    assert util.bits_equal(dev.probe(0, x), np.sqrt(x))
    a = rng.normal(size=400000) * 10.0 ** rng.integers(-20, 20, 400000)
    b = rng.normal(size=400000) * 10.0 ** rng.integers(-20, 20, 400000)
    assert util.bits_equal(dev.probe(1, a, b), a / b)
    # PCG floats: integer / 0xFFFFFFFF
    n = rng.integers(0, 2 ** 32, 200000).astype(np.float64)
    assert util.bits_equal(dev.probe(1, n, np.full_like(n, 4294967295.0)), n / 4294967295.0)
    assert util.bits_equal(dev.probe(6, a), np.floor(a))


def test_no_fma_contraction(dev):
    rng = np.random.default_rng(2)
    a, b = rng.normal(size=100000), rng.normal(size=100000)
    got = dev.probe(7, a, b)
    assert util.bits_equal(got, (a * b) + a), "a*b+c was fused: build without -ffp-contract=off?"


def test_transcendental_ulp_distance(dev):
    """ocml vs glibc: report and bound the distance (not bit-exact by construction)."""
    rng = np.random.default_rng(3)
    x = rng.uniform(0, 2 * math.pi, 20000)
    ref_sin = np.array([math.sin(v) for v in x])
    ref_cos = np.array([math.cos(v) for v in x])
    for op, ref in ((2, ref_sin), (3, ref_cos)):
        got = dev.probe(op, x)
        assert np.max(np.abs(got - ref)) <= 4 * np.finfo(np.float64).eps
    y, xx = rng.uniform(-1, 1, 20000), rng.uniform(-1, 1, 20000)
    got = dev.probe(4, y, xx)
    ref = np.array([math.atan2(a, b) for a, b in zip(y, xx)])
    assert np.max(np.abs(got - ref)) <= 8 * np.finfo(np.float64).eps
    z = rng.uniform(-1, 1, 20000)
    got = dev.probe(5, z)
    ref = np.array([math.acos(v) for v in z])
    assert np.max(np.abs(got - ref)) <= 8 * np.finfo(np.float64).eps


# ---- golden frames ----------------------------------------------------------------------------------------
def test_pcg_known_answers_and_jump_ahead_on_the_device(dev):
    """The device generator on its own: the outputs of PCG(45, seq) against the host copy that
    test_oracle_golden pins to the reference (test_all.py:872-887 and the g1 fixture), and `pcg_advance` -- the
    jump the second pass of the path tracer speculates with -- against n single steps, n up to 5000."""
    from pytracer_amd.hostmodel import PCG

    seqs = np.array([54, 55, 54 + 921599, 2 ** 40 + 7, 12345678901], dtype=np.float64)
    for k in (0, 1, 5, 15):
        got = dev.probe(8, seqs, np.full_like(seqs, k))
        gotf = dev.probe(9, seqs, np.full_like(seqs, k))
        for i, q in enumerate(seqs):
            g = PCG(45, int(q))
            outs = [g.random() for _ in range(k + 1)]
            assert int(got[i]) == outs[-1]
            g = PCG(45, int(q))
            fl = [g.random_float() for _ in range(k + 1)]
            assert gotf[i] == fl[-1]
    n = np.array([0, 1, 2, 3, 4, 7, 8, 63, 64, 255, 256, 1000, 4095, 4096, 5000], dtype=np.float64)
    for q in (54.0, 999983.0):
        assert np.all(dev.probe(10, np.full_like(n, q), n) == 1.0)


@pytest.mark.parametrize("name", util.FRAME_FIXTURES)
def test_frame_vs_reference_golden(dev, oracle, name):
    scene, cam, par, pixels = util.load_frame(name)
    if not _device_ok(par):
        pytest.skip("path tracer under the reference's global sequential scattering stream (serial by construction)")
    with dev.DeviceScene(scene) as ds:
        out = ds.render(cam, par)
        st = ds.stats()
    assert out.shape == pixels.shape
    # (1) against the oracle in the device's own arithmetic (x*x): exact unless libm is involved
    ora, n_rays = oracle.render(scene, cam, par, sqr_mode=oracle.SQR_MUL)  # (PT_PCG_SEQ: the oracle's serial loop)
    oracle.set_sqr_mode(oracle.SQR_POW)
    if not _uses_libm(scene, par):
        assert util.bits_equal(out, ora), f"device != oracle(x*x): max rel {util.rel_err(out, ora).max()}"
    # (2) against the reference's own output
    if par.renderer == abi.RENDERER_ONOFF:
        assert util.bits_equal(out, pixels), "OnOff must be bit-exact"
    err = util.rel_err(out, pixels)
    bad = int((err > TOL).any(axis=-1).sum())
    print(f"{name}: max rel err {err.max():.3e}, pixels over {TOL:g}: {bad}/{err.shape[0] * err.shape[1]}")
    assert bad == 0, f"{bad} pixels differ from the reference by more than {TOL}"
    # ray accounting equals the oracle's count of world queries
    if not _uses_libm(scene, par) or par.renderer != abi.RENDERER_PATHTRACER:
        assert st.n_rays == n_rays
    assert st.n_pixels == par.width * par.height


def test_f32_output_is_rounded_f64(dev):
    scene, cam, par, _ = util.load_frame("g5_c2_flat_160x90")
    with dev.DeviceScene(scene) as ds:
        o64 = ds.render(cam, par)
        o32 = ds.render(cam, abi.copy_params(par, out_format=abi.OUT_F32))
    assert o32.dtype == np.float32 and np.array_equal(o32, o64.astype(np.float32))


# ---- the synthetic benchmark scenes against the oracle, at sizes the oracle finishes in seconds ------------
def _synthetic(n_spheres, with_plane, wide, w, h):
    from pytracer_amd import flatten, scenes

    world = scenes.synthetic_world(n_spheres, with_plane=with_plane, wide=wide)
    return flatten.flatten_world(world), flatten.flatten_camera(scenes.synthetic_camera(w, h))


def test_c2_flat_320x180_bit_exact_vs_oracle(dev, oracle):
    scene, cam = _synthetic(32, True, False, 320, 180)
    par = abi.make_params(320, 180, abi.RENDERER_FLAT)
    with dev.DeviceScene(scene) as ds:
        out = ds.render(cam, par)
    ora, _ = oracle.render(scene, cam, par, sqr_mode=oracle.SQR_MUL)
    oracle.set_sqr_mode(oracle.SQR_POW)
    assert util.bits_equal(out, ora)


def test_c5_many_spheres_flat_vs_oracle(dev, oracle):
    # 10 k spheres (config 5) at a reduced frame: 96x54 x 10 000 shapes = 52 M tests for the oracle
    scene, cam = _synthetic(10000, False, True, 96, 54)
    par = abi.make_params(96, 54, abi.RENDERER_FLAT)
    with dev.DeviceScene(scene) as ds:
        out = ds.render(cam, par)
    ora, _ = oracle.render(scene, cam, par, sqr_mode=oracle.SQR_MUL)
    oracle.set_sqr_mode(oracle.SQR_POW)
    assert util.bits_equal(out, ora)


def test_tree_kernel_deep_and_wide(dev, oracle):
    """The node stack of pt_path_tree_kernel at depth 14 (N = 2: up to 2^15 rays per pixel, roulette from depth 3 on) and a
    family of 200 children (four rounds of 64 lanes per node) on a small frame, against the oracle."""
    scene, cam = _synthetic(32, False, False, 24, 16)
    for n_rays, depth, rr in ((2, 14, 3), (200, 1, 3), (3, 40, 2)):
        par = abi.make_params(24, 16, abi.RENDERER_PATHTRACER, samples_per_side=1, num_of_rays=n_rays, max_depth=depth, rr_limit=rr,
                              path_state=45, path_seq=54)
        with dev.DeviceScene(scene) as ds:
            out = ds.render(cam, par)
            st = ds.stats()
        assert st.kernel == TREE_KERNEL
        ora, n = oracle.render(scene, cam, par, sqr_mode=oracle.SQR_MUL)
        oracle.set_sqr_mode(oracle.SQR_POW)
        bad = int((util.rel_err(out, ora) > TOL).any(axis=-1).sum())
        assert bad <= 1, (n_rays, depth, bad)
        assert abs(int(st.n_rays) - n) <= 8 + n // 100000, (n_rays, depth, int(st.n_rays), n)
    # a rank's rows of a partitioned frame are the same bits (per-pixel seeds; the tree kernel takes one pixel per unit)
    par = abi.make_params(24, 16, abi.RENDERER_PATHTRACER, samples_per_side=2, num_of_rays=4, max_depth=3, rr_limit=2,
                          path_state=45, path_seq=54)
    with dev.DeviceScene(scene) as ds:
        full = ds.render(cam, par)
        n_full = int(ds.stats().n_rays)
        got, n_sum = np.zeros_like(full), 0
        for rank in range(3):
            p = abi.copy_params(par, n_ranks=3, rank=rank, row_block=5)
            got[abi.rows_for_rank(16, 5, 3, rank)] = ds.render(cam, p)
            assert ds.stats().kernel == TREE_KERNEL
            n_sum += int(ds.stats().n_rays)
    assert util.bits_equal(got, full) and n_sum == n_full


@pytest.mark.parametrize("n_rays,depth,rr", [(3, 3, 2), (10, 2, 0), (5, 4, 3)])
def test_tree_kernel_in_a_world_without_a_dome(dev, oracle, n_rays, depth, rr):
    """num_of_rays > 1 where scattered rays can MISS (no sphere around the scene), over mirrors (no scatter draws) and a
    checkered plane: the draw counts the children's start states are speculated from take every value the reference
    can produce, and whatever the speculation gets wrong is only done again -- the frame must equal the oracle's."""
    from pytracer_amd import flatten, hostmodel as hm

    g = hm.PCG(77, 5)
    r = g.random_float
    w = hm.World()
    for i in range(14):
        rad = 0.25 + 0.35 * r()
        brdf = hm.SpecularBRDF(hm.UniformPigment(hm.Color(0.3 + 0.6 * r(), 0.3 + 0.6 * r(), 0.3 + 0.6 * r()))) if i % 3 == 0 else \
            hm.DiffuseBRDF(hm.UniformPigment(hm.Color(0.2 + 0.7 * r(), 0.2 + 0.7 * r(), 0.2 + 0.7 * r())))
        emit = hm.UniformPigment(hm.Color(2.0 * r(), 2.0 * r(), 2.0 * r())) if i % 4 == 1 else hm.UniformPigment(hm.BLACK)
        w.add_shape(hm.Sphere(hm.translation(hm.Vec(1.5 + 3.5 * r(), 3.0 * (r() - 0.5), rad + 0.8 * r())) * hm.scaling(hm.Vec(rad, rad, rad)),
                              hm.Material(brdf, emit)))
    w.add_shape(hm.Plane(hm.Transformation(), hm.Material(hm.DiffuseBRDF(hm.CheckeredPigment(hm.Color(0.3, 0.5, 0.1), hm.Color(0.1, 0.2, 0.5), 4)))))
    scene = flatten.flatten_world(w)
    W, H = 96, 56
    cam = flatten.flatten_camera(hm.PerspectiveCamera(1.0, W / H, hm.translation(hm.Vec(-1.0, 0.0, 1.0))))
    cam_o = flatten.flatten_camera(hm.OrthogonalCamera(W / H, hm.translation(hm.Vec(-1.0, 0.0, 1.2)) * hm.scaling(hm.Vec(1.0, 2.5, 1.5))))
    for S, mode, cam in ((1, abi.PCG_PIXEL, cam), (2, abi.PCG_SAMPLE, cam), (1, abi.PCG_PIXEL, cam_o)):
        par = abi.make_params(W, H, abi.RENDERER_PATHTRACER, samples_per_side=S, num_of_rays=n_rays, max_depth=depth, rr_limit=rr,
                              pcg_mode=mode, path_state=45, path_seq=54, background=(0.05, 0.1, 0.3))
        with dev.DeviceScene(scene) as ds:
            out = ds.render(cam, par)
            st = ds.stats()
        assert st.kernel == TREE_KERNEL
        ora, n = oracle.render(scene, cam, par, sqr_mode=oracle.SQR_MUL)
        oracle.set_sqr_mode(oracle.SQR_POW)
        bad = int((util.rel_err(out, ora) > TOL).any(axis=-1).sum())
        assert bad <= 1, (S, mode, bad)  # (every logged run: 0; observed + 1)
        assert abs(int(st.n_rays) - n) <= 8 + n // 100000, (S, mode, int(st.n_rays), n)


@pytest.mark.parametrize("W,H", [(17, 1), (1, 33), (16, 16), (15, 31), (33, 2), (2, 2)])
def test_sixteen_by_sixteen_tiles_at_awkward_frame_sizes(dev, oracle, W, H):
    """pt_tile4_kernel (OnOff / Flat, pixel-centre rays): frames narrower or lower than a tile, one pixel beyond a tile
    edge, a 2x2-block grid with idle waves -- against the oracle, both renderers, with and without the dome shortcut."""
    scene, cam = _synthetic(32, True, False, W, H)
    for renderer in (abi.RENDERER_FLAT, abi.RENDERER_ONOFF):
        par = abi.make_params(W, H, renderer)
        ora, n_rays = oracle.render(scene, cam, par, sqr_mode=oracle.SQR_MUL)
        oracle.set_sqr_mode(oracle.SQR_POW)
        with dev.DeviceScene(scene) as ds:
            for dome in (True, False):
                ds.set_dome_shortcut(dome)
                out = ds.render(cam, par)
                st = ds.stats()
                assert st.kernel == TILE4_KERNEL
                assert util.bits_equal(out, ora), (renderer, dome)
                assert int(st.n_rays) == n_rays == W * H
                assert (st.n_rays_resolved == 0) if not dome else (st.n_rays_resolved <= st.n_rays)


def test_rays_from_far_away_do_not_lose_spheres_to_the_grid(dev, oracle):
    """the grid's insertion margin is sized from the grid's own coordinates, the walk uses an fp32 copy of the
    ray -- a scattered ray starting ~1e4 grid extents away (here: reflected by a distant mirror back through a cluster
    of > 1024 spheres, where the grid is on) deviates by more than that margin.  Such lanes take the exhaustive filter
    (its slack scales with |o|); the frame must equal the oracle's bit for bit wherever no libm function is involved."""
    from pytracer_amd import flatten, hostmodel as hm

    g = hm.PCG(2024, 9)
    r = g.random_float
    w = hm.World()
    for i in range(1100):
        rad = 0.05 + 0.08 * r()
        col = hm.Color(0.2 + 0.7 * r(), 0.2 + 0.7 * r(), 0.2 + 0.7 * r())
        w.add_shape(hm.Sphere(hm.translation(hm.Vec(3.0 + 4.0 * r(), 4.0 * (r() - 0.5), 4.0 * (r() - 0.5))) * hm.scaling(hm.Vec(rad, rad, rad)),
                              hm.Material(hm.DiffuseBRDF(hm.UniformPigment(hm.BLACK)), hm.UniformPigment(col))))
    # a concave spherical mirror of radius 1e5 around the cluster: primary rays that pass between the spheres are
    # reflected 1e5 away, converge behind the cluster and run back through it -- scattered rays with |o| ~ 1e5
    w.add_shape(hm.Sphere(hm.translation(hm.Vec(5.0, 0.0, 0.0)) * hm.scaling(hm.Vec(1.0e5, 1.0e5, 1.0e5)),
                          hm.Material(hm.SpecularBRDF(hm.UniformPigment(hm.Color(0.9, 0.9, 0.9))))))
    scene = flatten.flatten_world(w)
    W, H = 192, 128
    cam = flatten.flatten_camera(hm.PerspectiveCamera(2.0, W / H, hm.translation(hm.Vec(-1.0, 0.0, 0.0))))
    par = abi.make_params(W, H, abi.RENDERER_PATHTRACER, samples_per_side=0, num_of_rays=1, max_depth=2, rr_limit=5,
                          path_state=45, path_seq=54, background=(0.0, 0.0, 0.0))
    with dev.DeviceScene(scene) as ds:
        out = ds.render(cam, par)
        st = ds.stats()
    ora, n = oracle.render(scene, cam, par, sqr_mode=oracle.SQR_MUL)
    oracle.set_sqr_mode(oracle.SQR_POW)
    reflected = int(st.n_rays) - W * H
    assert reflected > 2000, f"only {reflected} rays came back from the mirror: the scene does not exercise far origins"
    # pixels whose primary ray met the mirror (value = 0.9 x what came back) and whose reflected ray then met a sphere
    _, n_primary_only = oracle.render(scene, cam, abi.copy_params(par, max_depth=0), sqr_mode=oracle.SQR_MUL)
    assert n_primary_only == W * H
    came_back_lit = int(((ora.sum(axis=-1) > 0) & (oracle.render(scene, cam, abi.copy_params(par, max_depth=0), sqr_mode=oracle.SQR_MUL)[0].sum(axis=-1) == 0)).sum())
    assert came_back_lit > 300, f"only {came_back_lit} reflected rays meet a sphere of the cluster"
    assert int(st.n_rays) == n
    assert util.bits_equal(out, ora), f"{int((util.rel_err(out, ora) > 0).any(axis=-1).sum())} pixels differ"


def _cluster_world(n, spread, seed, rotated=False):
    """n small spheres bunched around the view axis (many survivors per tile: exercises survivor-mask
    bits >= 31 and several culling passes), optionally with non-diagonal transforms, plus duplicates."""
    from pytracer_amd import hostmodel as hm

    g = hm.PCG(seed, 3)
    r = g.random_float
    w = hm.World()
    w.add_shape(hm.Sphere(hm.scaling(hm.Vec(60.0, 60.0, 60.0)),
                          hm.Material(hm.DiffuseBRDF(hm.UniformPigment(hm.BLACK)), hm.UniformPigment(hm.Color(0.2, 0.3, 0.4)))))
    for i in range(n):
        rad = 0.03 + 0.1 * r()
        t = hm.translation(hm.Vec(2.0 + 4.0 * r(), spread * (r() - 0.5), 1.0 + spread * (r() - 0.5)))
        sc = hm.scaling(hm.Vec(rad, rad * (0.5 + r()), rad))
        T = t * hm.rotation_z(360 * r()) * hm.rotation_x(360 * r()) * sc if (rotated and i % 3 == 0) else t * sc
        mat = hm.Material(hm.DiffuseBRDF(hm.UniformPigment(hm.Color(r(), r(), r()))), hm.UniformPigment(hm.Color(r(), r(), r())))
        w.add_shape(hm.Sphere(T, mat))
        if i % 17 == 0:  # exact duplicate geometry, different material: the first one must win the tie
            w.add_shape(hm.Sphere(T, hm.Material(hm.DiffuseBRDF(hm.UniformPigment(hm.Color(9.0, 9.0, 9.0))))))
    w.add_shape(hm.Plane(hm.translation(hm.Vec(0.0, 0.0, -0.5)),
                         hm.Material(hm.DiffuseBRDF(hm.CheckeredPigment(hm.Color(0.3, 0.5, 0.1), hm.Color(0.1, 0.2, 0.5), 2)))))
    return w


@pytest.mark.parametrize("n,spread,rotated,W,H,S,renderer", [
    (120, 0.6, False, 203, 117, 0, abi.RENDERER_FLAT),    # sizes not multiples of 8
    (120, 0.6, True, 203, 117, 0, abi.RENDERER_ONOFF),
    (300, 1.5, True, 160, 96, 2, abi.RENDERER_FLAT),       # 5 culling passes, jittered samples
    (40, 0.2, False, 64, 64, 3, abi.RENDERER_FLAT),
    (400, 1.5, True, 203, 117, 2, abi.RENDERER_FLAT),      # > 256 shapes: two-level culling (cells, then tiles)
    (400, 1.0, False, 160, 96, 0, abi.RENDERER_ONOFF),
    (120, 0.6, False, 120, 72, 0, abi.RENDERER_POINTLIGHT),  # primary rays culled, shadow rays see every shape
    (300, 1.0, False, 96, 64, 2, abi.RENDERER_POINTLIGHT),
])
def test_tile_culling_is_invisible(dev, oracle, n, spread, rotated, W, H, S, renderer):
    """The culled tile kernel must equal the oracle bit for bit (uniform pigments on the spheres,
    checkered plane: no libm involved), including ties between duplicated shapes."""
    from pytracer_amd import flatten, hostmodel as hm

    world = _cluster_world(n, spread, seed=11 + n, rotated=rotated)
    if renderer == abi.RENDERER_POINTLIGHT:
        world.add_light(hm.PointLight(hm.Vec(-2.0, 3.0, 6.0), hm.Color(1.0, 0.9, 0.8), 0.0))
        world.add_light(hm.PointLight(hm.Vec(1.0, -4.0, 5.0), hm.Color(0.2, 0.3, 0.9), 2.0))
    scene = flatten.flatten_world(world)
    cam = flatten.flatten_camera(hm.PerspectiveCamera(1.0, W / H, hm.translation(hm.Vec(-1.0, 0.0, 1.0))))
    par = abi.make_params(W, H, renderer, samples_per_side=S, path_state=5, path_seq=77)
    ora, n_rays = oracle.render(scene, cam, par, sqr_mode=oracle.SQR_MUL)
    oracle.set_sqr_mode(oracle.SQR_POW)
    with dev.DeviceScene(scene) as ds:
        out = ds.render(cam, par)
        assert ds.stats().kernel in TILE_KERNELS, "expected a tile kernel"
        assert util.bits_equal(out, ora), f"max rel {util.rel_err(out, ora).max()}"
        assert ds.stats().n_rays == n_rays
        # and under an awkward row partition (blocks of 7 rows over 3 ranks), the usual one, and 16- / 32-row blocks
        # (where pixel-centre OnOff / Flat frames keep the 16x16-tile kernel: a tile's rows are consecutive image rows)
        for rb in (7, 8, 16, 32):
            got = np.zeros_like(out)
            for rank in range(3):
                p = abi.copy_params(par, n_ranks=3, rank=rank, row_block=rb)
                got[abi.rows_for_rank(H, rb, 3, rank)] = ds.render(cam, p)
                if S == 0 and n <= 254 and renderer in (abi.RENDERER_FLAT, abi.RENDERER_ONOFF):
                    assert ds.stats().kernel == (TILE4_KERNEL if rb % 16 == 0 or NO_CULL else abi.KERNEL_TILE), (rb, ds.stats().kernel)
            assert util.bits_equal(got, ora), rb


@pytest.mark.parametrize("n,renderer,S", [(120, abi.RENDERER_FLAT, 0), (300, abi.RENDERER_ONOFF, 2),
                                          (120, abi.RENDERER_POINTLIGHT, 0), (60, abi.RENDERER_FLAT, 3)])
def test_orthogonal_camera_beam_culling_is_invisible(dev, oracle, n, renderer, S):
    """Orthogonal camera: the tile's rays fill a beam; culling against it (spheres and planes) and the
    un-hoisted tile query must reproduce the oracle bit for bit, whatever the row partition."""
    from pytracer_amd import flatten, hostmodel as hm

    world = _cluster_world(n, 1.2, seed=31 + n, rotated=True)
    world.add_shape(hm.Plane(hm.translation(hm.Vec(6.0, 0.0, 0.0)) * hm.rotation_y(70.0),
                             hm.Material(hm.DiffuseBRDF(hm.CheckeredPigment(hm.Color(0.9, 0.1, 0.1), hm.Color(0.1, 0.1, 0.9), 3)))))
    if renderer == abi.RENDERER_POINTLIGHT:
        world.add_light(hm.PointLight(hm.Vec(-2.0, 3.0, 6.0), hm.Color(1.0, 0.9, 0.8), 0.0))
    scene = flatten.flatten_world(world)
    W, H = 144, 96
    camera = hm.OrthogonalCamera(W / H, hm.translation(hm.Vec(-1.0, 0.2, 1.0)) * hm.rotation_z(12.0) * hm.rotation_y(8.0) *
                                 hm.scaling(hm.Vec(1.0, 2.5, 2.0)))
    cam = flatten.flatten_camera(camera)
    par = abi.make_params(W, H, renderer, samples_per_side=S, path_state=9, path_seq=3)
    ora, n_rays = oracle.render(scene, cam, par, sqr_mode=oracle.SQR_MUL)
    oracle.set_sqr_mode(oracle.SQR_POW)
    with dev.DeviceScene(scene) as ds:
        out = ds.render(cam, par)
        assert ds.stats().kernel in TILE_KERNELS, "expected a tile kernel"
        assert util.bits_equal(out, ora), f"max rel {util.rel_err(out, ora).max()}"
        assert ds.stats().n_rays == n_rays
        got = np.zeros_like(out)
        for rank in range(3):
            p = abi.copy_params(par, n_ranks=3, rank=rank, row_block=8)
            got[abi.rows_for_rank(H, 8, 3, rank)] = ds.render(cam, p)
        assert util.bits_equal(got, ora)


def test_plain_kernels_without_culling(dev, oracle):
    """The `cull` switch off (PTRACE_CULL=0; settable in the running process since round 5 -- this test used to start a child
    process for it, and a fork of a process that holds a GPU context is something the suite can do without): every renderer
    through the one-lane-per-pixel kernels and the one-queue path tracer, both cameras, against the oracle."""
    from pytracer_amd import flatten, hostmodel as hm, scenes

    world = scenes.synthetic_world(24, with_plane=True)
    world.add_light(hm.PointLight(hm.Vec(-2.0, 3.0, 6.0), hm.Color(1.0, 0.9, 0.8), 0.0))
    scene = flatten.flatten_world(world)
    W, H = 72, 40
    saved = dev.get_tuning("cull")
    try:
        dev.set_tuning("cull", 0)
        for camera in (hm.PerspectiveCamera(1.0, W / H, hm.translation(hm.Vec(-1.0, 0.0, 1.0))),
                       hm.OrthogonalCamera(W / H, hm.translation(hm.Vec(-1.0, 0.0, 1.0)) * hm.scaling(hm.Vec(1.0, 3.0, 2.0)))):
            cam = flatten.flatten_camera(camera)
            with dev.DeviceScene(scene) as ds:
                for renderer, S in ((abi.RENDERER_ONOFF, 0), (abi.RENDERER_FLAT, 2), (abi.RENDERER_POINTLIGHT, 0)):
                    par = abi.make_params(W, H, renderer, samples_per_side=S)
                    ora, n_rays = oracle.render(scene, cam, par, sqr_mode=oracle.SQR_MUL)
                    out = ds.render(cam, par)
                    assert ds.stats().kernel == abi.KERNEL_SIMPLE, "culling should be off"
                    assert util.bits_equal(out, ora), (renderer, S)
                    assert ds.stats().n_rays == n_rays
                par = abi.make_params(W, H, abi.RENDERER_PATHTRACER, samples_per_side=2, num_of_rays=2, max_depth=3, rr_limit=2,
                                      path_state=45, path_seq=54)
                ora, _ = oracle.render(scene, cam, par, sqr_mode=oracle.SQR_MUL)
                out = ds.render(cam, par)
                assert np.all(util.rel_err(out, ora) <= 1e-5)
    finally:
        dev.set_tuning("cull", saved)
        oracle.set_sqr_mode(oracle.SQR_POW)


def _dome_world(dome, dome_material, n_small, seed, planes=()):
    """A dome around the camera, a few small spheres that leave most tiles to the dome alone, planes."""
    from pytracer_amd import hostmodel as hm

    g = hm.PCG(seed, 9)
    r = g.random_float
    w = hm.World()
    w.add_shape(hm.Sphere(dome, dome_material))
    for _ in range(n_small):
        rad = 0.05 + 0.1 * r()
        w.add_shape(hm.Sphere(hm.translation(hm.Vec(2.0 + 3.0 * r(), 4.0 * (r() - 0.5), 1.0 + 2.0 * (r() - 0.5))) *
                              hm.scaling(hm.Vec(rad, rad, rad)),
                              hm.Material(hm.DiffuseBRDF(hm.UniformPigment(hm.Color(r(), r(), r()))),
                                          hm.UniformPigment(hm.Color(0.1 * r(), 0.0, 0.0)))))
    for t in planes:
        w.add_shape(hm.Plane(t, hm.Material(hm.DiffuseBRDF(hm.CheckeredPigment(hm.Color(0.3, 0.5, 0.1), hm.Color(0.1, 0.2, 0.5), 2)))))
    return w


def _dome_cases():
    from pytracer_amd import hostmodel as hm

    sky = hm.Material(hm.DiffuseBRDF(hm.UniformPigment(hm.BLACK)), hm.UniformPigment(hm.Color(0.7, 0.5, 1.0)))
    lit = hm.Material(hm.DiffuseBRDF(hm.UniformPigment(hm.Color(0.4, 0.3, 0.2))), hm.UniformPigment(hm.Color(0.2, 0.3, 0.4)))
    chk = hm.Material(hm.DiffuseBRDF(hm.CheckeredPigment(hm.Color(0.9, 0.1, 0.1), hm.Color(0.1, 0.9, 0.1), 8)),
                      hm.UniformPigment(hm.BLACK))
    V, T, S_, RX, RZ = hm.Vec, hm.translation, hm.scaling, hm.rotation_x, hm.rotation_z
    # the camera sits at (-1, 0, 1): |o'|^2 in the dome's frame decides whether the shortcut may run
    return {
        "sky50": (S_(V(50.0, 50.0, 50.0)), sky, ()),                                   # |o'|^2 = 8e-4
        "ellipsoid": (T(V(0.5, -0.3, 0.8)) * RZ(25.0) * RX(40.0) * S_(V(9.0, 14.0, 7.0)), lit, ()),
        "near_wall_in": (T(V(-1.0 + 6.2, 0.0, 1.0)) * S_(V(10.0, 10.0, 10.0)), sky, ()),   # |o'|^2 = 0.384: shortcut
        "near_wall_out": (T(V(-1.0 + 7.5, 0.0, 1.0)) * S_(V(10.0, 10.0, 10.0)), sky, ()),  # |o'|^2 = 0.5625: no shortcut
        "checkered": (S_(V(30.0, 30.0, 30.0)), chk, ()),                                # needs (u, v): no shortcut
        "huge": (S_(V(1e7, 1e7, 1e7)), sky, ()),                                        # scale outside 1e-6..1e6
        "planes": (S_(V(50.0, 50.0, 50.0)), sky,
                   (T(V(0.0, 0.0, -0.25)), T(V(0.0, 0.0, 3.0)) * RX(180.0), T(V(8.0, 0.0, 0.0)) * RZ(20.0) * RX(80.0),
                    T(V(0.0, 0.0, 1.0)), T(V(0.0, 6.0, 0.0)) * RX(93.0))),                 # 4th: the camera lies ON it
    }


@pytest.mark.parametrize("case", ["sky50", "ellipsoid", "near_wall_in", "near_wall_out", "checkered", "huge", "planes"])
def test_dome_shortcut_and_plane_culling_are_invisible(dev, oracle, case):
    """Tiles whose only survivor is a dome around the camera skip ray generation; planes are culled per
    tile by the sign of d'.z at the tile corners.  Neither may change a bit, for any renderer."""
    from pytracer_amd import flatten, hostmodel as hm

    dome, material, planes = _dome_cases()[case]
    world = _dome_world(dome, material, 6, seed=5, planes=planes)
    scene = flatten.flatten_world(world)
    W, H = 136, 88
    cam = flatten.flatten_camera(hm.PerspectiveCamera(1.0, W / H, hm.translation(hm.Vec(-1.0, 0.0, 1.0))))
    cam_o = flatten.flatten_camera(hm.OrthogonalCamera(W / H, hm.translation(hm.Vec(-1.0, 0.0, 1.0)) * hm.rotation_z(10.0) *
                                                       hm.scaling(hm.Vec(1.0, 2.0, 1.5))))
    with dev.DeviceScene(scene) as ds:
        for renderer, S in ((abi.RENDERER_FLAT, 0), (abi.RENDERER_FLAT, 3), (abi.RENDERER_ONOFF, 2)):
            par = abi.make_params(W, H, renderer, samples_per_side=S, path_state=7, path_seq=11)
            for c in (cam, cam_o):  # (the orthogonal view: the dome test looks at the four corner origins)
                ora, n_rays = oracle.render(scene, c, par, sqr_mode=oracle.SQR_MUL)
                out = ds.render(c, par)
                assert ds.stats().kernel in TILE_KERNELS, "expected a tile kernel"
                assert util.bits_equal(out, ora), f"{case} renderer {renderer} S={S}: max rel {util.rel_err(out, ora).max()}"
                assert ds.stats().n_rays == n_rays
        if case != "checkered":  # (sin/cos/atan2 of the device differ from libm in the last bit: not bit-exact)
            par = abi.make_params(W, H, abi.RENDERER_PATHTRACER, samples_per_side=2, num_of_rays=2, max_depth=2,
                                  rr_limit=1, path_state=45, path_seq=54)
            ora, _ = oracle.render(scene, cam, par, sqr_mode=oracle.SQR_MUL)
            out = ds.render(cam, par)
            assert np.all(util.rel_err(out, ora) <= TOL)
    oracle.set_sqr_mode(oracle.SQR_POW)


@pytest.mark.parametrize("n_rays,depth,S,mode", [(1, 3, 4, abi.PCG_PIXEL), (2, 2, 2, abi.PCG_PIXEL),
                                                 (1, 5, 2, abi.PCG_SAMPLE), (3, 3, 0, abi.PCG_PIXEL),
                                                 # num_of_rays > 1: one pixel per wave, a node's children on lanes (pt_path_tree_kernel)
                                                 (10, 3, 1, abi.PCG_PIXEL),    # the CLI's defaults (main.py:95-102)
                                                 (10, 3, 2, abi.PCG_SAMPLE),
                                                 (4, 5, 2, abi.PCG_PIXEL),     # roulette (depth >= 3) inside families that still branch
                                                 (70, 1, 0, abi.PCG_PIXEL),    # more children than lanes: several rounds per node
                                                 (12, 2, 1, abi.PCG_PIXEL),    # leaf families larger than the ten rows of hypotheses
                                                 (2, 6, 3, abi.PCG_PIXEL)])
def test_c3_pathtracer_vs_oracle(dev, oracle, n_rays, depth, S, mode):
    scene, cam = _synthetic(32, False, False, 160, 90)
    par = abi.make_params(160, 90, abi.RENDERER_PATHTRACER, samples_per_side=S, num_of_rays=n_rays,
                          max_depth=depth, rr_limit=3, pcg_mode=mode, path_state=45, path_seq=54)
    with dev.DeviceScene(scene) as ds:
        out = ds.render(cam, par)
        st = ds.stats()
    assert st.kernel == (TREE_KERNEL if n_rays > 1 else REGIONS_KERNEL)
    ora, n = oracle.render(scene, cam, par, sqr_mode=oracle.SQR_MUL)
    oracle.set_sqr_mode(oracle.SQR_POW)
    err = util.rel_err(out, ora)
    bad = int((err > TOL).any(axis=-1).sum())
    print(f"path N={n_rays} D={depth} S={S}: max rel {err.max():.3e}, outliers {bad}, rays {st.n_rays} vs {n}")
    # sin/cos last-ulp differences can flip a silhouette/checker decision after a bounce: allow a
    # handful of outlier pixels, never a systematic difference
    assert bad <= 1  # (every logged run: 0; observed + 1)
    assert abs(int(st.n_rays) - n) <= 8


def test_pathtracer_many_spheres_vs_oracle(dev, oracle):
    """1 000 spheres: the scale+translate records no longer fit the LDS staging of the second pass (they
    are gathered from HBM/L2), the first pass culls on two levels; within 1e-5 of the oracle."""
    scene, cam = _synthetic(1000, False, True, 96, 54)
    par = abi.make_params(96, 54, abi.RENDERER_PATHTRACER, samples_per_side=2, num_of_rays=2, max_depth=2,
                          rr_limit=2, path_state=45, path_seq=54)
    with dev.DeviceScene(scene) as ds:
        out = ds.render(cam, par)
        n_dev = ds.stats().n_rays
    ora, n_rays = oracle.render(scene, cam, par, sqr_mode=oracle.SQR_MUL)
    oracle.set_sqr_mode(oracle.SQR_POW)
    err = util.rel_err(out, ora)
    assert err.max() <= TOL, f"max rel err {err.max():.3e}"
    assert abs(n_dev - n_rays) <= n_rays // 1000  # (a last-bit libm difference may flip a Russian-roulette draw)


@pytest.mark.parametrize("n,renderer,lights", [(2000, abi.RENDERER_PATHTRACER, 0), (1500, abi.RENDERER_POINTLIGHT, 2),
                                               (4000, abi.RENDERER_PATHTRACER, 0)])
def test_cell_walk_of_large_scenes_vs_oracle(dev, oracle, n, renderer, lights):
    """Scenes of >= 1024 bounded spheres: scattered rays (path tracer) and shadow rays (point lights) find their
    candidates by walking a uniform grid (world_query_lanes: 3D-DDA in fp32 with entered margins) instead of
    testing every bounding sphere.  A few outsized spheres (the dome, two big balls) stay outside the grid on the
    'always' list; a dense cluster puts many spheres into few cells.  Against the oracle: <= 1e-5, no outliers
    beyond the libm ones."""
    from pytracer_amd import flatten, scenes
    from pytracer_amd import hostmodel as hm

    world = scenes.synthetic_world(n, wide=True)
    g = hm.PCG(7, n)
    r = g.random_float
    for _ in range(2):  # outsized
        world.add_shape(hm.Sphere(hm.translation(hm.Vec(12.0 + 8 * r(), -6 + 12 * r(), 1.5)) * hm.scaling(hm.Vec(1.5, 1.5, 1.5)),
                                  hm.Material(hm.DiffuseBRDF(hm.UniformPigment(hm.Color(0.6, 0.7, 0.2))))))
    for _ in range(60):  # a dense cluster
        c = hm.Vec(6.0 + 0.4 * r(), 1.0 + 0.4 * r(), 0.6 + 0.4 * r())
        world.add_shape(hm.Sphere(hm.translation(c) * hm.scaling(hm.Vec(0.05, 0.05, 0.05)),
                                  hm.Material(hm.SpecularBRDF(hm.UniformPigment(hm.Color(0.8, 0.8, 0.8))) if r() < 0.3 else
                                              hm.DiffuseBRDF(hm.UniformPigment(hm.Color(0.2 + 0.6 * r(), 0.5, 0.4))))))
    for k in range(lights):
        world.add_light(hm.PointLight(hm.Vec(-3.0 + 20.0 * k, 6.0 - 14.0 * k, 9.0), hm.Color(1.0, 0.9, 0.8), 0.0))
    W, H = 112, 63
    scene = flatten.flatten_world(world)
    cam = flatten.flatten_camera(scenes.synthetic_camera(W, H))
    par = abi.make_params(W, H, renderer, samples_per_side=2, num_of_rays=2 if n < 4000 else 1, max_depth=3, rr_limit=2,
                          path_state=45, path_seq=54)
    with dev.DeviceScene(scene) as ds:
        out = ds.render(cam, par)
        n_dev = ds.stats().n_rays
    ora, n_rays = oracle.render(scene, cam, par, sqr_mode=oracle.SQR_MUL)
    oracle.set_sqr_mode(oracle.SQR_POW)
    err = util.rel_err(out, ora)
    bad = int((err > TOL).any(axis=-1).sum())
    print(f"grid n={n} renderer={renderer}: max rel {err.max():.3e}, outliers {bad}, rays {n_dev} vs {n_rays}")
    assert bad <= 1
    assert abs(int(n_dev) - n_rays) <= max(8, n_rays // 1000)


def _random_world(seed):
    """Random spheres (scale+translate and rotated ellipsoids, some enclosing the camera), tilted planes,
    mixed materials: a scene generator with no regard for what the culling code finds convenient."""
    from pytracer_amd import hostmodel as hm

    g = hm.PCG(1000 + seed, 17)
    r = g.random_float
    V = hm.Vec
    w = hm.World()

    def material(allow_pattern=True):
        kind = r()
        col = hm.Color(r(), r(), r())
        pig = hm.UniformPigment(col)
        if allow_pattern and kind < 0.2:
            pig = hm.CheckeredPigment(col, hm.Color(r(), r(), r()), 1 + int(6 * r()))
        brdf = hm.SpecularBRDF(pig) if r() < 0.2 else hm.DiffuseBRDF(pig)
        emit = hm.UniformPigment(hm.Color(0.5 * r(), 0.5 * r(), 0.5 * r()) if r() < 0.3 else hm.BLACK)
        return hm.Material(brdf, emit)

    if r() < 0.7:  # a dome: centred near the camera or not, black or not
        sc = 5.0 + 60.0 * r()
        t = hm.translation(V(6.0 * (r() - 0.5), 6.0 * (r() - 0.5), 3.0 * (r() - 0.5)))
        dome = t * hm.scaling(V(sc, sc * (0.6 + 0.8 * r()), sc))
        black = r() < 0.5
        w.add_shape(hm.Sphere(dome, hm.Material(hm.DiffuseBRDF(hm.UniformPigment(hm.BLACK if black else hm.Color(r(), r(), r()))),
                                                hm.UniformPigment(hm.Color(0.3 + r(), 0.3 + r(), 0.3 + r())))))
    n = 3 + int(r() * r() * 400)
    spread = 1.0 + 8.0 * r()
    for i in range(n):
        rad = 0.02 + 0.5 * r() * r()
        t = hm.translation(V(0.5 + 10.0 * r(), spread * (r() - 0.5), 1.0 + 0.5 * spread * (r() - 0.5)))
        if r() < 0.3:
            T = t * hm.rotation_z(360 * r()) * hm.rotation_y(360 * r()) * hm.scaling(V(rad, rad * (0.3 + 2 * r()), rad * (0.3 + 2 * r())))
        else:
            T = t * hm.scaling(V(rad, rad, rad))
        w.add_shape(hm.Sphere(T, material(allow_pattern=False)))
    for _ in range(int(3.5 * r())):
        T = hm.translation(V(4.0 * r(), 4.0 * (r() - 0.5), 3.0 * (r() - 0.6))) * hm.rotation_x(180.0 * r() * (r() < 0.5)) * \
            hm.rotation_y(60.0 * (r() - 0.5))
        w.add_shape(hm.Plane(T, material()))
    if r() < 0.5:
        w.add_light(hm.PointLight(V(-2.0 + 4 * r(), 6.0 * (r() - 0.5), 4.0 + 4 * r()), hm.Color(1.0, 0.9, 0.8), 3.0 * r() * (r() < 0.5)))
    cam_t = hm.translation(V(-1.0 - 2 * r(), r() - 0.5, 0.5 + r())) * hm.rotation_z(40.0 * (r() - 0.5)) * hm.rotation_y(30.0 * (r() - 0.5))
    W, H = 40 + 8 * int(12 * r()), 24 + 8 * int(8 * r())
    cam = hm.PerspectiveCamera(0.5 + 1.5 * r(), W / H, cam_t) if r() < 0.85 else hm.OrthogonalCamera(W / H, cam_t)
    return w, cam, W, H


@pytest.mark.parametrize("seed", range(int(os.environ.get("PT_FUZZ_SEEDS", "24"))))
def test_random_scenes_match_oracle(dev, oracle, seed):
    """Differential test over random scenes: OnOff / Flat / PointLight bit for bit (no libm on these
    scenes' paths except specular PointLight), PathTracer within 1e-5; ray counts equal."""
    from pytracer_amd import flatten

    world, camera, W, H = _random_world(seed)
    scene = flatten.flatten_world(world)
    cam = flatten.flatten_camera(camera)
    with dev.DeviceScene(scene) as ds:
        for renderer, S in ((abi.RENDERER_ONOFF, 0), (abi.RENDERER_FLAT, 0), (abi.RENDERER_FLAT, 2), (abi.RENDERER_POINTLIGHT, 0)):
            par = abi.make_params(W, H, renderer, samples_per_side=S, path_state=3 + seed, path_seq=21)
            ora, n_rays = oracle.render(scene, cam, par, sqr_mode=oracle.SQR_MUL)
            out = ds.render(cam, par)
            if _uses_libm(scene, par):
                assert np.all(util.rel_err(out, ora) <= TOL), f"seed {seed} renderer {renderer}"
            else:
                assert util.bits_equal(out, ora), f"seed {seed} renderer {renderer} S={S}: max rel {util.rel_err(out, ora).max()}"
                assert ds.stats().n_rays == n_rays
        # (PT_FUZZ_RAYS: more rays per hit than the default 1 - 3, for runs that force the num_of_rays > 1 kernels' hand-over)
        n_rays_fuzz = int(os.environ.get("PT_FUZZ_RAYS", "0")) or 1 + seed % 3
        par = abi.make_params(W, H, abi.RENDERER_PATHTRACER, samples_per_side=2, num_of_rays=n_rays_fuzz, max_depth=1 + seed % 4,
                              rr_limit=seed % 3, path_state=45 + seed, path_seq=54)
        ora, n_rays = oracle.render(scene, cam, par, sqr_mode=oracle.SQR_MUL)
        out = ds.render(cam, par)
        err = util.rel_err(out, ora)
        bad = int((err > TOL).any(axis=-1).sum())
        # a last-bit difference in sin/cos can flip a Russian-roulette decision or a checker cell for one pixel
        assert bad <= 1, f"seed {seed}: {bad} pixels off, max rel {err.max():.3e}"
    oracle.set_sqr_mode(oracle.SQR_POW)


def test_furnace(dev):
    """test_all.py:1015-1051 on the device: closed diffuse sphere, N=1, D=100, rr_limit=101."""
    g = util.load("g9_furnace")
    eye = [1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0]
    cam = abi.make_camera(abi.CAMERA_PERSPECTIVE, eye, 1.0, 1.0)
    for i, row in enumerate(g["rows"]):
        scene = abi.FlatScene.from_dict(g, prefix=f"f{i}_scene_")
        par = abi.make_params(8, 8, abi.RENDERER_PATHTRACER, num_of_rays=1, max_depth=100, rr_limit=101,
                              path_state=7, path_seq=100 * i)
        with dev.DeviceScene(scene) as ds:
            out = ds.render(cam, par)
            st = ds.stats()
        assert st.n_rays == 64 * 101
        assert np.allclose(out, row[5], rtol=1e-3)


# ---- properties at full size (1280x720) -------------------------------------------------------------------------
def test_full_size_partition_invariance_and_determinism(dev):
    scene, cam = _synthetic(32, True, False, 1280, 720)
    par = abi.make_params(1280, 720, abi.RENDERER_FLAT)
    with dev.DeviceScene(scene) as ds:
        full = ds.render(cam, par)
        again = ds.render(cam, par)
        assert util.bits_equal(full, again)
        assert ds.stats().n_rays == 1280 * 720
        for n_ranks, rb in ((2, 8), (8, 16), (3, 7)):
            got = np.zeros_like(full)
            for rank in range(n_ranks):
                p = abi.copy_params(par, n_ranks=n_ranks, rank=rank, row_block=rb)
                part = ds.render(cam, p)
                rows = abi.rows_for_rank(720, rb, n_ranks, rank)
                assert part.shape[0] == len(rows)
                got[rows] = part
            assert util.bits_equal(got, full)
        # OnOff == "Flat hit something": the sky sphere encloses the camera, so every pixel is lit
        onoff = ds.render(cam, abi.copy_params(par, renderer=abi.RENDERER_ONOFF))
        assert np.all(onoff == 1.0)
    assert math.isfinite(float(full.sum())) and float(full.min()) >= 0.0


def test_full_size_many_spheres_partition_invariance(dev):
    """C5 (10 000 spheres, two-level culling): the cells are cut in GLOBAL pixels, so every partition
    must give the frame of a single rank, bit for bit; fp32 output is the rounded fp64 output."""
    scene, cam = _synthetic(10000, False, True, 1280, 720)
    par = abi.make_params(1280, 720, abi.RENDERER_FLAT)
    with dev.DeviceScene(scene) as ds:
        full = ds.render(cam, par)
        assert ds.stats().n_rays == 1280 * 720
        for n_ranks, rb in ((2, 8), (8, 16), (5, 24), (3, 7)):  # (3, 7): blocks not multiples of 8 -> one-level culling
            got = np.zeros_like(full)
            n_rays = 0
            for rank in range(n_ranks):
                p = abi.copy_params(par, n_ranks=n_ranks, rank=rank, row_block=rb)
                got[abi.rows_for_rank(720, rb, n_ranks, rank)] = ds.render(cam, p)
                n_rays += ds.stats().n_rays
            assert util.bits_equal(got, full), f"{n_ranks} ranks, blocks of {rb} rows"
            assert n_rays == 1280 * 720
        onoff = ds.render(cam, abi.copy_params(par, renderer=abi.RENDERER_ONOFF))
        assert np.all(onoff == 1.0)
        f32 = ds.render(cam, abi.copy_params(par, out_format=abi.OUT_F32))
        assert np.array_equal(f32, full.astype(np.float32))


def test_full_size_pathtracer_partition_invariance(dev):
    scene, cam = _synthetic(32, False, False, 1280, 720)
    par = abi.make_params(1280, 720, abi.RENDERER_PATHTRACER, samples_per_side=2, num_of_rays=1, max_depth=3,
                          rr_limit=3, path_state=45, path_seq=54)
    with dev.DeviceScene(scene) as ds:
        full = ds.render(cam, par)
        n_full = ds.stats().n_rays
        got = np.zeros_like(full)
        n_parts = 0
        for rank in range(4):
            p = abi.copy_params(par, n_ranks=4, rank=rank, row_block=8)
            got[abi.rows_for_rank(720, 8, 4, rank)] = ds.render(cam, p)
            n_parts += ds.stats().n_rays
    assert util.bits_equal(got, full), "per-pixel seeds must make the image independent of the partition"
    assert n_parts == n_full


# ---- error behaviour of the C-ABI ----------------------------------------------------------------------------------
def test_errors(dev):
    from pytracer_amd._lib import PtraceError

    scene, cam = _synthetic(4, False, False, 16, 9)
    with dev.DeviceScene(scene) as ds:
        with pytest.raises(PtraceError) as e:
            ds.render(cam, abi.make_params(16, 9, abi.RENDERER_PATHTRACER, pcg_mode=abi.PCG_SEQ))
        assert e.value.code == -3
        with pytest.raises(PtraceError):
            ds.render(cam, abi.make_params(0, 9, abi.RENDERER_FLAT))
        with pytest.raises(PtraceError):
            ds.render(cam, abi.make_params(16, 9, abi.RENDERER_PATHTRACER, num_of_rays=0))
        with pytest.raises(PtraceError):
            ds.render(cam, abi.make_params(16, 9, 7))


def test_degenerate_sizes_and_parameters(dev, oracle):
    """Edges of the parameter space: an empty world, one-pixel and one-row frames, many samples, depth 0 and 1,
    Russian roulette from depth 0, more ranks than row blocks -- all against the oracle."""
    from pytracer_amd import flatten, hostmodel as hm, scenes

    cases = []
    empty = flatten.flatten_world(hm.World())
    cam_for = lambda w, h: flatten.flatten_camera(scenes.synthetic_camera(w, h))  # noqa: E731
    for renderer in (abi.RENDERER_ONOFF, abi.RENDERER_FLAT, abi.RENDERER_PATHTRACER, abi.RENDERER_POINTLIGHT):
        cases.append((empty, 24, 10, dict(renderer=renderer, samples_per_side=2)))
    c2 = flatten.flatten_world(scenes.synthetic_world(32, with_plane=True))
    for W, H in ((1, 1), (1, 37), (53, 1), (9, 8), (7, 9)):
        cases.append((c2, W, H, dict(renderer=abi.RENDERER_FLAT)))
        cases.append((c2, W, H, dict(renderer=abi.RENDERER_ONOFF, samples_per_side=3)))
    cases.append((c2, 40, 24, dict(renderer=abi.RENDERER_FLAT, samples_per_side=8)))
    two = flatten.flatten_world(scenes.synthetic_world(3))  # < 4 shapes: the one-lane-per-pixel kernel
    cases.append((two, 33, 17, dict(renderer=abi.RENDERER_FLAT, samples_per_side=2)))
    for scene, W, H, kw in cases:
        cam = cam_for(W, H)
        par = abi.make_params(W, H, background=(0.25, 0.5, 0.125), **kw)
        ora, n_rays = oracle.render(scene, cam, par, sqr_mode=oracle.SQR_MUL)
        with dev.DeviceScene(scene) as ds:
            out = ds.render(cam, par)
            assert util.bits_equal(out, ora), f"{W}x{H} {kw}"
            assert ds.stats().n_rays == n_rays
    # path tracer corners (tolerance: sin/cos)
    c3 = flatten.flatten_world(scenes.synthetic_world(32))
    cam = cam_for(48, 27)
    with dev.DeviceScene(c3) as ds:
        for kw in (dict(max_depth=0), dict(max_depth=1, num_of_rays=3), dict(max_depth=2, rr_limit=0, num_of_rays=2),
                   dict(max_depth=6, rr_limit=1, num_of_rays=1, samples_per_side=3), dict(max_depth=-1)):
            base = dict(renderer=abi.RENDERER_PATHTRACER, samples_per_side=2, num_of_rays=1, max_depth=3, rr_limit=3,
                        path_state=45, path_seq=54)
            base.update(kw)
            par = abi.make_params(48, 27, **base)
            ora, _ = oracle.render(c3, cam, par, sqr_mode=oracle.SQR_MUL)
            out = ds.render(cam, par)
            assert np.all(util.rel_err(out, ora) <= TOL), f"path tracer {kw}: {util.rel_err(out, ora).max():.3e}"
        # more ranks than 8-row blocks: some ranks own nothing
        par = abi.make_params(48, 27, abi.RENDERER_FLAT)
        full = ds.render(cam, par)
        got = np.zeros_like(full)
        for rank in range(6):
            rows = abi.rows_for_rank(27, 8, 6, rank)
            part = ds.render(cam, abi.copy_params(par, n_ranks=6, rank=rank, row_block=8))
            assert part.shape[0] == len(rows)
            if rows:
                got[rows] = part
        assert util.bits_equal(got, full)
    oracle.set_sqr_mode(oracle.SQR_POW)


# ---- the Python drop-in surface ----------------------------------------------------------------------------------------
def test_gpu_image_tracer_dropin(dev, oracle):
    from pytracer_amd import flatten, hostmodel as hm, scenes
    from pytracer_amd.tracer import GpuImageTracer

    world, camera = scenes.demo_world()
    image = hm.HdrImage(160, 120)
    calls = []
    tracer = GpuImageTracer(image, camera)
    assert tracer.fire_all_rays(hm.OnOffRenderer(world), callback=lambda col, row, **kw: calls.append((col, row, kw)),
                                tag=1) is None
    assert calls[0] == (0, 0, {"tag": 1}) and len(calls) >= 1
    gold = util.load("g5_demo_onoff_160x120")["pixels"]
    assert util.bits_equal(image.array, gold)
    assert image.get_pixel(3, 2).r == gold[2, 3, 0] and len(image.pixels) == 160 * 120
    tracer.fire_all_rays(hm.FlatRenderer(world))
    gold = util.load("g5_demo_flat_160x120")["pixels"]
    assert util.rel_err(image.array, gold).max() <= TOL
    # path tracer through the object interface == C-ABI with the same seeds
    pt = hm.PathTracer(world, pcg=hm.PCG(45, 54), num_of_rays=2, max_depth=2)
    small = hm.HdrImage(40, 30)
    GpuImageTracer(small, camera).fire_all_rays(pt)
    gold = util.load("g5_demo_path_40x30_n2d2_pixel")["pixels"]
    assert util.rel_err(small.array, gold).max() <= TOL
    # a func that is no renderer runs the host per-pixel loop, whatever ran before
    GpuImageTracer(small, camera).fire_all_rays(lambda ray: hm.Color(1.0, 2.0, 3.0))
    assert small.get_pixel(39, 29) == hm.Color(1.0, 2.0, 3.0)
    tracer.close()


def test_gpu_image_tracer_bands_callback_and_world_mutation(dev, oracle):
    """With a callback the frame is rendered in bands (progress for long frames): the image is bit-identical to
    the one-launch frame, and a callback_time_s of 0 reports every band but the last in ascending row order.
    The world is re-read on every call like the reference does: a shape added between two calls is rendered."""
    from pytracer_amd import hostmodel as hm, scenes
    from pytracer_amd.tracer import GpuImageTracer

    world = scenes.synthetic_world(32, with_plane=True)
    camera = scenes.synthetic_camera(160, 90)
    pt = lambda: hm.PathTracer(world, pcg=hm.PCG(45, 54), num_of_rays=1, max_depth=3)  # noqa: E731
    one, banded = hm.HdrImage(160, 90), hm.HdrImage(160, 90)
    t1 = GpuImageTracer(one, camera, samples_per_side=2)
    t1.fire_all_rays(pt())
    assert t1.last_bands == 1
    calls = []
    t2 = GpuImageTracer(banded, camera, samples_per_side=2)
    t2.fire_all_rays(pt(), callback=lambda col, row: calls.append((col, row)), callback_time_s=0.0)
    assert util.bits_equal(one.array, banded.array)
    assert t2.last_bands >= 2 and t2.last_stats.n_rays == t1.last_stats.n_rays and t2.last_stats.n_pixels == 160 * 90
    assert calls[0] == (0, 0) and len(calls) == t2.last_bands
    assert all(c == 159 for c, _ in calls[1:]) and [r for _, r in calls[1:]] == sorted(r for _, r in calls[1:])
    assert calls[-1][1] < 89
    # the statistics of a banded frame are sums over its bands, the resolved rays included (a subset count of n_rays)
    assert t2.last_stats.n_rays_resolved == t1.last_stats.n_rays_resolved <= t2.last_stats.n_rays
    # default callback_time_s: the frame is far quicker, so only the initial call is made -- in two launches, not four
    calls.clear()
    t2.fire_all_rays(pt(), callback=lambda col, row: calls.append((col, row)))
    assert calls == [(0, 0)] and t2.last_bands == 2
    # resident=True: the same frame, left in HBM (banded or not), nothing written into `image` until download()
    held = hm.HdrImage(160, 90)
    t4 = GpuImageTracer(held, camera, samples_per_side=2, resident=True)
    t4.fire_all_rays(pt(), callback=lambda col, row: None, callback_time_s=0.0)
    assert t4.last_bands >= 2 and not held.array.any()
    assert util.bits_equal(t4.device_image.numpy(), one.array)
    t4.fire_all_rays(pt())
    assert t4.last_bands == 1 and util.bits_equal(t4.device_image.numpy(), one.array)
    t4.download()
    assert util.bits_equal(held.array, one.array)
    t4.close()
    # mutate the SAME World object between two calls (a cache keyed on identity rendered the old scene)
    flat_before = hm.HdrImage(160, 90)
    t3 = GpuImageTracer(flat_before, camera)
    t3.fire_all_rays(hm.FlatRenderer(world))
    handle = t3._scene
    t3.fire_all_rays(hm.FlatRenderer(world))
    assert t3._scene is handle  # unchanged world: the device copy is reused
    world.add_shape(hm.Sphere(hm.translation(hm.Vec(2.0, 0.0, 1.0)) * hm.scaling(hm.Vec(0.5, 0.5, 0.5)),
                              hm.Material(hm.DiffuseBRDF(hm.UniformPigment(hm.Color(9.0, 8.0, 7.0))))))
    before = flat_before.array.copy()
    t3.fire_all_rays(hm.FlatRenderer(world))
    assert t3._scene is not handle and not np.array_equal(before, flat_before.array)
    assert flat_before.get_pixel(80, 45) == hm.Color(9.0, 8.0, 7.0)
    for t in (t1, t2, t3):
        t.close()


def test_c4_scale_partition_invariance(dev):
    """BASELINE.json config 4 at full resolution (3840x2160, 256 spheres, PathTracer depth 5), sharded the way
    the 8-GPU run shards it: the 8 ranks' row blocks, rendered one after the other on this GPU, must
    assemble to exactly the single-rank frame (spp reduced to 4 to keep the test short)."""
    scene, cam = _synthetic(256, False, True, 3840, 2160)
    par = abi.make_params(3840, 2160, abi.RENDERER_PATHTRACER, samples_per_side=2, num_of_rays=1, max_depth=5,
                          rr_limit=3, path_state=45, path_seq=54, out_format=abi.OUT_F32)
    with dev.DeviceScene(scene) as ds:
        full = ds.render(cam, par)
        n_full = ds.stats().n_rays
        got = np.zeros_like(full)
        n_parts = 0
        for rank in range(8):
            p = abi.copy_params(par, n_ranks=8, rank=rank, row_block=8)
            got[abi.rows_for_rank(2160, 8, 8, rank)] = ds.render(cam, p)
            n_parts += ds.stats().n_rays
    assert np.array_equal(got, full)
    assert n_parts == n_full and n_full >= 3840 * 2160 * 4
    assert np.isfinite(full).all() and float(full.min()) >= 0.0


# ---- BASELINE.json config 4 AS SPECIFIED: 256 "wide" spheres, PathTracer N=1 D=5 rr=3, S=8 (spp 64) ------------------
# (the reference's own pixels for this configuration: fixtures g5_c4_path_32x18_n1d5_s8_{pixel,sample}, run by
# test_frame_vs_reference_golden above)
def _c4_params(w, h, mode, **kw):
    return abi.make_params(w, h, abi.RENDERER_PATHTRACER, samples_per_side=8, num_of_rays=1, max_depth=5, rr_limit=3,
                           pcg_mode=mode, path_state=45, path_seq=54, **kw)


@pytest.mark.parametrize("mode", [abi.PCG_PIXEL, abi.PCG_SAMPLE])
def test_c4_as_specified_vs_oracle(dev, oracle, mode):
    """128x72 crop of the C4 view (same camera recipe, same scene, same renderer parameters, spp 64) against the
    oracle: <= 1e-5 relative per channel, outlier pixels counted (a last-ulp sin/cos difference may flip a
    silhouette or Russian-roulette decision after a bounce, H3), ray counts equal up to those flips."""
    W, H = 128, 72
    scene, cam = _synthetic(256, False, True, W, H)
    par = _c4_params(W, H, mode)
    with dev.DeviceScene(scene) as ds:
        out = ds.render(cam, par)
        st = ds.stats()
    ora, n = oracle.render(scene, cam, par, sqr_mode=oracle.SQR_MUL)
    oracle.set_sqr_mode(oracle.SQR_POW)
    err = util.rel_err(out, ora)
    bad = int((err > TOL).any(axis=-1).sum())
    print(f"C4 {W}x{H} mode={mode}: max rel {err.max():.3e}, outliers {bad}/{W * H}, rays {st.n_rays} vs {n}")
    assert bad <= 1
    assert abs(int(st.n_rays) - n) <= max(8, n // 100000)
    assert n >= W * H * 64


@pytest.mark.parametrize("mode", [abi.PCG_PIXEL, abi.PCG_SAMPLE])
def test_c4_as_specified_three_rank_seven_row_partition_vs_oracle(dev, oracle, mode):
    """The same frame cut in 7-row blocks over 3 ranks (blocks that are no multiple of the 8-row regions, a
    rank count that does not divide the block count): every rank's shard against the oracle's shard, and the
    assembled frame bit-identical to the single-rank render."""
    W, H = 128, 72
    scene, cam = _synthetic(256, False, True, W, H)
    par = _c4_params(W, H, mode)
    with dev.DeviceScene(scene) as ds:
        full = ds.render(cam, par)
        got = np.zeros_like(full)
        for rank in range(3):
            p = abi.copy_params(par, n_ranks=3, rank=rank, row_block=7)
            shard = ds.render(cam, p)
            rows = abi.rows_for_rank(H, 7, 3, rank)
            got[rows] = shard
            ora, _ = oracle.render(scene, cam, p, sqr_mode=oracle.SQR_MUL)
            err = util.rel_err(shard, ora)
            assert int((err > TOL).any(axis=-1).sum()) <= 1, f"rank {rank}: max rel {err.max():.3e}"
    oracle.set_sqr_mode(oracle.SQR_POW)
    assert util.bits_equal(got, full)


def test_c4_full_size_spp64_row_subset_partition_invariance(dev):
    """C4 at its full 3840x2160 frame and full spp 64, on a row subset: rows [1024, 1152) rendered as ONE
    128-row block (rank 8 of a 16-rank, 128-row-block partition) must equal, bit for bit, the same rows as
    they come out of the 8-GPU run's partition (8 ranks, interleaved 8-row blocks; two of the eight shards are
    rendered in full and cross-checked)."""
    W, H = 3840, 2160
    scene, cam = _synthetic(256, False, True, W, H)
    par = _c4_params(W, H, abi.PCG_PIXEL, out_format=abi.OUT_F32)
    with dev.DeviceScene(scene) as ds:
        pa = abi.copy_params(par, n_ranks=16, rank=8, row_block=128)
        rows_a = abi.rows_for_rank(H, 128, 16, 8)
        a = ds.render(cam, pa)
        n_a = ds.stats().n_rays
        assert rows_a[0] == 1024 and len(rows_a) == 128 and a.shape[0] == 128
        for rank in (0, 5):
            pb = abi.copy_params(par, n_ranks=8, rank=rank, row_block=8)
            rows_b = np.array(abi.rows_for_rank(H, 8, 8, rank))
            sel = (rows_b >= 1024) & (rows_b < 1152)
            assert sel.sum() == 16
            b = ds.render(cam, pb)
            assert np.array_equal(b[sel], a[rows_b[sel] - 1024]), f"rank {rank}"
    assert n_a >= 128 * W * 64 and np.isfinite(a).all() and float(a.min()) >= 0.0


# ---- statistics, streams and host buffers at the C-ABI --------------------------------------------------------------
def test_stats_follow_the_frame_just_rendered(dev):
    """pt_render_device(stream=NULL) -> pt_get_stats must report THIS frame's ray count, not the one
    before (the count leaves the device after the timing event): alternate two workloads and look every time."""
    import torch

    scene, cam = _synthetic(32, True, False, 320, 180)
    out = torch.empty((180, 320, 3), dtype=torch.float32, device="cuda")
    small = abi.make_params(160, 90, abi.RENDERER_FLAT, out_format=abi.OUT_F32)
    big = abi.make_params(320, 180, abi.RENDERER_FLAT, samples_per_side=2, out_format=abi.OUT_F32)
    cam_small = _synthetic(32, True, False, 160, 90)[1]
    with dev.DeviceScene(scene) as ds:
        for k in range(6):
            par, c, want = (small, cam_small, 160 * 90) if k % 2 == 0 else (big, cam, 320 * 180 * 4)
            ds.render_into(c, par, out.data_ptr(), out.numel() * 4, None)
            st = ds.stats()
            assert st.n_rays == want and st.n_pixels == par.width * par.height, (k, st.n_rays, want)
            assert st.vgprs > 0 and st.block == 256 and st.grid > 0 and st.kernel_ms > 0.0
        # the same on a caller-owned stream: stats() synchronises that stream's frame
        stream = torch.cuda.Stream()
        for k in range(4):
            par, c, want = (small, cam_small, 160 * 90) if k % 2 == 0 else (big, cam, 320 * 180 * 4)
            ds.render_into(c, par, out.data_ptr(), out.numel() * 4, stream.cuda_stream)
            assert ds.stats().n_rays == want
        ds.set_count_rays(False)
        ds.render_into(cam, big, out.data_ptr(), out.numel() * 4, None)
        assert ds.stats().n_rays == 0
        stream.synchronize()


def test_launches_on_alternating_streams_share_the_workspace_safely(dev, oracle):
    """the per-scene workspace (argument block, hoisted constants, region tables, LDS-backed frame
    stack) is shared by all launches.  Frames launched alternately on two caller streams, with different
    cameras and parameters, must each equal the frame rendered alone."""
    import torch

    scene, cam_a = _synthetic(32, False, False, 256, 144)
    cam_b = _synthetic(32, False, False, 192, 144)[1]  # another aspect ratio: other hoisted cones
    pa = abi.make_params(256, 144, abi.RENDERER_PATHTRACER, samples_per_side=2, num_of_rays=1, max_depth=3, rr_limit=3,
                         path_state=45, path_seq=54)
    pb = abi.make_params(192, 144, abi.RENDERER_PATHTRACER, samples_per_side=3, num_of_rays=2, max_depth=2, rr_limit=2,
                         path_state=7, path_seq=99)
    with dev.DeviceScene(scene) as ds:
        want_a, want_b = ds.render(cam_a, pa).copy(), ds.render(cam_b, pb).copy()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        outs = []
        for k in range(6):
            st, cam, par, shape = (s1, cam_a, pa, (144, 256, 3)) if k % 2 == 0 else (s2, cam_b, pb, (144, 192, 3))
            o = torch.empty(shape, dtype=torch.float64, device="cuda")
            ds.render_into(cam, par, o.data_ptr(), o.numel() * 8, st.cuda_stream)
            outs.append(o)
        torch.cuda.synchronize()
        for k, o in enumerate(outs):
            assert util.bits_equal(o.cpu().numpy(), want_a if k % 2 == 0 else want_b), k


def test_pinned_and_pageable_host_outputs_agree(dev):
    scene, cam = _synthetic(32, True, False, 320, 180)
    for fmt in (abi.OUT_F64, abi.OUT_F32):
        par = abi.make_params(320, 180, abi.RENDERER_FLAT, out_format=fmt)
        with dev.DeviceScene(scene) as ds:
            a = ds.render(cam, par, pinned=True)
            b = ds.render(cam, par, pinned=False)
            assert a.dtype == b.dtype and np.array_equal(a, b)
            keep = a.copy()
            del a
            c = ds.render(cam, par)  # the pool hands the same page-locked buffer out again
            assert np.array_equal(c, keep)



def test_queue_blocks_alternate_cleanly(dev):
    """The path tracer's queue block exists twice; a frame's path kernel zeroes the block of the NEXT frame (no
    memset per frame).  Path-traced frames of different sizes and modes, Flat frames, a refused call and an
    orthogonal-camera path frame (the one-queue kernel) in any order must each equal the frame rendered by a fresh
    scene, and an odd/even number of frames must not matter."""
    from pytracer_amd import flatten, hostmodel as hm
    from pytracer_amd._lib import PtraceError

    scene, cam = _synthetic(32, False, False, 256, 144)
    cam_small = _synthetic(32, False, False, 64, 40)[1]
    cam_ortho = flatten.flatten_camera(hm.OrthogonalCamera(96 / 64, hm.translation(hm.Vec(-1.0, 0.0, 1.0)) * hm.scaling(hm.Vec(1.0, 4.0, 3.0))))
    kw = dict(num_of_rays=1, max_depth=3, rr_limit=3, path_state=45, path_seq=54)
    jobs = {
        "pixel": (cam, abi.make_params(256, 144, abi.RENDERER_PATHTRACER, samples_per_side=4, **kw)),
        "sample": (cam, abi.make_params(256, 144, abi.RENDERER_PATHTRACER, samples_per_side=4, pcg_mode=abi.PCG_SAMPLE, **kw)),
        "small": (cam_small, abi.make_params(64, 40, abi.RENDERER_PATHTRACER, samples_per_side=2, **kw)),
        "n2": (cam_small, abi.make_params(64, 40, abi.RENDERER_PATHTRACER, samples_per_side=1, num_of_rays=2, max_depth=2,
                                          rr_limit=2, path_state=3, path_seq=5)),
        "ortho": (cam_ortho, abi.make_params(96, 64, abi.RENDERER_PATHTRACER, samples_per_side=2, **kw)),
        "flat": (cam, abi.make_params(256, 144, abi.RENDERER_FLAT)),
    }
    want = {}
    for name, (c, p) in jobs.items():
        with dev.DeviceScene(scene) as fresh:
            want[name] = fresh.render(c, p).copy()
    order = ["pixel", "sample", "flat", "small", "small", "ortho", "pixel", "n2", "flat", "sample", "ortho", "ortho", "small",
             "pixel", "pixel", "sample"]
    with dev.DeviceScene(scene) as ds:
        for k, name in enumerate(order):
            if k in (3, 9):  # a refused frame in between leaves the queue state alone
                with pytest.raises(PtraceError):
                    ds.render(cam, abi.make_params(256, 144, abi.RENDERER_PATHTRACER, num_of_rays=0))
            c, p = jobs[name]
            got = ds.render(c, p)
            assert util.bits_equal(got, want[name]), (k, name)


def test_tile_kernels_keep_their_occupancy(dev):
    """The C2-class kernels are issue-bound at five (Flat) / six (OnOff) waves per SIMD: a change that pushes them over
    96 / 80 registers costs a wave and ~8 % of the headline (it has happened through code shared with the path
    tracer's first pass).  pt_stats.vgprs is what hipFuncGetAttributes reports for the kernel just launched."""
    import torch

    scene, cam = _synthetic(32, True, False, 320, 180)
    out = torch.empty((180, 320, 3), dtype=torch.float32, device="cuda")
    with dev.DeviceScene(scene) as ds:
        for renderer, limit in ((abi.RENDERER_FLAT, 96), (abi.RENDERER_ONOFF, 80)):
            ds.render_into(cam, abi.make_params(320, 180, renderer, out_format=abi.OUT_F32), out.data_ptr(), out.numel() * 4, None)
            assert 0 < ds.stats().vgprs <= limit, (renderer, ds.stats().vgprs)


def test_frames_in_flight_are_the_frames_of_one_stream(dev):
    """pytracer_amd.pipeline.FramePipeline: an animation (the camera turns from frame to frame, as the reference's demo
    does with --angle-deg) with three frames in flight; every frame bit-identical to the one a single handle renders."""
    import torch

    from pytracer_amd import flatten, hostmodel as hm, scenes
    from pytracer_amd.pipeline import FramePipeline

    W, H = 320, 176
    flat = flatten.flatten_world(scenes.synthetic_world(32, with_plane=True))
    cams = [flatten.flatten_camera(hm.PerspectiveCamera(1.0, W / H, hm.rotation_z(7.0 * k) * hm.translation(hm.Vec(-1.0, 0.0, 1.0))))
            for k in range(9)]
    for renderer, kw in ((abi.RENDERER_FLAT, {}), (abi.RENDERER_PATHTRACER, dict(samples_per_side=2, num_of_rays=1, max_depth=3,
                                                                                 rr_limit=2, path_state=45, path_seq=54))):
        par = abi.make_params(W, H, renderer, out_format=abi.OUT_F32, **kw)
        outs = [torch.empty((H, W, 3), dtype=torch.float32, device="cuda") for _ in cams]
        with FramePipeline(flat, n_in_flight=3) as pipe:
            for cam, out in zip(cams, outs):
                pipe.submit(cam, par, out)
            pipe.wait()
        with dev.DeviceScene(flat) as ds:
            for k, cam in enumerate(cams):
                ref = ds.render(cam, abi.copy_params(par, out_format=abi.OUT_F64)).astype(np.float32)
                assert np.array_equal(outs[k].cpu().numpy().view(np.uint32), ref.view(np.uint32)), (renderer, k)
        assert not np.array_equal(outs[0].cpu().numpy(), outs[5].cpu().numpy())


def test_frames_in_flight_in_buffers_of_the_c_abi(dev):
    """The same pipeline with the frames in `pytracer_amd.devmem.DeviceBuffer`s (pt_device_alloc, ABI 1.5) instead of torch
    tensors: what a caller without a GPU framework uses.  Buffers download through pt_device_download, freeing twice is
    harmless, a freed buffer refuses to hand out its pointer, and a zero-byte buffer is a null pointer."""
    from pytracer_amd import flatten, hostmodel as hm, scenes
    from pytracer_amd.devmem import DeviceBuffer, Stream
    from pytracer_amd.pipeline import FramePipeline

    W, H = 320, 176
    flat = flatten.flatten_world(scenes.synthetic_world(32, with_plane=True))
    cams = [flatten.flatten_camera(hm.PerspectiveCamera(1.0, W / H, hm.rotation_z(7.0 * k) * hm.translation(hm.Vec(-1.0, 0.0, 1.0))))
            for k in range(5)]
    par = abi.make_params(W, H, abi.RENDERER_FLAT, out_format=abi.OUT_F64)
    outs = [DeviceBuffer((H, W, 3), np.float64) for _ in cams]
    with FramePipeline(flat, n_in_flight=2) as pipe:
        for cam, out in zip(cams, outs):
            pipe.submit(cam, par, out)
        pipe.wait()
        with pytest.raises(ValueError, match="DeviceBuffer or a contiguous CUDA tensor"):
            pipe.submit(cams[0], par, np.zeros((H, W, 3)))
    with dev.DeviceScene(flat) as ds:
        for k, cam in enumerate(cams):
            assert util.bits_equal(outs[k].numpy(), ds.render(cam, par)), k
    st = Stream()
    assert util.bits_equal(outs[2].numpy(st), outs[2].numpy())  # (ordered behind a stream of the C-ABI as well)
    st.close()
    st.close()
    outs[0].free()
    outs[0].free()
    with pytest.raises(RuntimeError, match="after free"):
        outs[0].data_ptr()
    empty = DeviceBuffer((0, W, 3), np.float32)
    assert empty.data_ptr() == 0 and empty.numpy().shape == (0, W, 3)


def test_cloned_handles_share_the_scene_and_outlive_the_first(dev):
    """pt_scene_clone: further handles on one uploaded scene (own per-camera constants and queues, shared tables) render
    the frames the first handle renders, for a world with a grid and cell lists as well, and keep working after the
    handle they were cloned from has been freed (the last handle of the family frees the tables)."""
    from pytracer_amd import flatten, hostmodel as hm, scenes

    for n_spheres, W, H in ((32, 320, 176), (1500, 160, 96)):
        flat = flatten.flatten_world(scenes.synthetic_world(n_spheres, with_plane=n_spheres < 100, wide=n_spheres > 100))
        cams = [flatten.flatten_camera(hm.PerspectiveCamera(1.0, W / H, hm.rotation_z(9.0 * k) * hm.translation(hm.Vec(-1.0, 0.0, 1.0))))
                for k in range(3)]
        pars = [abi.make_params(W, H, abi.RENDERER_FLAT),
                abi.make_params(W, H, abi.RENDERER_PATHTRACER, samples_per_side=2, num_of_rays=1, max_depth=3, rr_limit=2,
                                path_state=45, path_seq=54)]
        first = dev.DeviceScene(flat)
        ref = [[first.render(cam, par) for cam in cams] for par in pars]
        clones = [first.clone(), first.clone()]
        grand = clones[0].clone()  # (a clone of a clone is a handle like the others)
        first.close()
        for h in clones + [grand]:
            for pi, par in enumerate(pars):
                for ci in (2, 0, 1):  # (another camera order than the first handle's: the per-camera constants are the handle's own)
                    out = h.render(cams[ci], par)
                    assert util.bits_equal(out, ref[pi][ci]), (n_spheres, pi, ci)
            assert h.stats().n_rays > 0
        for h in (clones[1], grand, clones[0]):
            h.close()
