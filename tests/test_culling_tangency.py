"""Adversarial culling tests.

PART 1 -- the predicate on its own (`test_cull_predicate_*`): the device evaluates `tile_cone` / `pixel_cone` /
`cone_keeps` for an image rectangle exactly as the render kernels do (`pt_debug_cull_probe`), on spheres built to
touch the rectangle's CORNER rays -- the rays the circular cones are tangent to -- by 1 ... 1e9 ulp of the radius;
whether a sphere really touches its ray is decided in 60-digit decimal arithmetic from the numbers the device was
given.  Every sphere that touches MUST be kept.  Mutation check (round 2, profiles/r02_culling_mutants.log): with the
slack constants of `cone_keeps` / `tile_cone` / `pixel_cone` set to zero the far-camera case loses 27 % of the
touching spheres; with the inflation of the bounding radii at upload removed as well, all three cameras lose
20-47 % -- and these tests fail: they do sit on the slack.

PART 2 -- whole frames: spheres at tangency +- a few ulp of rays that run ON the
boundary of a tile / strip / cell / pixel cone (perspective camera) or of a tile's beam (orthogonal camera).

Every conservative cull of the kernels (`cone_keeps`, `pixel_cone`, the strip and cell pre-passes, the beam of an
orthogonal camera; csrc/pt_tile.h) works in fp32 with hand-derived slack.  Random scenes exercise the slack only
statistically; here the geometry is built to sit on it:

* the per-pixel random streams are searched (vectorised PCG, numpy) for a seed whose FIRST jitter number of a
  chosen pixel is within 1e-7 of 0 or of 1 -- with samples_per_side = 1 that pixel's only primary ray then
  crosses the image plane within 1e-7 pixel of the pixel's left or right edge, and the pixel is chosen on the
  edge of an 8x8 tile (of a 32-pixel strip and cell, too): the ray runs along the lateral surface of all those
  cones at once, closer to it than fp32 resolves;
* the ray itself is taken from the CPU oracle (`pto_tracer_fire_ray`, the restatement of
  `ImageTracer.fire_ray`), and spheres are placed OUTSIDE the tile, tangent to that ray to within
  delta = 0, +-1, +-4, +-64, +-1e3, +-1e6, +-1e9 ulp of their radius (negative: the ray enters the sphere by that much),
  for radii from 1e-3 to 8 at distances from 0.5 to 300;
* the frame must equal the oracle's BIT FOR BIT (Flat: the sphere's colour flips a pixel when a hit is lost;
  OnOff likewise), in scenes of 2, 70 (strips) and 300 shapes (cells), under a dome (pixel classification of the
  path tracer's first pass: the flagged pixel must still be path-traced), and for both cameras.

A sphere culled although the boundary ray hits it shows up as a differing pixel.
"""
import numpy as np
import pytest

from pytracer_amd import abi
from tests import util


M = np.uint64(6364136223846793005)


def first_two_floats(init_state: int, seqs: np.ndarray):
    """Vectorised pcg.py:29-62: the first two random_float() of PCG(init_state, seq) for an array of seqs."""
    with np.errstate(over="ignore"):
        inc = (seqs.astype(np.uint64) << np.uint64(1)) | np.uint64(1)
        state = np.zeros_like(inc)
        state = state * M + inc
        state = state + np.uint64(init_state)
        state = state * M + inc
        outs = []
        for _ in range(2):
            old = state
            state = old * M + inc
            xs = (((old >> np.uint64(18)) ^ old) >> np.uint64(27)).astype(np.uint32)
            rot = (old >> np.uint64(59)).astype(np.uint32)
            out = (xs >> rot) | (xs << ((np.uint32(0) - rot) & np.uint32(31)))
            outs.append(out.astype(np.float64) / 4294967295.0)
    return outs


def find_seed(want_high: bool, eps=1e-7, init_state=45, span=1 << 22, start=1000):
    """A sequence number whose first jitter number is within eps of 0 (or of 1)."""
    for chunk in range(512):
        seqs = np.arange(start + chunk * span, start + (chunk + 1) * span, dtype=np.uint64)
        u, v = first_two_floats(init_state, seqs)
        hit = np.nonzero((u > 1.0 - eps) if want_high else (u < eps))[0]
        if hit.size:
            k = int(hit[0])
            return int(seqs[k]), float(u[k]), float(v[k])
    raise AssertionError("no such seed found")


@pytest.fixture(scope="module")
def seeds():
    return {False: find_seed(False), True: find_seed(True)}


def _world(extra, n_fill, dome):
    """`extra`: the adversarial spheres [(centre, radius)]; fillers far behind the camera bring the scene to the
    size that switches strips (> 64 shapes) or cells (> 256) on; `dome`: a sky sphere around everything."""
    from pytracer_amd import hostmodel as hm

    w = hm.World()
    if dome:
        w.add_shape(hm.Sphere(hm.scaling(hm.Vec(900.0, 900.0, 900.0)),
                              hm.Material(hm.DiffuseBRDF(hm.UniformPigment(hm.Color(0.0, 0.0, 0.0))),
                                          hm.UniformPigment(hm.Color(0.25, 0.5, 1.0)))))
    for k, (c, r) in enumerate(extra):
        col = hm.Color(0.2 + 0.1 * (k % 7), 0.9 - 0.1 * (k % 5), 0.3 + 0.05 * (k % 11))
        w.add_shape(hm.Sphere(hm.translation(hm.Vec(*c)) * hm.scaling(hm.Vec(r, r, r)),
                              hm.Material(hm.DiffuseBRDF(hm.UniformPigment(col)), hm.UniformPigment(hm.Color(0.0, 0.0, 0.0)))))
    for k in range(n_fill):
        w.add_shape(hm.Sphere(hm.translation(hm.Vec(-50.0 - k, 3.0 * (k % 9) - 12.0, 2.0 * (k % 5) - 4.0)) *
                              hm.scaling(hm.Vec(0.3, 0.3, 0.3)), hm.Material()))
    return w


def _boundary_ray(oracle, cam, W, H, col, row, u, v):
    out = oracle.tracer_fire_ray(cam, W, H, col, row, u, v)  # ImageTracer.fire_ray restated (imagetracer.py:48-58)
    return out[:3].copy(), out[3:6].copy()


def _tangent_spheres(o, d, outward, radii_dists, deltas_ulp):
    """Spheres whose surface is (1 + delta) radii away from the line o + t d at parameter t = dist / |d|, on the
    side `outward` (made perpendicular to d here)."""
    dn = d / np.linalg.norm(d)
    n = outward - np.dot(outward, dn) * dn
    n /= np.linalg.norm(n)
    out = []
    for (r, dist), k in zip(radii_dists, deltas_ulp):
        delta = k * np.spacing(r)
        out.append((tuple(o + dist * dn + (r + delta) * n), r))
    return out


DELTAS = [0, 1, -1, 4, -4, 64, -64, 1e3, -1e3, 1e6, -1e6, 1e9, -1e9]
RADII_DISTS = [(1e-3, 0.5), (0.01, 1.0), (0.05, 2.0), (0.3, 5.0), (1.0, 12.0), (2.5, 40.0), (8.0, 300.0)]


@pytest.mark.gpu
@pytest.mark.parametrize("camera", ["perspective", "orthogonal"])
@pytest.mark.parametrize("high", [False, True])
@pytest.mark.parametrize("n_fill,dome,renderer", [(0, False, abi.RENDERER_FLAT), (70, False, abi.RENDERER_ONOFF),
                                                  (300, False, abi.RENDERER_FLAT), (70, True, abi.RENDERER_PATHTRACER)])
def test_spheres_tangent_to_rays_on_cone_boundaries(oracle, seeds, camera, high, n_fill, dome, renderer):
    from pytracer_amd import device, flatten
    from pytracer_amd import hostmodel as hm

    W, H = 96, 64
    seq, u, v = seeds[high]
    assert (u > 1.0 - 1e-7) if high else (u < 1e-7)
    if camera == "perspective":
        camobj = hm.PerspectiveCamera(screen_distance=1.3, aspect_ratio=W / H,
                                      transformation=hm.rotation_z(17.0) * hm.translation(hm.Vec(-1.0, 0.2, 0.4)))
    else:
        camobj = hm.OrthogonalCamera(aspect_ratio=W / H, transformation=hm.translation(hm.Vec(-2.0, 0.1, 0.3)) *
                                     hm.scaling(hm.Vec(1.0, 4.0, 3.0)))
    cam = flatten.flatten_camera(camobj)
    # the pixel whose left (u ~ 0) or right (u ~ 1) edge is a tile, strip AND cell boundary: column 32 or 31
    col = 31 if high else 32
    worst = 0
    for row in (16, 40):
        gpix = row * W + col
        q0 = seq - gpix  # this pixel's generator is PCG(45, q0 + gpix) = the one found
        assert q0 > 0
        o, d = _boundary_ray(oracle, cam, W, H, col, row, u, v)
        # "outward": towards the neighbouring tile, i.e. along the image's x axis away from this pixel's tile
        o2, d2 = _boundary_ray(oracle, cam, W, H, col + (1 if high else -1), row, 0.5, 0.5)
        outward = (o2 + d2) - (o + d)
        spheres = []
        for shift in range(0, len(DELTAS), len(RADII_DISTS)):
            spheres += _tangent_spheres(o, d, outward, RADII_DISTS, (DELTAS + DELTAS)[shift:shift + len(RADII_DISTS)])
        for one in spheres:  # one adversarial sphere per frame: nothing else can hide a lost hit
            scene = flatten.flatten_world(_world([one], n_fill, dome))
            par = abi.make_params(W, H, renderer, samples_per_side=1, num_of_rays=1, max_depth=2, rr_limit=3,
                                  background=(0.01, 0.02, 0.03), path_state=45, path_seq=q0, jitter_state=45, jitter_seq=q0)
            with device.DeviceScene(scene) as ds:
                got = ds.render(cam, par)
            want, _ = oracle.render(scene, cam, par, sqr_mode=oracle.SQR_MUL)
            if renderer == abi.RENDERER_PATHTRACER:
                bad = int((util.rel_err(got, want) > 1e-5).any(axis=-1).sum())
                worst = max(worst, bad)
                assert bad <= 1, f"{camera} row {row} sphere {one}: {bad} pixels differ"
                # the boundary pixel itself: identical decision (hit or miss) on both sides
                assert util.rel_err(got[row, col], want[row, col]).max() <= 1e-5
            else:
                assert util.bits_equal(got, want), f"{camera} row {row} sphere {one}: device != oracle"
    oracle.set_sqr_mode(oracle.SQR_POW)


# ---- PART 1: the culling predicate against exact geometry --------------------------------------------------------------
def _dec_vec(v):
    from decimal import Decimal

    return [Decimal(float(x)) for x in v]


def _penetration(o, d, c, r):
    """r - distance(c, line o + t d) in 60-digit arithmetic (all inputs taken as the exact doubles they are)."""
    from decimal import Decimal, getcontext

    getcontext().prec = 60
    o, d, c = _dec_vec(o), _dec_vec(d), _dec_vec(c)
    w = [c[i] - o[i] for i in range(3)]
    cr = [w[1] * d[2] - w[2] * d[1], w[2] * d[0] - w[0] * d[2], w[0] * d[1] - w[1] * d[0]]
    dist = (sum(x * x for x in cr) / sum(x * x for x in d)).sqrt()
    along = sum(w[i] * d[i] for i in range(3))
    return float(Decimal(float(r)) - dist), float(along)


def _camera_rays(camobj, W, H):
    """-> fire(x, y): the primary ray through image position (x, y) in pixels (camera.py + imagetracer.py:56-58)."""
    from pytracer_amd import flatten

    cam = flatten.flatten_camera(camobj)
    m = np.array(list(cam.m)).reshape(3, 4)

    def fire(x, y):
        u, v = x / W, 1.0 - y / H
        if cam.kind == abi.CAMERA_PERSPECTIVE:
            o = np.array([-cam.screen_distance, 0.0, 0.0])
            d = np.array([cam.screen_distance, (1.0 - 2 * u) * cam.aspect_ratio, 2 * v - 1])
        else:
            o = np.array([-1.0, (1.0 - 2 * u) * cam.aspect_ratio, 2 * v - 1])
            d = np.array([1.0, 0.0, 0.0])
        return m[:, :3] @ o + m[:, 3], m[:, :3] @ d

    return cam, fire


PROBE_DELTAS = [-1, -4, -64, -1e3, -1e5, -1e7, -1e9, -3e10]
PROBE_RT = [(1e-3, 0.4), (0.02, 1.5), (0.3, 4.0), (1.0, 25.0), (5.0, 120.0), (40.0, 900.0), (0.05, 60.0)]


def _probe_case(dev, camobj, W, H, rect, pixel=None):
    """Spheres touching the four corner rays of `rect` (x0, x1, row0, row1) -- or of `pixel` -- from outside."""
    from pytracer_amd import flatten
    from pytracer_amd import hostmodel as hm

    cam, fire = _camera_rays(camobj, W, H)
    x0, x1, r0, r1 = rect
    if pixel is not None:
        cx0, cx1, cy0, cy1 = pixel[0], pixel[0] + 1, pixel[1], pixel[1] + 1
    else:
        cx0, cx1, cy0, cy1 = x0, x1, r0, r1 + 1
    oc, dc = fire(0.5 * (cx0 + cx1), 0.5 * (cy0 + cy1))
    world = hm.World()
    truth = []
    for (x, y) in ((cx0, cy0), (cx1, cy0), (cx0, cy1), (cx1, cy1)):
        o, d = fire(float(x), float(y))
        dn = d / np.linalg.norm(d)
        away = (o + d) - (oc + dc)  # from the rectangle's centre ray to this corner ray: outward
        n = away - np.dot(away, dn) * dn
        n /= np.linalg.norm(n)
        for r, t in PROBE_RT:
            for k in PROBE_DELTAS:
                c = o + t * dn + (r + k * np.spacing(r)) * n
                world.add_shape(hm.Sphere(hm.translation(hm.Vec(*c)) * hm.scaling(hm.Vec(r, r, r)), hm.Material()))
                truth.append((o, d, c, r))
    flat = flatten.flatten_world(world)
    with dev.DeviceScene(flat) as ds:
        keep = ds.cull_probe(cam, W, H, x0, x1, r0, r1, pixel)
    touching = lost = 0
    for i, (o, d, c_built, r) in enumerate(truth):
        c = flat.m[[3, 7, 11], i]  # the centre the device was given (the translation column of the shape's matrix)
        pen, along = _penetration(o, d, c, flat.m[0, i])
        if pen > 0.0 and along > 0.0:
            touching += 1
            lost += 0 if keep[i] else 1
    return touching, lost


@pytest.mark.gpu
@pytest.mark.parametrize("camera", ["perspective", "perspective_far", "orthogonal"])
def test_cull_predicate_keeps_every_sphere_that_touches_a_corner_ray(camera):
    from pytracer_amd import device as dev
    from pytracer_amd import hostmodel as hm

    if camera == "perspective":
        camobj, (W, H) = hm.PerspectiveCamera(1.3, 1.5, hm.rotation_z(17.0) * hm.translation(hm.Vec(-1.0, 0.2, 0.4))), (96, 64)
    elif camera == "perspective_far":  # large coordinates: the absolute error of C - O in fp32 matters
        camobj, (W, H) = hm.PerspectiveCamera(1.0, 3840 / 2160, hm.translation(hm.Vec(-700.0, 350.0, 90.0)) *
                                              hm.rotation_y(-8.0)), (3840, 2160)
    else:
        camobj, (W, H) = hm.OrthogonalCamera(1.5, hm.translation(hm.Vec(-2.0, 0.1, 0.3)) * hm.scaling(hm.Vec(1.0, 4.0, 3.0))), (96, 64)
    rects = [(0, 8, 0, 7), (W - 8, W, H - 8, H - 1), (32, 40, 24, 31), (32, 64, 24, 31), (32, 64, 32, 63)]  # tiles, a strip, a cell
    total = 0
    for rect in rects:
        touching, lost = _probe_case(dev, camobj, W, H, rect)
        total += touching
        assert lost == 0, f"{camera} rect {rect}: {lost} of {touching} touching spheres culled"
    assert total > 500  # (the construction does produce touching spheres: rounding leaves most of the negative deltas touching)
    if camera != "orthogonal":
        for rect, pixel in (((32, 40, 24, 31), (32, 24)), ((32, 40, 24, 31), (39, 31)), ((0, 8, 0, 7), (0, 0)), ((W - 8, W, H - 8, H - 1), (W - 1, H - 1))):
            touching, lost = _probe_case(dev, camobj, W, H, rect, pixel)
            assert touching > 50 and lost == 0, f"{camera} pixel {pixel}: {lost} of {touching} touching spheres culled"


def test_seed_search_reproduces_the_host_pcg():
    """The vectorised generator used for the search is the reference's (pcg.py), checked against the host copy."""
    from pytracer_amd.hostmodel import PCG

    seqs = np.array([54, 55, 1000, 2 ** 40 + 7], dtype=np.uint64)
    u, v = first_two_floats(45, seqs)
    for k, s in enumerate(seqs):
        g = PCG(45, int(s))
        assert g.random_float() == u[k] and g.random_float() == v[k]
