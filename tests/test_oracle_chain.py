"""The chain of evidence, closed at FULL size (VERDICT r4 missing #4 / next 5a).

    reference == oracle(pow)     bit for bit, on every fixture the reference itself produced   (tests/test_oracle_golden.py)
    device    == oracle(x*x)     bit for bit, on the bench's own frames at full size           (tests/test_gpu_fullsize.py)

The two halves meet only if the oracle's two arithmetic modes agree ON THOSE FRAMES: CPython evaluates ``x**2`` in
``Vec.squared_norm`` (geometry.py:122-128) with libm ``pow``, the device multiplies -- 1 ulp apart for 0.08 % of inputs
(SURVEY.md H2), which can move a pixel only where a discriminant or a normalisation sits within rounding of a decision.
Here: the oracle in its ``pow`` mode (the mode pinned to the reference) against its ``x*x`` mode (the mode the device is
compared with) on BASELINE.json's configurations at 1280x720 -- 0 differing pixels, equal ray counts -- so "device ==
oracle(x*x)" there does imply "device == what the reference computes".  north_star's "bit-exact for OnOffRenderer" at
1280x720 rests on the OnOff case.  No GPU; CPU only; the oracle is the thing under test.
"""
import numpy as np
import pytest

from pytracer_amd import abi, flatten, scenes

W, H = 1280, 720
C3 = dict(renderer=abi.RENDERER_PATHTRACER, samples_per_side=4, num_of_rays=1, max_depth=3, rr_limit=3, path_state=45, path_seq=54)


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as orc

    orc.build()
    yield orc
    orc.set_sqr_mode(orc.SQR_POW)


def _both(oracle, scene, cam, par):
    a, na = oracle.render(scene, cam, par, sqr_mode=oracle.SQR_POW)
    b, nb = oracle.render(scene, cam, par, sqr_mode=oracle.SQR_MUL)
    oracle.set_sqr_mode(oracle.SQR_POW)
    differing = int((np.ascontiguousarray(a).view(np.uint64) != np.ascontiguousarray(b).view(np.uint64)).any(axis=-1).sum())
    return differing, na, nb


CASES = {
    "C2 Flat": (32, True, False, dict(renderer=abi.RENDERER_FLAT)),
    "C2 OnOff": (32, True, False, dict(renderer=abi.RENDERER_ONOFF)),
    "C3 PIXEL": (32, False, False, dict(C3, pcg_mode=abi.PCG_PIXEL)),
    "C3 SAMPLE": (32, False, False, dict(C3, pcg_mode=abi.PCG_SAMPLE)),
    "C3 N=10 (the CLI's defaults)": (32, False, False, dict(C3, samples_per_side=1, num_of_rays=10)),
}


@pytest.mark.parametrize("name", list(CASES))
def test_pow_and_mul_modes_render_the_same_full_size_frame(oracle, name):
    n, plane, wide, kw = CASES[name]
    scene = flatten.flatten_world(scenes.synthetic_world(n, with_plane=plane, wide=wide))
    cam = flatten.flatten_camera(scenes.synthetic_camera(W, H))
    differing, n_pow, n_mul = _both(oracle, scene, cam, abi.make_params(W, H, **kw))
    assert differing == 0 and n_pow == n_mul, f"{name}: {differing} pixels differ between pow and x*x, rays {n_pow} vs {n_mul}"


@pytest.mark.slow
def test_pow_and_mul_modes_render_the_same_C5_frame(oracle):
    """C5: 10 000 spheres, Flat, 1280x720 -- 9.2e9 ray-sphere tests per mode (about a minute on 8 cores)."""
    scene = flatten.flatten_world(scenes.synthetic_world(10000, wide=True))
    cam = flatten.flatten_camera(scenes.synthetic_camera(W, H))
    differing, n_pow, n_mul = _both(oracle, scene, cam, abi.make_params(W, H, abi.RENDERER_FLAT))
    assert differing == 0 and n_pow == n_mul


def test_oracle_ray_image_adds_up_to_the_ray_count(oracle):
    """pto_set_ray_image (diagnostics for tools/ray_histogram.py: rays per pixel of a frame): the counts it writes add up to
    what pto_render returns, a pixel whose primary ray scatters nothing counts 1, and nothing of it changes the frame."""
    import ctypes as C

    W, H = 96, 54
    flat = flatten.flatten_world(scenes.synthetic_world(32, with_plane=True))
    cam = flatten.flatten_camera(scenes.synthetic_camera(W, H))
    par = abi.make_params(W, H, abi.RENDERER_PATHTRACER, samples_per_side=1, num_of_rays=4, max_depth=3, rr_limit=3,
                          path_state=45, path_seq=54)
    plain, n0 = oracle.render(flat, cam, par, sqr_mode=oracle.SQR_MUL)
    img = np.zeros((H, W), dtype=np.uint32)
    oracle.lib().pto_set_ray_image.argtypes = [C.c_void_p]
    oracle.lib().pto_set_ray_image(img.ctypes.data_as(C.c_void_p))
    try:
        out, n = oracle.render(flat, cam, par, sqr_mode=oracle.SQR_MUL)
    finally:
        oracle.lib().pto_set_ray_image(None)
        oracle.set_sqr_mode(oracle.SQR_POW)
    assert n == n0 and int(img.sum()) == n and img.min() >= 1 and img.max() <= 1 + 4 + 16 + 64
    assert np.array_equal(out.view(np.uint8), plain.view(np.uint8))
