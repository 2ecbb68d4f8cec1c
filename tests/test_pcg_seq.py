"""PT_PCG_SEQ -- the reference's OWN jitter stream on the device.

``ImageTracer.fire_all_rays`` draws exactly two numbers per sample from ONE sequential generator (imagetracer.py:84-101),
so for the renderers without a scattering stream sample k of pixel i starts 2 (i S^2 + k) draws in, which a linear
congruential generator reaches by jump-ahead.  CPU part: the host restatement of the jump against single steps, the
seeds solved back from a generator that has been drawn from, the new reference-held fixtures against the oracle.  GPU
part: the device against the oracle's SERIAL loop on frames large enough for 64-bit distances to matter, under
partitions, and the drop-in's ``pcg`` left where the reference would leave it."""
import numpy as np
import pytest

from pytracer_amd import abi, flatten, hostmodel as hm, scenes
from tests import util


def test_host_jump_ahead_equals_single_steps():
    for seeds in ((42, 54), (7, 11), (123456789, 2 ** 40 + 3)):
        g = hm.PCG(*seeds)
        s0, inc = g.state, g.inc
        for n in (0, 1, 2, 3, 17, 64, 1000, 4097):
            h = hm.PCG(*seeds)
            for _ in range(n):
                h.random()
            assert hm.pcg_advance(s0, inc, n) == h.state
        # composition: a + b draws == a draws, then b more (distances beyond 2^32 included)
        for a, b in ((5, 2 ** 33 + 1), (2 ** 40, 2 ** 41 + 12345), (2 ** 62, 3)):
            assert hm.pcg_advance(hm.pcg_advance(s0, inc, a), inc, b) == hm.pcg_advance(s0, inc, a + b)


def test_seeds_are_recovered_from_a_generator_that_has_been_drawn_from():
    g = hm.PCG(42, 54)
    assert flatten.recover_seeds(g) == (42, 54)
    for _ in range(11):
        g.random()
    s, q = flatten.recover_seeds(g)
    h = hm.PCG(s, q)
    assert (h.state, h.inc) == (g.state, g.inc) and [h.random() for _ in range(5)] == [g.random() for _ in range(5)]


def test_derived_alignments_keep_the_seeds_a_generator_was_built_with():
    """pixel / sample generators are derived from a seed pair, so a PCG that has been drawn from still names
    the pair it was BUILT with (round 3's frames); only "seq" continues the stream from where the generator has got to."""
    world = scenes.synthetic_world(8)
    g, t = hm.PCG(45, 54), hm.PCG(7, 9)
    for _ in range(5):
        g.random()
        t.random()
    assert flatten.recover_seeds(g, constructed=True) == (45, 54) and flatten.recover_seeds(g) != (45, 54)
    pt = hm.PathTracer(world, pcg=g, num_of_rays=1, max_depth=2)
    for mode in (abi.PCG_PIXEL, abi.PCG_SAMPLE):
        par = flatten.renderer_params(pt, 16, 9, samples_per_side=2, tracer_pcg=t, pcg_mode=mode)
        assert (par.path_state, par.path_seq, par.jitter_state, par.jitter_seq) == (45, 54, 7, 9)
    par = flatten.renderer_params(hm.FlatRenderer(world), 16, 9, samples_per_side=2, tracer_pcg=t, pcg_mode=abi.PCG_SEQ)
    h = hm.PCG(par.jitter_state, par.jitter_seq)
    assert (h.state, h.inc) == (t.state, t.inc)  # "seq": the stream goes on where the tracer's generator is

    class Bare:  # the reference's PCG: (state, inc) only -> solved for, whatever the mode
        state, inc = hm.PCG(45, 54).state, hm.PCG(45, 54).inc

    assert flatten.recover_seeds(Bare, constructed=True) == (45, 54)


def test_seq_fixtures_exist_for_the_three_renderers():
    names = [n for n in util.FRAME_FIXTURES if n.startswith("g5_seq_")]
    kinds = set()
    for n in names:
        _, _, par, _ = util.load_frame(n)
        assert par.pcg_mode == abi.PCG_SEQ and par.samples_per_side > 0
        kinds.add(par.renderer)
    assert kinds == {abi.RENDERER_ONOFF, abi.RENDERER_FLAT, abi.RENDERER_POINTLIGHT}


# ---- on the MI355X ----------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def dev():
    from pytracer_amd import device

    assert device.device_count() >= 1, "no HIP device visible"
    return device


def _world(lights=0):
    world = scenes.synthetic_world(32, with_plane=True)
    for l in range(lights):
        world.add_light(hm.PointLight(hm.Vec(-3.0 + 4.0 * l, 6.0 - 9.0 * l, 8.0), hm.Color(1.0, 0.9, 0.8), 0.0))
    return world


@pytest.mark.gpu
@pytest.mark.parametrize("renderer,S,W,H", [(abi.RENDERER_FLAT, 2, 640, 360), (abi.RENDERER_ONOFF, 3, 320, 180),
                                            (abi.RENDERER_POINTLIGHT, 1, 320, 180), (abi.RENDERER_FLAT, 9, 96, 54),
                                            (abi.RENDERER_FLAT, 1, 1280, 720)])
def test_device_seq_equals_the_serial_oracle(dev, oracle, renderer, S, W, H):
    scene = flatten.flatten_world(_world(2 if renderer == abi.RENDERER_POINTLIGHT else 0))
    cam = flatten.flatten_camera(scenes.synthetic_camera(W, H))
    par = abi.make_params(W, H, renderer, samples_per_side=S, pcg_mode=abi.PCG_SEQ, jitter_state=42, jitter_seq=54,
                          path_state=999, path_seq=777)  # (the path seeds must play no part)
    with dev.DeviceScene(scene) as ds:
        out = ds.render(cam, par)
        st = ds.stats()
        # three ranks, 7-row blocks: every pixel enters the ONE stream at its own place, whoever renders it
        got = np.zeros_like(out)
        for rank in range(3):
            p = abi.copy_params(par, n_ranks=3, rank=rank, row_block=7)
            got[abi.rows_for_rank(H, 7, 3, rank)] = ds.render(cam, p)
    ora, n = oracle.render(scene, cam, par, sqr_mode=oracle.SQR_MUL)
    oracle.set_sqr_mode(oracle.SQR_POW)
    if renderer == abi.RENDERER_POINTLIGHT:  # (specular eval: acos)
        assert util.rel_err(out, ora).max() <= 1e-5
    else:
        assert util.bits_equal(out, ora)
    assert util.bits_equal(got, out) and int(st.n_rays) == n


@pytest.mark.gpu
def test_seq_distances_beyond_32_bits(dev, oracle):
    """The last rows of a frame whose pixels lie more than 2^32 draws into the stream (a 4K frame at 256 samples per
    pixel needs 4.2e9; here 3840 x 2160 at S = 24: 9.6e9), as one rank's 8-row block, against the serial oracle told to
    render the same rows -- which it reaches by drawing everything before them... no: it cannot skip either, so the
    comparison uses the oracle on a generator advanced on the host to the band's first pixel."""
    W, H, S = 3840, 2160, 24
    scene = flatten.flatten_world(_world())
    cam = flatten.flatten_camera(scenes.synthetic_camera(W, H))
    nblocks = H // 8
    par = abi.make_params(W, H, abi.RENDERER_FLAT, samples_per_side=S, pcg_mode=abi.PCG_SEQ, jitter_state=42, jitter_seq=54,
                          n_ranks=nblocks, rank=nblocks - 1, row_block=8)
    first_pixel = (H - 8) * W
    assert 2 * S * S * first_pixel > 2 ** 32
    with dev.DeviceScene(scene) as ds:
        out = ds.render(cam, par)
    assert out.shape == (8, W, 3)
    # the oracle's serial loop starts at the partition's first pixel with the generator it is given: hand it the seeds
    # of the generator that is where the reference's would be after (H - 8) * W pixels
    g = hm.PCG(42, 54)
    g.state = hm.pcg_advance(g.state, g.inc, 2 * S * S * first_pixel)
    js, jq = flatten.recover_seeds(g)
    ora, _ = oracle.render(scene, cam, abi.copy_params(par, jitter_state=js, jitter_seq=jq), sqr_mode=oracle.SQR_MUL)
    oracle.set_sqr_mode(oracle.SQR_POW)
    assert util.bits_equal(out, ora)


@pytest.mark.gpu
def test_drop_in_default_is_the_reference_stream_and_advances_the_tracers_pcg(dev, oracle):
    from pytracer_amd.tracer import GpuImageTracer

    W, H, S = 64, 36, 2
    world, camera = _world(), scenes.synthetic_camera(W, H)
    pcg = hm.PCG(42, 54)
    image = hm.HdrImage(W, H)
    tracer = GpuImageTracer(image, camera, samples_per_side=S, pcg=pcg)  # pcg_mode="auto"
    tracer.fire_all_rays(hm.FlatRenderer(world))
    first = image.array.copy()
    want = hm.PCG(42, 54)
    for _ in range(2 * W * H * S * S):
        want.random()
    assert pcg.state == want.state
    par = abi.make_params(W, H, abi.RENDERER_FLAT, samples_per_side=S, pcg_mode=abi.PCG_SEQ, jitter_state=42, jitter_seq=54)
    scene, cam = flatten.flatten_world(world), flatten.flatten_camera(camera)
    ora, _ = oracle.render(scene, cam, par, sqr_mode=oracle.SQR_MUL)
    assert util.bits_equal(first, ora)
    # a second frame through the same tracer continues the stream, as the reference's ImageTracer would
    tracer.fire_all_rays(hm.FlatRenderer(world))
    js, jq = flatten.recover_seeds(want)
    ora2, _ = oracle.render(scene, cam, abi.copy_params(par, jitter_state=js, jitter_seq=jq), sqr_mode=oracle.SQR_MUL)
    oracle.set_sqr_mode(oracle.SQR_POW)
    assert util.bits_equal(image.array, ora2) and not util.bits_equal(image.array, first)
    # ... and so does a tracer built WITHOUT a generator: its own PCG() is the reference's default PCG(42, 54),
    # frame 1 = the frame above, frame 2 continues 2 W H S^2 draws further on
    own = GpuImageTracer(image, camera, samples_per_side=S)
    own.fire_all_rays(hm.FlatRenderer(world))
    assert util.bits_equal(image.array, first) and own.pcg.state == want.state
    own.fire_all_rays(hm.FlatRenderer(world))
    assert util.bits_equal(image.array, ora2)
    own.close()
    # the path tracer takes one generator per sample under "auto" (one per pixel on request), and "seq" is refused for it
    pt = hm.PathTracer(world, pcg=hm.PCG(45, 54), num_of_rays=1, max_depth=2)
    tracer.fire_all_rays(pt)
    par_pt = abi.make_params(W, H, abi.RENDERER_PATHTRACER, samples_per_side=S, num_of_rays=1, max_depth=2, pcg_mode=abi.PCG_SAMPLE,
                             path_state=45, path_seq=54)
    with dev.DeviceScene(scene) as ds:
        assert util.bits_equal(image.array, ds.render(cam, par_pt))
        px_tracer = GpuImageTracer(image, camera, samples_per_side=S, pcg_mode="pixel")
        px_tracer.fire_all_rays(pt)
        assert util.bits_equal(image.array, ds.render(cam, abi.copy_params(par_pt, pcg_mode=abi.PCG_PIXEL)))
        px_tracer.close()
    with pytest.raises(Exception, match="serial"):
        GpuImageTracer(image, camera, samples_per_side=S, pcg_mode="seq").fire_all_rays(pt)
    tracer.close()
