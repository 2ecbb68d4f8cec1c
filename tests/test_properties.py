"""Size-independent properties of the path (what a frame must satisfy whatever its size), on the oracle here
and on the device at BASELINE's full sizes in tests/test_gpu_fullsize.py.

LINEARITY IN THE EMITTED RADIANCE.  Every renderer's pixel is a sum of products in which the emitted radiances (and the
background, and the light colours) enter linearly (render.py:65-74, 103-139, 157-193), and multiplying a double by two is
exact: a scene whose emitted radiances, background and lights are all doubled renders EXACTLY twice the frame -- bit for bit,
sums and all, since scaling by two commutes with every rounding (no overflow or underflow anywhere near these magnitudes).
The BRDF pigments (which steer Russian roulette: render.py:116-123) stay, so the same random numbers are drawn and the same
rays traced.  A kernel that drops a term, adds one twice or mixes the order of a sum in a value-dependent way fails this at
any frame size without needing a reference frame."""
import dataclasses

import numpy as np
import pytest

from pytracer_amd import abi, flatten, hostmodel as hm, scenes


def doubled(flat, pigments_too=False):
    """The same scene with every emitted radiance and light colour times two (exact).  `pigments_too`: the BRDF pigments as
    well -- FlatRenderer returns pigment + emitted radiance (render.py:72-74), so ITS sources are both."""
    k = 2.0 if pigments_too else 1.0
    return dataclasses.replace(flat, emi_c1=flat.emi_c1 * 2.0, emi_c2=flat.emi_c2 * 2.0, light_color=flat.light_color * 2.0,
                               pig_c1=flat.pig_c1 * k, pig_c2=flat.pig_c2 * k, tex_data=flat.tex_data.copy())


CASES = [("flat", dict(renderer=abi.RENDERER_FLAT, samples_per_side=2, pcg_mode=abi.PCG_PIXEL)),
         ("path N=1 D=3", dict(renderer=abi.RENDERER_PATHTRACER, samples_per_side=2, num_of_rays=1, max_depth=3, rr_limit=2, path_state=45, path_seq=54)),
         ("path N=3 D=2", dict(renderer=abi.RENDERER_PATHTRACER, samples_per_side=1, num_of_rays=3, max_depth=2, rr_limit=1, path_state=45, path_seq=54,
                               pcg_mode=abi.PCG_SAMPLE)),
         ("pointlight", dict(renderer=abi.RENDERER_POINTLIGHT, samples_per_side=0))]


def _world():
    w = scenes.synthetic_world(32, with_plane=True)
    w.add_light(hm.PointLight(hm.Vec(-3.0, 6.0, 8.0), hm.Color(1.0, 0.9, 0.8), 0.0))
    # something that emits besides the sky: a small lamp (emitted radiance enters through more than one path length)
    w.add_shape(hm.Sphere(hm.translation(hm.Vec(3.0, 0.5, 1.5)) * hm.scaling(hm.Vec(0.3, 0.3, 0.3)),
                          hm.Material(hm.DiffuseBRDF(hm.UniformPigment(hm.Color(0.6, 0.6, 0.6))), hm.UniformPigment(hm.Color(3.0, 2.5, 1.25)))))
    return w


@pytest.mark.parametrize("name,kw", CASES)
def test_oracle_frames_are_linear_in_the_emitted_radiance(oracle, name, kw):
    W, H = 96, 54
    flat = flatten.flatten_world(_world())
    cam = flatten.flatten_camera(scenes.synthetic_camera(W, H))
    bg = (0.125, 0.25, 0.0625)
    one, n1 = oracle.render(flat, cam, abi.make_params(W, H, background=bg, ambient=(0.0625, 0.03125, 0.125), **kw))
    two, n2 = oracle.render(doubled(flat, pigments_too=kw["renderer"] == abi.RENDERER_FLAT), cam, abi.make_params(W, H, background=tuple(2 * c for c in bg), ambient=(0.125, 0.0625, 0.25), **kw))
    assert n1 == n2, "the same rays are traced"
    assert np.isfinite(one).all() and (one > 0).any()
    assert (2.0 * one).tobytes() == two.tobytes(), f"{name}: doubling every source does not double the frame bit for bit"
