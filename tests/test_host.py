"""CPU-only tests of the host side: the stand-in scene objects flatten to exactly the arrays the
reference's objects flatten to, seeds are recovered from PCG objects, the C-ABI library loads and
exports every symbol include/ptrace.h declares (no compute without a GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from pytracer_amd import abi, flatten, hostmodel as hm, scenes
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_demo_world_matches_reference_parse():
    """scenes.demo_world() == what the reference's parser builds from examples/demo.txt (fixture)."""
    world, camera = scenes.demo_world()
    gold_scene, gold_cam, _, _ = util.load_frame("g5_demo_flat_160x120")
    assert flatten.flatten_world(world).same_bits(gold_scene)
    cam = flatten.flatten_camera(camera)
    assert bytes(cam) == bytes(gold_cam)


def test_synthetic_world_matches_reference_recipe():
    gold_scene, gold_cam, _, _ = util.load_frame("g5_c2_flat_160x90")
    assert flatten.flatten_world(scenes.synthetic_world(32, with_plane=True)).same_bits(gold_scene)
    assert bytes(flatten.flatten_camera(scenes.synthetic_camera(160, 90))) == bytes(gold_cam)
    gold_scene, _, _, _ = util.load_frame("g5_c3_path_80x45_seq")
    assert flatten.flatten_world(scenes.synthetic_world(32)).same_bits(gold_scene)


def test_transform_builders_match_golden_matrices():
    # g4 camera 2: rotation_z(30) * translation(-4, 0, 1); camera 4: translation(-2 * VEC_Y) * rotation_z(90)
    g = util.load("g4_camera")
    t = hm.rotation_z(30.0) * hm.translation(hm.Vec(-4.0, 0.0, 1.0))
    assert np.array_equal(np.array(t.m[:3]).reshape(-1), g["c2_cam_m"])
    t = hm.translation(hm.Vec(-0.0, -2.0, -0.0)) * hm.rotation_z(90)
    assert np.array_equal(np.array(t.m[:3]).reshape(-1), g["c4_cam_m"])
    # inverse really is the inverse
    t = hm.translation(hm.Vec(1.0, -2.0, 3.0)) * hm.rotation_x(33.0) * hm.rotation_y(-71.0) * hm.scaling(hm.Vec(2.0, 0.5, 4.0))
    prod = np.array(t.m) @ np.array(t.invm)
    assert np.allclose(prod, np.eye(4), atol=1e-12)


def test_recover_seeds():
    for s, q in ((42, 54), (45, 54), (0, 0), (123456789012345, 2 ** 62 + 17)):
        p = hm.PCG(s, q)

        class Bare:  # like the reference's PCG: only state and inc
            state, inc = p.state, p.inc

        assert flatten.recover_seeds(Bare) == (s, q)
        assert flatten.recover_seeds(p) == (s, q)


def test_renderer_params_and_errors():
    world, _ = scenes.demo_world()
    p = flatten.renderer_params(hm.PathTracer(world, pcg=hm.PCG(45, 54), num_of_rays=7, max_depth=4,
                                              russian_roulette_limit=2), 64, 48, samples_per_side=3)
    assert (p.renderer, p.num_of_rays, p.max_depth, p.rr_limit) == (abi.RENDERER_PATHTRACER, 7, 4, 2)
    assert (p.path_state, p.path_seq, p.samples_per_side) == (45, 54, 3)
    p = flatten.renderer_params(hm.OnOffRenderer(world, color=hm.Color(0.5, 0.25, 1.0)), 8, 8)
    assert list(p.onoff_color) == [0.5, 0.25, 1.0]
    with pytest.raises(flatten.UnsupportedSceneError):
        flatten.renderer_params(lambda ray: None, 8, 8)
    # non-affine matrices are refused, not silently mis-rendered
    s = hm.Sphere()
    s.transformation.m = [[1.0, 0, 0, 0], [0, 1.0, 0, 0], [0, 0, 1.0, 0], [0, 0, 0.1, 1.0]]
    w = hm.World()
    w.add_shape(s)
    with pytest.raises(flatten.UnsupportedSceneError):
        flatten.flatten_world(w)

    class Torus:
        transformation = hm.Transformation()
        material = hm.Material()

    w = hm.World()
    w.add_shape(Torus())
    with pytest.raises(flatten.UnsupportedSceneError):
        flatten.flatten_world(w)


def test_rows_for_rank_partition():
    for h, rb, nr in ((720, 8, 8), (721, 8, 3), (5, 8, 2), (2160, 16, 8)):
        seen = []
        for r in range(nr):
            seen += abi.rows_for_rank(h, rb, nr, r)
        assert sorted(seen) == list(range(h))


def test_hdrimage_standin():
    img = hm.HdrImage(7, 4)
    assert img.pixel_offset(3, 2) == 17  # test_all.py:163-168
    img.set_pixel(3, 2, hm.Color(1.0, 2.0, 3.0))
    assert img.get_pixel(3, 2) == hm.Color(1.0, 2.0, 3.0)
    assert img.pixels[17] == hm.Color(1.0, 2.0, 3.0)
    assert not img.valid_coordinates(7, 0) and img.valid_coordinates(6, 3)


def test_c_abi_library_exports_header_symbols():
    """libptrace.so loads without a GPU and exports every function include/ptrace.h declares."""
    from pytracer_amd import _lib, build

    build.build()
    header = open(os.path.join(ROOT, "include", "ptrace.h")).read()
    declared = set(re.findall(r"\b(pt_[a-z_]+)\s*\(", header))
    declared -= {"pt_scene"}
    lib = _lib.lib()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in ptrace.h but not exported"
    assert set(_lib.EXPORTS) == declared
    # ... and the diagnostics live in their own header: everything the library exports is declared in one of the two
    dbg_header = open(os.path.join(ROOT, "include", "ptrace_debug.h")).read().split("#ifdef PT_DEBUG_TIME")[0]
    dbg_declared = set(re.findall(r"\b(pt_debug_[a-z_]+)\s*\(", dbg_header))
    assert set(_lib.DEBUG_EXPORTS) == dbg_declared
    for name in dbg_declared:
        assert hasattr(lib, name), f"{name} declared in ptrace_debug.h but not exported"
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.lib_path()], capture_output=True, text=True).stdout
    exported = set(re.findall(r"\bT (pt_[a-z_0-9]+)$", nm, flags=re.M))
    assert exported == declared | dbg_declared, exported ^ (declared | dbg_declared)
    assert lib.pt_version() >> 16 == 1 and (lib.pt_version() & 0xFFFF) >= 3  # (1.3: include/ptrace.h)
    # pure host-side entry points work without a device
    p = abi.make_params(1280, 721, abi.RENDERER_FLAT, n_ranks=3, rank=1, row_block=8, out_format=abi.OUT_F32)
    rows = len(abi.rows_for_rank(721, 8, 3, 1))
    assert lib.pt_rows_for_rank(C.byref(p)) == rows
    assert lib.pt_output_bytes(C.byref(p)) == rows * 1280 * 3 * 4


def test_struct_sizes_match_header():
    # catches ctypes/ABI drift: sizes computed from the C declarations
    assert C.sizeof(abi.Camera) == 4 + 4 + 12 * 8 + 16
    assert C.sizeof(abi.Params) == 4 * 4 + 9 * 8 + 4 * 4 + 4 * 8 + 4 * 4
    assert C.sizeof(abi.Stats) == 2 * 8 + 2 * 8 + 4 * 4 + 8 + 2 * 4
    assert C.sizeof(abi.SceneDesc) == 25 * 8


def test_fill_image_into_reference_style_hdrimage():
    """pytracer's HdrImage keeps a list of Color objects (hdrimages.py:70): filled with its own class."""
    from pytracer_amd.tracer import _fill_image

    class RefColor:
        def __init__(self, r=0.0, g=0.0, b=0.0):
            self.r, self.g, self.b = r, g, b

    class RefImage:
        def __init__(self, w, h):
            self.width, self.height = w, h
            self.pixels = [RefColor() for _ in range(w * h)]

    img = RefImage(3, 2)
    arr = np.arange(18, dtype=np.float64).reshape(2, 3, 3)
    held = img.pixels
    _fill_image(img, arr)  # (in place; the opt-in lazy sequence: tests/test_lazy_pixels.py)
    assert img.pixels is held
    assert all(isinstance(c, RefColor) for c in img.pixels)
    assert (img.pixels[4].r, img.pixels[4].g, img.pixels[4].b) == (12.0, 13.0, 14.0)  # (x=1, y=1) -> 1*3+1


def test_sparse_shard_sizes_agree_between_library_and_python():
    """pt_image_sparse_fixed_bytes (no device needed) against pytracer_amd.dist.sparse_fixed_bytes: the two sides of the
    gather must agree on what the fixed-size message weighs."""
    from pytracer_amd import _lib, abi, dist as ptdist

    L = _lib.lib()
    for npx in (1, 2, 127, 128, 129, 255, 256, 257, 3840 * 270, 3840 * 1080, 10 ** 7 + 3):
        assert L.pt_image_sparse_fixed_bytes(npx, abi.OUT_F32) == ptdist.sparse_fixed_bytes(npx, 4)
        assert L.pt_image_sparse_fixed_bytes(npx, abi.OUT_F64) == ptdist.sparse_fixed_bytes(npx, 8)
    assert L.pt_image_sparse_fixed_bytes(0, abi.OUT_F32) == -1 and L.pt_image_sparse_fixed_bytes(10, 7) == -1


def test_the_build_recipe_watches_every_file_the_library_includes():
    """`needs_build()` must see an edit of ANY header of the translation unit (pt_plan.h was missing)."""
    import re

    from pytracer_amd import build

    seen, todo = set(), ["ptrace.hip"]
    while todo:
        f = todo.pop()
        if f in seen:
            continue
        seen.add(f)
        for inc in re.findall(r'#include\s+"([^"]+)"', open(os.path.join(build.CSRC, f)).read()):
            todo.append(os.path.normpath(os.path.join(os.path.dirname(f), inc)))
    deps = {os.path.normpath(d) for d in build.DEPS}
    assert seen <= deps, seen - deps
    assert "pt_plan.h" in deps and all(os.path.exists(os.path.join(build.CSRC, d)) for d in build.DEPS)
