"""the ``render`` command takes pytracer's flags (main.py:76-129) and produces what
``python -m pytracer render`` produces -- checked EXACTLY against frames the reference itself computed with the
CLI's defaults (tests/golden/make_golden.py: g5_cli_*; the CLI always jitters, S = 1)."""
import os
import sys

import numpy as np
import pytest
from click.testing import CliRunner

from tests import util

REF_SRC = "/root/reference/src"
REF_DEMO = "/root/reference/examples/demo.txt"


def test_cli_options_match_reference():
    from pytracer_amd.cli import render

    opts = {p.name: p for p in render.params}
    expected = {"width": 640, "height": 480, "algorithm": "pathtracing", "pfm_output": "output.pfm",
                "png_output": "output.png", "num_of_rays": 10, "max_depth": 3, "init_state": 45, "init_seq": 54,
                "samples_per_pixel": 1}
    for name, default in expected.items():
        assert opts[name].default == default, name
    assert opts["declare_float"].multiple and "-d" in opts["declare_float"].opts
    assert set(opts["algorithm"].type.choices) == {"onoff", "flat", "pathtracing", "pointlight"}
    assert "input_scene_name" in opts


def test_cli_rejects_bad_values_before_touching_a_gpu():
    from pytracer_amd.cli import cli

    for args, needle in ((["--samples-per-pixel", "3", "builtin:demo"], "perfect square"),
                         (["-d", "clock", "builtin:demo"], "NAME:VALUE"),
                         (["-d", "a:b:c", "builtin:demo"], "NAME:VALUE"),
                         (["-d", "clock:fast", "builtin:demo"], "not a floating-point"),
                         (["builtin:nosuch"], "no built-in scene"),
                         (["--pcg-mode", "seq", "builtin:demo"], "serial by"),  # (pathtracing is the default algorithm)
                         (["/nonexistent/scene.txt"], "")):
        r = CliRunner().invoke(cli, ["render"] + args)
        assert r.exit_code == 2, (args, r.output)
        assert needle in r.output


def test_float_overrides():
    from pytracer_amd.cli import UsageError, parse_float_overrides

    assert parse_float_overrides(["clock:150", "x:-2.5e-1"]) == {"clock": 150.0, "x": -0.25}
    assert parse_float_overrides([]) == {}
    for bad in ("clock", ":1", "a:1:2", "a:"):
        with pytest.raises(UsageError):
            parse_float_overrides([bad])


def test_plan_render_builds_the_four_renderers():
    from pytracer_amd import flatten
    from pytracer_amd.cli import plan_render

    for algo, kind in (("onoff", "OnOffRenderer"), ("flat", "FlatRenderer"), ("pathtracing", "PathTracer"),
                       ("pointlight", "PointLightRenderer")):
        job = plan_render(64, 48, algo, 7, 2, 45, 54, 4, ("clock:10",), "builtin:demo")
        assert type(job.renderer).__name__ == kind and job.samples_per_side == 2
        par = flatten.renderer_params(job.renderer, 64, 48, samples_per_side=job.samples_per_side)
        if algo == "pathtracing":
            assert (par.num_of_rays, par.max_depth, par.rr_limit, par.path_state, par.path_seq) == (7, 2, 3, 45, 54)


@pytest.fixture()
def reference_importable():
    """Build container only: pytracer's parser on sys.path for the scene-file branch (removed afterwards)."""
    if not (os.path.isdir(os.path.join(REF_SRC, "pytracer")) and os.path.exists(REF_DEMO)):
        pytest.skip("the reference is not present here")
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF_SRC)
    try:
        yield
    finally:
        sys.path.remove(REF_SRC)
        for name in [m for m in sys.modules if m == "pytracer" or m.startswith("pytracer.")]:
            del sys.modules[name]


def test_scene_file_branch_flattens_to_the_golden_scene(reference_importable):
    """`render SCENE.txt` up to (not including) the GPU: pytracer's parser reads examples/demo.txt, the renderer is
    the reference's own class, and flattening gives bit for bit the scene + camera + parameters of the golden
    fixture the reference rendered (so the frame the device would produce is the one test_cli_* pins)."""
    from pytracer_amd import abi, flatten
    from pytracer_amd.cli import plan_render

    job = plan_render(32, 24, "pathtracing", 10, 3, 45, 54, 1, ("clock:150",), REF_DEMO)
    assert type(job.renderer).__module__.startswith("pytracer.")
    scene, cam, par, _ = util.load_frame("g5_cli_demo_path_s1_32x24_n10d3")
    assert flatten.flatten_world(job.world).same_bits(scene)
    got_cam = flatten.flatten_camera(job.camera)
    assert bytes(got_cam) == bytes(cam)
    from pytracer_amd.hostmodel import PCG
    got_par = flatten.renderer_params(job.renderer, 32, 24, samples_per_side=job.samples_per_side, tracer_pcg=PCG(),
                                      pcg_mode=abi.PCG_PIXEL)
    assert bytes(got_par) == bytes(par)
    # the built-in demo scene is the same data
    builtin = plan_render(32, 24, "pathtracing", 10, 3, 45, 54, 1, (), "builtin:demo")
    assert flatten.flatten_world(builtin.world).same_bits(scene)
    assert bytes(flatten.flatten_camera(builtin.camera)) == bytes(cam)
    # a -d override reaches the parser
    other = plan_render(32, 24, "flat", 10, 3, 45, 54, 1, ("clock:10",), REF_DEMO)
    assert not flatten.flatten_world(other.world).same_bits(scene)


def _read_pfm(path, w, h):
    raw = open(path, "rb").read()
    head = f"PF\n{w} {h}\n-1.0\n".encode()
    assert raw.startswith(head)
    return np.frombuffer(raw[len(head):], dtype="<f4").reshape(h, w, 3)[::-1]


@pytest.mark.gpu
@pytest.mark.parametrize("algo,fixture,w,h,extra", [
    # the reference CLI's own frames: jitter from ImageTracer's default PCG(42, 54), ONE sequential stream (main.py:168-170;
    # tests/golden/make_golden.py g5_seq: the verbatim reference, no re-seeding) -- the default of `render` since round 4
    ("flat", "g5_seq_cli_demo_flat_s1_64x48", 64, 48, []),
    ("onoff", "g5_seq_cli_demo_onoff_s1_48x36", 48, 36, []),
    ("pointlight", "g5_seq_cli_demo_pointlight_s1_48x36", 48, 36, []),
    ("flat", "g5_seq_cli_demo_flat_s1_64x48", 64, 48, ["--pcg-mode", "seq"]),
    # one generator per pixel (the reference driven through a re-seeding proxy: g5_cli)
    ("flat", "g5_cli_demo_flat_s1_64x48", 64, 48, ["--pcg-mode", "pixel"]),
    ("pathtracing", "g5_cli_demo_path_s1_32x24_n10d3", 32, 24, []),
    ("pathtracing", "g5_cli_demo_path_s1_32x24_n10d3", 32, 24, ["--pcg-mode", "pixel"])])
def test_cli_frame_equals_reference_cli_frame(tmp_path, algo, fixture, w, h, extra):
    """The PFM the CLI writes == the reference's frame for the same command line, rounded to the PFM's fp32
    (hdrimages.py:35-43): exactly for OnOff and Flat (no libm on the path: checkered planes and a uniform mirror), within
    1e-5 for PointLight and the path tracer (acos / sin / cos of ocml vs glibc)."""
    from pytracer_amd.cli import cli

    pfm, png = str(tmp_path / "o.pfm"), str(tmp_path / "o.png")
    r = CliRunner().invoke(cli, ["render", "--width", str(w), "--height", str(h), "--algorithm", algo,
                                 "--pfm-output", pfm, "--png-output", png, "-d", "clock:150"] + extra + ["builtin:demo"])
    assert r.exit_code == 0, r.output
    assert os.path.getsize(png) > 100
    img = _read_pfm(pfm, w, h)
    gold = util.load(fixture)["pixels"]
    if algo in ("flat", "onoff"):
        assert np.array_equal(img, gold.astype(np.float32))
    else:
        err = util.rel_err(img, gold.astype(np.float32))
        assert err.max() <= 1e-5, f"max rel err {err.max():.3e}"


@pytest.mark.gpu
@pytest.mark.parametrize("algo,w,h,extra", [("flat", 64, 48, []), ("pathtracing", 32, 24, []),
                                            ("pathtracing", 40, 30, ["--samples-per-pixel", "4", "--num-of-rays", "2", "--pcg-mode", "sample"]),
                                            ("pointlight", 64, 48, []), ("onoff", 33, 17, [])])
def test_cli_post_processing_on_the_resident_frame_writes_the_same_bytes(tmp_path, algo, w, h, extra):
    """main.py:203-213 on the frame left in HBM (the PFM floats and the tone-mapped bytes are the only device-to-host
    traffic) == the same steps after copying the fp64 frame to the host first (--host-postprocess): PFM and PNG files
    byte for byte."""
    from pytracer_amd.cli import cli

    files = {}
    for tag, flag in (("dev", []), ("host", ["--host-postprocess"])):
        pfm, png = str(tmp_path / f"{tag}.pfm"), str(tmp_path / f"{tag}.png")
        r = CliRunner().invoke(cli, ["render", "--width", str(w), "--height", str(h), "--algorithm", algo, "--pfm-output", pfm,
                                     "--png-output", png, "-d", "clock:150"] + extra + flag + ["builtin:demo"])
        assert r.exit_code == 0, r.output
        files[tag] = (open(pfm, "rb").read(), open(png, "rb").read())
    assert files["dev"][0] == files["host"][0], "PFM differs"
    assert files["dev"][1] == files["host"][1], "PNG differs"
    assert len(files["dev"][0]) == len(f"PF\n{w} {h}\n-1.0\n") + w * h * 12


@pytest.mark.gpu
def test_cli_pcg_mode_sample_is_the_sample_aligned_frame(tmp_path):
    """--pcg-mode sample: sample k of pixel i owns PCG(init_state, init_seq + i * S^2 + k) (SURVEY.md 8c Mode SAMPLE)."""
    from pytracer_amd import abi, flatten, scenes
    from pytracer_amd.cli import cli
    from pytracer_amd.device import DeviceScene

    w, h = 48, 36
    pfm, png = str(tmp_path / "o.pfm"), str(tmp_path / "o.png")
    r = CliRunner().invoke(cli, ["render", "--width", str(w), "--height", str(h), "--algorithm", "pathtracing", "--samples-per-pixel", "4",
                                 "--num-of-rays", "1", "--max-depth", "3", "--pcg-mode", "sample", "--pfm-output", pfm,
                                 "--png-output", png, "-d", "clock:150", "builtin:demo"])
    assert r.exit_code == 0, r.output
    world, camera = scenes.demo_world(clock=150.0)
    par = abi.make_params(w, h, abi.RENDERER_PATHTRACER, samples_per_side=2, num_of_rays=1, max_depth=3, pcg_mode=abi.PCG_SAMPLE,
                          path_state=45, path_seq=54, out_format=abi.OUT_F32)
    with DeviceScene(flatten.flatten_world(world)) as ds:
        direct = ds.render(flatten.flatten_camera(camera), par)
    assert np.array_equal(_read_pfm(pfm, w, h), direct)


@pytest.mark.gpu
def test_cli_scene_file_on_the_gpu(tmp_path, reference_importable):
    """Where both the reference and a GPU exist: the scene-file branch end to end."""
    from pytracer_amd.cli import cli

    pfm, png = str(tmp_path / "o.pfm"), str(tmp_path / "o.png")
    r = CliRunner().invoke(cli, ["render", "--width", "64", "--height", "48", "--algorithm", "flat",
                                 "--pfm-output", pfm, "--png-output", png, REF_DEMO])
    assert r.exit_code == 0, r.output
    assert np.array_equal(_read_pfm(pfm, 64, 48), util.load("g5_seq_cli_demo_flat_s1_64x48")["pixels"].astype(np.float32))


@pytest.mark.gpu
def test_cli_renders_without_torch(tmp_path):
    """`python -m pytracer_amd render builtin:c2` in a process where torch CANNOT be imported (masked in
    sys.modules) -- the resident frame's HBM and stream come from the C-ABI (pt_device_alloc, ABI 1.5) -- writes the same PFM
    and PNG bytes as the in-process run of this test process, where torch may be loaded."""
    import subprocess
    import sys

    from pytracer_amd.cli import cli

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = ["render", "--width", "160", "--height", "90", "--algorithm", "pathtracing", "--samples-per-pixel", "4", "--num-of-rays", "2"]
    pfm0, png0 = str(tmp_path / "a.pfm"), str(tmp_path / "a.png")
    r = CliRunner().invoke(cli, args + ["--pfm-output", pfm0, "--png-output", png0, "builtin:c2"])
    assert r.exit_code == 0, r.output
    pfm1, png1 = str(tmp_path / "b.pfm"), str(tmp_path / "b.png")
    code = ("import sys; sys.modules['torch'] = None\n"
            "from pytracer_amd.cli import cli\n"
            "try:\n"
            "    cli(sys.argv[1:], standalone_mode=False)\n"
            "finally:\n"
            "    mods = sorted(m for m in sys.modules if m == 'torch' or m.startswith('torch.'))\n"
            "    assert mods == ['torch'] and sys.modules['torch'] is None, mods\n"
            "    maps = open('/proc/self/maps').read()\n"
            "    assert 'libptrace.so' in maps and 'site-packages/torch' not in maps and 'libtorch' not in maps\n"
            "    print('no torch in this process')\n")
    r = subprocess.run([sys.executable, "-c", code] + args + ["--pfm-output", pfm1, "--png-output", png1, "builtin:c2"],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0 and "no torch in this process" in r.stdout, r.stdout + r.stderr
    assert open(pfm0, "rb").read() == open(pfm1, "rb").read() and open(png0, "rb").read() == open(png1, "rb").read()
