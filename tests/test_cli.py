"""the ``render`` command mirrors pytracer's flags (main.py:76-129)."""
import os

import numpy as np
import pytest
from click.testing import CliRunner

from tests import util


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


def test_cli_rejects_non_square_samples():
    from pytracer_amd.cli import cli

    r = CliRunner().invoke(cli, ["render", "--samples-per-pixel", "3", "builtin:demo"])
    assert "must be a perfect square" in r.output


@pytest.mark.gpu
def test_cli_renders_demo(tmp_path):
    from pytracer_amd.cli import cli

    pfm, png = str(tmp_path / "o.pfm"), str(tmp_path / "o.png")
    r = CliRunner().invoke(cli, ["render", "--width", "160", "--height", "120", "--algorithm", "flat",
                                 "--samples-per-pixel", "1", "--pfm-output", pfm, "--png-output", png,
                                 "-d", "clock:150", "builtin:demo"])
    assert r.exit_code == 0, r.output
    assert "Using flat renderer" in r.output and os.path.getsize(png) > 100
    raw = open(pfm, "rb").read()
    assert raw.startswith(b"PF\n160 120\n-1.0\n")
    img = np.frombuffer(raw[len(b"PF\n160 120\n-1.0\n"):], dtype="<f4").reshape(120, 160, 3)[::-1]
    # the CLI always jitters (samples_per_side = 1): compare to the un-jittered golden loosely
    gold = util.load("g5_demo_flat_160x120")["pixels"]
    assert np.mean(np.abs(img - gold) < 1e-6) > 0.9
