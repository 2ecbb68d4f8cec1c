"""The committed fixtures are what tests/golden/make_golden.py makes of the reference: where the reference is present
(the build container; it never travels to the GPU box) a fixture is regenerated and compared array by array."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference/src/pytracer"


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference is not present here")
@pytest.mark.parametrize("gen,files", [("g1", ["g1_pcg"]), ("g4", ["g4_camera"]),
                                       ("g5cli", ["g5_cli_demo_flat_s1_64x48", "g5_cli_demo_path_s1_32x24_n10d3"])])
def test_regenerated_fixture_equals_the_committed_one(tmp_path, gen, files):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_golden.py"), "--out", str(tmp_path), gen],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    for name in files:
        new = np.load(tmp_path / (name + ".npz"))
        old = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        assert sorted(new.files) == sorted(old.files), name
        for k in old.files:
            assert new[k].dtype == old[k].dtype and new[k].shape == old[k].shape, (name, k)
            assert new[k].tobytes() == old[k].tobytes(), (name, k)


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference is not present here")
def test_two_parses_in_one_process_are_refused(tmp_path):
    """the parser's default World() is shared by all parses of a process."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_golden.py"), "--out", str(tmp_path), "g5", "g5cli"],
                       capture_output=True, text=True, env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"), timeout=600)
    assert r.returncode != 0 and "separate processes" in (r.stderr + r.stdout)
