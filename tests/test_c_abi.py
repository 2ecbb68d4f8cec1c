"""The drop-in boundary is a C ABI (include/ptrace.h: `extern "C"`, plain pointers and sizes): a C99 program built with
gcc against the header and the library -- no Python, no torch in the process -- must see the same struct layouts as the
ctypes mirror (pytracer_amd/abi.py), and, on the GPU box, render the frame the Python binding renders."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from pytracer_amd import _lib, abi, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_abi", "probe.c")


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    build.build()
    exe = str(tmp_path_factory.mktemp("c_abi") / "probe")
    libdir = os.path.dirname(_lib.lib_path())
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
                        "-L", libdir, "-lptrace", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr  # the header must be valid, warning-free C99
    return exe


def test_header_is_c99_and_layouts_match_the_ctypes_mirror(probe):
    r = subprocess.run([probe, "sizes"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert int(lines[0].split()[1]) == _lib.lib().pt_version()
    w = lines[1].split()
    sizes = {w[i]: int(w[i + 1]) for i in range(1, len(w), 2)}
    assert sizes == {"pt_scene_desc": C.sizeof(abi.SceneDesc), "pt_camera": C.sizeof(abi.Camera),
                     "pt_params": C.sizeof(abi.Params), "pt_stats": C.sizeof(abi.Stats)}
    w = lines[2].split()
    p = abi.make_params(7, 21, abi.RENDERER_ONOFF, row_block=8, n_ranks=2, rank=1, out_format=abi.OUT_F32)
    assert int(w[1]) == len(abi.rows_for_rank(21, 8, 2, 1)) == 8
    assert int(w[3]) == 8 * 7 * 3 * 4 == int(_lib.lib().pt_output_bytes(C.byref(p)))
    assert int(w[5]) == _lib.lib().pt_image_sparse_fixed_bytes(1000, abi.OUT_F32)


@pytest.mark.gpu
def test_a_c_program_renders_the_frame_the_python_binding_renders(probe, oracle):
    from pytracer_amd import flatten, hostmodel as hm
    from pytracer_amd.device import DeviceScene

    W, H = 40, 24
    r = subprocess.run([probe, "render", str(W), str(H)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert lines[0].split()[:4] == ["rays", str(W * H), "pixels", str(W * H)]
    got = np.array([float.fromhex(x) for x in lines[1:1 + W * H * 3]]).reshape(H, W, 3)
    assert lines[-1] == f"short buffer -> {-5}"  # PT_ERR_SIZE
    # ABI 1.5: the same frame through pt_device_alloc + pt_stream_create + pt_render_device + pt_device_download
    assert lines[-4:-1] == ["device frame identical 1", "free(NULL) 0 0 alloc(0) 1", "bad device -> -1"]
    # the same scene through the Python objects and the ctypes binding, and through the oracle
    w = hm.World()
    w.add_shape(hm.Sphere(hm.translation(hm.Vec(2.0, 0.25, 0.5)) * hm.scaling(hm.Vec(0.5, 0.5, 0.5)),
                          hm.Material(hm.DiffuseBRDF(hm.UniformPigment(hm.Color(0.9, 0.3, 0.2))), hm.UniformPigment(hm.Color(0.125, 0.0, 0.0)))))
    w.add_shape(hm.Plane(hm.Transformation(), hm.Material(hm.DiffuseBRDF(hm.CheckeredPigment(hm.Color(0.5, 0.1, 0.1), hm.Color(0.2, 0.0, 0.5), 4)))))
    scene = flatten.flatten_world(w)
    cam = flatten.flatten_camera(hm.PerspectiveCamera(1.0, W / H, hm.translation(hm.Vec(-1.0, 0.0, 1.0))))
    par = abi.make_params(W, H, abi.RENDERER_FLAT, background=(0.0, 0.0, 0.25), num_of_rays=1)
    with DeviceScene(scene) as ds:
        py = ds.render(cam, par)
    assert got.tobytes() == py.tobytes()
    ora, _ = oracle.render(scene, cam, par, sqr_mode=oracle.SQR_MUL)
    oracle.set_sqr_mode(oracle.SQR_POW)
    assert got.tobytes() == ora.tobytes()
    assert len({tuple(px) for px in got.reshape(-1, 3).tolist()}) >= 4  # background, sphere, both checker colours
