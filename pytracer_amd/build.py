"""Build recipe for libptrace.so (HIP, gfx950 only).

``hipcc --offload-arch=gfx950 -O3 -ffp-contract=off``: the parity kernels must not fuse a*b+c (the
reference is Python: every operation rounds), and nothing enables fast-math.  The library is built
in-tree (``pytracer_amd/libptrace.so``) so it travels to the GPU box with the repository snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libptrace.so")
SOURCES = ["ptrace.hip"]
DEPS = ["ptrace.hip", "pt_kernels.h", "pt_math.h", "pt_query.h", "pt_shade.h", "pt_camera.h", "pt_simple.h", "pt_tile.h", "pt_path.h",
        "pt_tree.h", "pt_probes.h", "pt_layout.h", "pt_post.h", os.path.join("..", "..", "include", "ptrace.h"),
        os.path.join("..", "..", "include", "ptrace_debug.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-fast-math", "-std=c++17", "-fPIC",
         "-shared", "-Wall", "-Wno-unused-function", "-Wno-pass-failed"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm): cannot build libptrace.so")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc()] + FLAGS + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed ({r.returncode}):\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
