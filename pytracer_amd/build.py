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
# every file the one translation unit includes: all of csrc/ (tests/test_host.py checks this list against the #include lines)
DEPS = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))) + [os.path.join("..", "..", "include", "ptrace.h"),
                                                                            os.path.join("..", "..", "include", "ptrace_debug.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-fast-math", "-std=c++17", "-fPIC",
         "-shared", "-Wall", "-Wno-unused-function", "-Wno-pass-failed",
         # a fixed compilation-unit id: by default clang derives it from the command line (paths included), which would make
         # code_hash() depend on WHERE the library was built
         "-cuid=libptrace"]


def code_hash(lib: str = LIB) -> str:
    """sha256 of the DEVICE code in ``lib``: the bytes of its ``.hip_fatbin`` ELF section (the gfx950 code object bundle
    hipcc embedded).  Two builds with this hash equal run the same kernels; bench.py compares it with the hash recorded
    in every ``profiles/pmc_*.json`` it prices a roofline from and refuses to price on a mismatch.
    Pure Python (ELF64 little-endian section table): needs no ROCm tool, works wherever the library file is."""
    import hashlib
    import struct

    with open(lib, "rb") as f:
        data = f.read()
    if data[:4] != b"\x7fELF" or data[4] != 2 or data[5] != 1:
        raise ValueError(f"{lib}: not a little-endian ELF64 file")
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)

    def section(i):
        name, _type, _flags, _addr, off, size = struct.unpack_from("<IIQQQQ", data, shoff + i * shentsize)
        return name, off, size

    _, stroff, strsize = section(shstrndx)
    names = data[stroff:stroff + strsize]
    for i in range(shnum):
        name, off, size = section(i)
        if names[name:names.index(b"\0", name)] == b".hip_fatbin":
            return hashlib.sha256(data[off:off + size]).hexdigest()
    raise ValueError(f"{lib}: no .hip_fatbin section (not a HIP library?)")


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
