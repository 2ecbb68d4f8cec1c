#!/usr/bin/env python3
"""Registers, scratch and LDS of every kernel in libptrace.so (from the gfx950 code object's metadata).

    python tools/kres.py [path/to/libptrace.so] [substring ...]

Importable: ``kernel_resources(lib) -> {demangled name: {"vgpr", "agpr", "sgpr", "scratch", "lds"}}`` (tools/pmc_summary.py
records a profiled kernel's row beside its counters) and ``code_object(lib, out)`` (tools/isa_mix.py disassembles it).
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
DEFAULT_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pytracer_amd", "libptrace.so")


def code_object(lib: str, out: str) -> str:
    """Unbundle the gfx950 code object (an ELF) of ``lib`` into ``out``."""
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", f"--output={out}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], check=True)
    return out


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def kernel_resources(lib: str = DEFAULT_LIB) -> dict:
    with tempfile.TemporaryDirectory() as d:
        co = code_object(lib, os.path.join(d, "gfx950.co"))
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    rows = {}
    for blk in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)

        def g(k):
            return int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1))

        rows[name] = {"vgpr": g("vgpr_count"), "agpr": int(blk.split()[0]), "sgpr": g("sgpr_count"),
                      "scratch": g("private_segment_fixed_size"), "lds": g("group_segment_fixed_size")}
    dem = demangle(list(rows))
    return {dem[k]: v for k, v in rows.items()}


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else DEFAULT_LIB
    needles = [a for a in sys.argv[1:] if not a.endswith(".so")]
    for dem, r in kernel_resources(lib).items():
        if needles and not any(n in dem for n in needles):
            continue
        print(f"{dem[:100]:100s} vgpr {r['vgpr']:3d} agpr {r['agpr']:3d} sgpr {r['sgpr']:3d} scratch {r['scratch']:4d} B  lds {r['lds']:6d} B")
