#!/usr/bin/env python3
"""Registers, scratch and LDS of every kernel in libptrace.so (from the gfx950 code object's metadata).

    python tools/kres.py [path/to/libptrace.so] [substring ...]
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pytracer_amd", "libptrace.so")
needles = [a for a in sys.argv[1:] if not a.endswith(".so")]
with tempfile.TemporaryDirectory() as d:
    fat = os.path.join(d, "fat.bin")
    subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    co = os.path.join(d, "gfx950.co")
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", f"--output={co}",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], check=True)
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
for blk in notes.split("- .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if needles and not any(n in dem for n in needles):
        continue
    g = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1))
    print(f"{dem[:100]:100s} vgpr {g('vgpr_count'):3d} agpr {int(blk.split()[0]):3d} sgpr {g('sgpr_count'):3d} scratch {g('private_segment_fixed_size'):4d} B  lds {g('group_segment_fixed_size'):6d} B")
