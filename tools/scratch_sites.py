#!/usr/bin/env python3
"""Where a kernel's scratch (spill) instructions sit relative to its loops: for every scratch_load / scratch_store of the
kernel's gfx950 disassembly, the number of loops (back edges) that enclose it and the size of the innermost one.

    python tools/scratch_sites.py "pt_path_tree_kernel<true, true>" [--lib pytracer_amd/libptrace.so]

A spill in a block that runs once per pixel or per sample costs nothing measurable; one inside the per-round or per-step loop is
in the dependent chain.  Loop = a branch to a lower address; nesting by address ranges (reducible code)."""
import argparse
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kres  # noqa: E402


def sites(kernel: str, lib: str = kres.DEFAULT_LIB):
    import tempfile

    with tempfile.TemporaryDirectory() as d:
        co = kres.code_object(lib, os.path.join(d, "gfx950.co"))
        dis = subprocess.run([f"{kres.LLVM}/llvm-objdump", "-d", "-C", co], capture_output=True, text=True).stdout
    out = []
    for f in re.split(r"\n(?=[0-9a-f]+ <[^\n]+>:\n)", dis):
        m = re.match(r"[0-9a-f]+ <(.+)>:\n", f)
        if not m or kernel not in m.group(1):
            continue
        ins = []
        for line in f.split("\n")[1:]:
            mm = re.match(r"\s*(\S.*?)\s+// ([0-9A-F]+): \S+(?: \S+)*?(?: <.*\+0x([0-9a-f]+)>)?$", line)
            if mm:
                ins.append((int(mm.group(2), 16), mm.group(1), int(mm.group(3), 16) if mm.group(3) else None))
        base = ins[0][0]
        loops = [(base + tgt, addr) for addr, text, tgt in ins if text.startswith(("s_cbranch", "s_branch")) and tgt is not None and base + tgt <= addr]
        rows = []
        for addr, text, _ in ins:
            if "scratch_" in text:
                enc = sorted(b - a for a, b in loops if a <= addr <= b)
                rows.append((addr - base, text.split()[0], len(enc), enc[0] if enc else 0))
        out.append((m.group(1), len(ins), len(loops), rows, sorted(((b - a), a - base, b - base) for a, b in loops)[-8:]))
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("kernel")
    ap.add_argument("--lib", default=kres.DEFAULT_LIB)
    args = ap.parse_args()
    for name, n, nl, rows, big in sites(args.kernel, args.lib):
        print(f"{name}: {n} instructions, {nl} back edges, {len(rows)} scratch instructions")
        print("  largest loops (bytes, from, to):", [(s, hex(a), hex(b)) for s, a, b in big])
        hist = {}
        for off, op, depth, inner in rows:
            hist.setdefault((depth, inner), []).append((off, op))
        for (depth, inner), v in sorted(hist.items()):
            ld = sum(1 for _, op in v if "load" in op)
            print(f"  loop depth {depth}, innermost loop {inner:6d} B: {ld:3d} loads {len(v) - ld:3d} stores  at {hex(v[0][0])} .. {hex(v[-1][0])}")
