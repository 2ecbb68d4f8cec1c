#!/usr/bin/env python3
"""Pivot rocprofv3 counter_collection.csv files.

    tools/pmc_summary.py DIR...                                   one line per (pass, kernel, dispatch) with all counters
    tools/pmc_summary.py DIR... --kernel SUBSTR [--grid N] --json OUT [--source TEXT]
        median over the matching dispatches of every counter -> OUT (what bench.py reads for roofline.executed /
        roofline.traffic).  HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes): gfx950 tallies 128-B
        read requests at 64 B (MI355X_MICROARCH.md, HBM section); the two come from separate passes.
        The JSON also records WHICH BINARY was measured: `code_hash` = sha256 of the device code of --lib (default: the
        in-tree libptrace.so, or $PTRACE_LIB; pytracer_amd.build.code_hash) and the kernel's registers / scratch / LDS
        from the code object (tools/kres.py); bench.py prices a roofline from the file only when the hash is the loaded
        library's.
"""
import argparse
import csv
import glob
import json
import os
import statistics
import sys
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

ap = argparse.ArgumentParser()
ap.add_argument("dirs", nargs="+")
ap.add_argument("--kernel")
ap.add_argument("--grid", type=int)
ap.add_argument("--json")
ap.add_argument("--source", default="")
ap.add_argument("--lib", default=os.environ.get("PTRACE_LIB") or os.path.join(ROOT, "pytracer_amd", "libptrace.so"))
args = ap.parse_args()

rows = OrderedDict()
for d in args.dirs:
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if "rocclr" in name or "sum_counts" in name or "prep_hoist" in name:
                continue
            key = (d.rstrip("/").split("/")[-1], name, r["Dispatch_Id"], r["Grid_Size"], r["VGPR_Count"], r["SGPR_Count"])
            rows.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
            rows[key]["dur_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3

if not args.json:
    for k, v in rows.items():
        print((k[0], k[1][:48]) + k[2:])
        print("    " + "  ".join(f"{a}={b:.4g}" for a, b in v.items()))
    raise SystemExit(0)

sel = [(k, v) for k, v in rows.items() if args.kernel in k[1] and (args.grid is None or int(k[3]) == args.grid)]
if not sel:
    raise SystemExit(f"no dispatch of a kernel matching {args.kernel!r} (grid {args.grid}) under {args.dirs}")
values = {}
for _, v in sel:
    for c, x in v.items():
        values.setdefault(c, []).append(x)
counters = {c: statistics.median(xs) for c, xs in values.items() if c != "dur_us"}
out = {
    "kernel": sel[0][0][1], "grid_size": int(sel[0][0][3]), "vgprs": int(sel[0][0][4]), "sgprs": int(sel[0][0][5]),
    "dispatches": len(sel), "dur_us": statistics.median(values["dur_us"]),
    "counters": counters,
    "source": args.source or ("rocprofv3 --pmc passes under " + ", ".join(args.dirs)),
}
try:  # which binary these counters belong to
    from pytracer_amd.build import code_hash

    out["code_hash"] = code_hash(args.lib)
    out["lib"] = os.path.relpath(args.lib, ROOT) if args.lib.startswith(ROOT) else args.lib
except (OSError, ValueError) as e:
    out["code_hash"] = None
    out["code_hash_error"] = str(e)
try:
    import kres

    res = kres.kernel_resources(args.lib)
    want = out["kernel"].replace("void ", "").strip()
    match = [v for k, v in res.items() if k.replace("void ", "").strip() == want]
    out["kernel_resources"] = match[0] if match else None
except Exception as e:  # noqa: BLE001  (llvm tools missing: the hash alone ties the file to the binary)
    out["kernel_resources"] = None
    out["kernel_resources_error"] = f"{type(e).__name__}: {e}"[:200]
if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
    out["fetch_size_kb_raw"] = counters["FETCH_SIZE"]
    out["write_size_kb_raw"] = counters["WRITE_SIZE"]
    out["hbm_bytes_per_launch"] = int(round((2.0 * counters["FETCH_SIZE"] + counters["WRITE_SIZE"]) * 1024))
with open(args.json, "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out))
