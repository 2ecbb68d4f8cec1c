#!/usr/bin/env python3
"""Per-kernel durations and the gaps between them from a rocprofv3 --kernel-trace CSV (the last N rows).

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/kbench.py c4:sample --rounds 3
    python3 tools/ktrace.py OUT [rows]
"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
prev = None
for r in rows[-(int(sys.argv[2]) if len(sys.argv) > 2 else 12):]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{r['Kernel_Name'][:56]:56s} grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8s} dur {(e - s) / 1e3:8.1f} us  gap before {gap:7.1f} us")
    prev = e
