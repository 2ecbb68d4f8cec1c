#!/usr/bin/env python3
"""Pivot rocprofv3 counter_collection.csv files: one line per (kernel, dispatch) with all counters."""
import csv
import glob
import sys
from collections import OrderedDict

rows = OrderedDict()
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if "rocclr" in name or "sum_counts" in name or "prep_hoist" in name:
                continue
            key = (d.split("/")[-1], name[:40], r["Dispatch_Id"], r["Grid_Size"], r["VGPR_Count"], r["SGPR_Count"])
            rows.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
            rows[key]["dur_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, v in rows.items():
    print(k)
    print("    " + "  ".join(f"{a}={b:.4g}" for a, b in v.items()))
