#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv as: kernel, calls, average microseconds."""
import csv
import sys

for r in csv.DictReader(open(sys.argv[1])):
    print(r["Name"][:60].ljust(60), r["Calls"].rjust(4), f'{float(r["AverageNs"]) / 1e3:9.1f} us')
