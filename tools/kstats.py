#!/usr/bin/env python3
"""Per-step summary of a rocprofv3 --kernel-trace --stats CSV:  python tools/kstats.py <b_kernel_stats.csv> <steps> [top]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6 / steps:.2f} ms/step over {len(rows)} kernels")
for r in rows[:top]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    n = n[:n.index("(")] if "(" in n else n
    print(f"{int(r['TotalDurationNs']) / 1e6 / steps:8.3f} ms/step {int(r['Calls']) / steps:7.1f} calls {float(r['AverageNs']) / 1e3:9.1f} us {float(r['Percentage']):5.1f}%  {n[:90]}")
