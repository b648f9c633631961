#!/usr/bin/env python3
"""Timeline of the LAST training step of a rocprofv3 --kernel-trace CSV: every non-GEMM kernel with its start (ms from the step's first kernel),
duration and queue; the conv / Winograd / weight-preparation kernels in between are summarised per queue as (first start, last end, count).
python tools/step_timeline.py <b_kernel_trace.csv> [t_from_ms t_to_ms]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sgd = [i for i, r in enumerate(rows) if "sgd_kernel" in r["Kernel_Name"]]
seg = rows[sgd[-2] + 1: sgd[-1] + 1]
t0 = int(seg[0]["Start_Timestamp"])
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return (n[:n.index("(")] if "(" in n else n)[:44]


agg = {}
for r in seg:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    if e < lo or s > hi:
        continue
    n, q = short(r["Kernel_Name"]), r["Queue_Id"]
    if any(x in n for x in ("conv_", "wino_", "x6_pack", "dgrad_weights", "wgrad_reduce")):
        a = agg.setdefault(q, [s, e, 0, 0.0])
        a[1] = e; a[2] += 1; a[3] += e - s
    else:
        if agg:
            print("        GEMM-side:", "  ".join("q%s %.2f-%.2f (%d kernels, %.2f ms busy)" % (k, v[0], v[1], v[2], v[3]) for k, v in sorted(agg.items())))
            agg = {}
        print("%7.3f %7.3f ms  q%s  %s" % (s, e - s, q, n))
if agg:
    print("        GEMM-side:", "  ".join("q%s %.2f-%.2f (%d kernels, %.2f ms busy)" % (k, v[0], v[1], v[2], v[3]) for k, v in sorted(agg.items())))
print("step: %.3f ms" % ((int(seg[-1]["End_Timestamp"]) - t0) / 1e6))
