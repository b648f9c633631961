#!/usr/bin/env python3
"""How full is the chip over the LAST training step of a rocprofv3 --kernel-trace CSV?  Every running kernel contributes its wave demand
(workgroups x waves per workgroup, capped at what it can have resident: 8 waves per SIMD x 1024 SIMDs), the step is cut at every kernel start /
end, and the time is binned by total demand / 4096 waves (= one wave per SIMD... x4).  Lists the kernels that run while the demand is low.
python tools/fill_timeline.py <b_kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sgd = [i for i, r in enumerate(rows) if "sgd_kernel" in r["Kernel_Name"]]
seg = rows[sgd[-2] + 1: sgd[-1] + 1]


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return (n[:n.index("(")] if "(" in n else n)[:44]


ev = []
for r in seg:
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    waves = grid // 64 if grid >= 64 else 1
    r["_waves"] = min(waves, 8192)
    r["_wgs"] = max(1, grid // max(wg, 1))
    ev.append((int(r["Start_Timestamp"]), 1, r))
    ev.append((int(r["End_Timestamp"]), 0, r))
ev.sort(key=lambda e: (e[0], e[1]))
active = {}
bins = collections.Counter()
low = collections.Counter()
t_prev = ev[0][0]
for t, kind, r in ev:
    dt = t - t_prev
    if dt > 0:
        demand = sum(a["_waves"] for a in active.values())
        b = "idle" if not active else ("< 1024 waves (one per SIMD)" if demand < 1024 else ("1024-2047" if demand < 2048 else ("2048-4095" if demand < 4096 else ">= 4096")))
        bins[b] += dt
        if demand < 2048:
            for a in active.values():
                low[short(a["Kernel_Name"])] += dt
    if kind:
        active[id(r)] = r
    else:
        active.pop(id(r), None)
    t_prev = t
tot = sum(bins.values())
print("step %.3f ms (between the last two sgd_kernel launches)" % (tot / 1e6))
for b in ("idle", "< 1024 waves (one per SIMD)", "1024-2047", "2048-4095", ">= 4096"):
    print("  %-30s %7.3f ms  %5.1f %%" % (b, bins[b] / 1e6, 100.0 * bins[b] / tot))
print("kernels running while the demand is below 2048 waves (ms of that time each was active):")
for k, v in low.most_common(18):
    print("  %-46s %7.3f" % (k, v / 1e6))
