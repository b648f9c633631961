#!/usr/bin/env python3
"""Device idle time per training step from a rocprofv3 --kernel-trace CSV: union of the kernel intervals of ALL queues between the
first and the last sgd_kernel; the largest gaps with the kernels on either side.  python tools/gpu_idle.py <b_kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return (n[:n.index("(")] if "(" in n else n)[:48]


sgd = [i for i, r in enumerate(rows) if "sgd_kernel" in r["Kernel_Name"]]
steps = len(sgd) - 1
seg = rows[sgd[0] + 1: sgd[-1] + 1]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
busy, end, last = 0, t0, None
gaps = []
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > end:
        gaps.append((s - end, short(last["Kernel_Name"]) if last else "-", short(r["Kernel_Name"])))
        busy += 0
        cur = s
    else:
        cur = end
    if e > cur:
        busy += e - cur
        if e > end:
            end, last = e, r
wall = t1 - t0
print(f"{steps} steps, {wall / steps / 1e6:.2f} ms/step wall, device busy {busy / steps / 1e6:.2f} ms/step, idle {(wall - busy) / steps / 1e6:.2f} ms/step "
      f"in {len(gaps) / steps:.0f} gaps/step; kernel launches/step {len(seg) / steps:.0f}")
aten = sum(1 for r in seg if r["Kernel_Name"].startswith("void at::") or r["Kernel_Name"].startswith("at::") or "rocclr" in r["Kernel_Name"])
print(f"ATen / runtime-copy launches per step: {aten / steps:.0f}")
gaps.sort(reverse=True)
for g, a, b in gaps[:12]:
    print(f"  {g / 1e3:8.1f} us  after {a}  before {b}")
hist = [0, 0, 0, 0]
for g, _, _ in gaps:
    hist[0 if g < 2000 else 1 if g < 10000 else 2 if g < 50000 else 3] += g
print("idle by gap size per step: <2us %.2f ms, 2-10us %.2f ms, 10-50us %.2f ms, >50us %.2f ms" % tuple(h / steps / 1e6 for h in hist))
