#!/usr/bin/env python3
"""Forward / backward windows of the LAST training step in a rocprofv3 --kernel-trace CSV: per-category busy time on the main queue,
idle gaps, the tiny launches.  python tools/trace_windows.py <b_kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sgd = [i for i, r in enumerate(rows) if "sgd_kernel" in r["Kernel_Name"]]


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n[:n.index("(")] if "(" in n else n


seg = rows[sgd[-2] + 1: sgd[-1] + 1]
t0 = int(seg[0]["Start_Timestamp"])
byq = collections.defaultdict(list)
for r in seg:
    byq[r["Queue_Id"]].append(r)
main_q = max(byq, key=lambda q: len(byq[q]))
wg_q = max((q for q in byq if q != main_q), key=lambda q: sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in byq[q]))
bw0 = min(int(r["Start_Timestamp"]) for r in seg if "conv_wgrad" in r["Kernel_Name"])   # backward starts with the first weight gradient
print(f"step {(int(seg[-1]['End_Timestamp']) - t0) / 1e6:.2f} ms; forward window {(bw0 - t0) / 1e6:.2f} ms; backward window {(int(seg[-1]['End_Timestamp']) - bw0) / 1e6:.2f} ms")
for name, sel in (("forward", lambda r: int(r["Start_Timestamp"]) < bw0), ("backward", lambda r: int(r["Start_Timestamp"]) >= bw0)):
    rs = [r for r in byq[main_q] if sel(r)]
    cat = collections.defaultdict(lambda: [0, 0])
    prev, gaps = None, []
    for r in rs:
        n = short(r["Kernel_Name"])
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        k = "conv gemm" if "conv_igemm" in n or "conv_wgrad" in n else "wino transforms" if "wino" in n else \
            "aten/copy" if (n.startswith("at::") or "rocclr" in n or "elementwise_kernel_with_index" in n) else "abr: " + n[:30]
        cat[k][0] += 1
        cat[k][1] += d
        s = int(r["Start_Timestamp"])
        if prev is not None and s > prev:
            gaps.append(s - prev)
        prev = max(prev or 0, int(r["End_Timestamp"]))
    print(f"-- {name}: main queue {len(rs)} kernels, gaps {sum(gaps) / 1e6:.3f} ms ({sum(1 for g in gaps if g > 10000)} > 10 us = {sum(g for g in gaps if g > 10000) / 1e6:.3f} ms)")
    for k, (n, d) in sorted(cat.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"   {d / 1e6:7.3f} ms {n:4d}  {k}")
for q, rs in byq.items():
    if q != main_q:
        print(f"queue {q}: {len(rs)} kernels, busy {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs) / 1e6:.2f} ms, "
              f"{(int(rs[0]['Start_Timestamp']) - t0) / 1e6:.2f} -> {(int(rs[-1]['End_Timestamp']) - t0) / 1e6:.2f} ms")
