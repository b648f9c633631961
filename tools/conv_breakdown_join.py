#!/usr/bin/env python3
"""Join the conv call sequence of the LAST training step (tools/conv_breakdown.py --seq) with a rocprofv3 --kernel-trace CSV of
the same process: the last len(seq) conv_igemm / conv_wgrad dispatches are that step's calls, in order.  Prints the per-shape
table with true kernel durations (no event / launch overhead)."""
import collections
import csv
import json
import sys


def main():
    seq = json.load(open(sys.argv[1]))
    target = float(sys.argv[3]) if len(sys.argv) > 3 else 125.0
    disp = {"igemm": [], "wgrad": []}
    with open(sys.argv[2]) as f:
        rows = sorted(csv.DictReader(f), key=lambda r: int(r["Start_Timestamp"]))
    for r in rows:
        n = r["Kernel_Name"]
        kind = "igemm" if "conv_igemm_kernel" in n else ("wgrad" if "conv_wgrad_kernel" in n else None)
        if kind:
            disp[kind].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), n, r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?")))
    rec = collections.OrderedDict()
    for kind in ("igemm", "wgrad"):
        calls = seq[kind]
        tail = disp[kind][-len(calls):]
        assert len(tail) == len(calls), (kind, len(tail), len(calls))
        for (key, fl), (ns, name, gx, wx) in zip(calls, tail):
            tile = name[name.find("<"):name.find(">") + 1] if "<" in name else ""
            wgs = int(gx) // max(int(wx), 1) if gx != "?" else 0
            r = rec.setdefault((kind,) + tuple(key) + (tile, wgs), [0, 0.0, fl])
            r[0] += 1
            r[1] += ns / 1e6
    out = []
    for key, (n, ms, fl) in rec.items():
        ideal = n * fl / target / 1e9
        out.append((ms - ideal, key, n, ms, fl * n / ms / 1e9))
    out.sort(reverse=True)
    print(f"total conv ms/step {sum(r[3] for r in out):.2f}; lost vs {target:.0f} TF: {sum(r[0] for r in out):.2f} ms")
    print(f"{'lost ms':>8s} {'ms':>7s} {'calls':>5s} {'TF/s':>6s}  kind   M       N     K     geometry            tile  WGs")
    for lost, key, n, ms, tf in out:
        print(f"{lost:8.3f} {ms:7.3f} {n:5d} {tf:6.1f}  {key[1]:6s} {key[2]:7d} {key[3]:5d} {key[4]:5d} {key[5]:18s} {key[6]:>14s} {key[7]:5d}")


if __name__ == "__main__":
    main()
