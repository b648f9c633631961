#!/usr/bin/env python3
"""Print the headline numbers of a bench.py JSON line read from stdin."""
import json
import sys

for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r = d.get("roofline", {})
    print(d["value"], "img/s", d["ms_per_step"], "ms | dominant", r.get("kernel"), r.get("achieved"), "TF |",
          "overlapped", r.get("overlapped", {}).get("tflops"), "|",
          {k: (round(v["ms"] / d["steps"], 2), v["tflops"], round(v.get("overlapped_ms", 0) / d["steps"], 2), v.get("overlapped_tflops"))
           for k, v in r.get("all_conv_kernels", {}).items()})
