#!/bin/bash
# HBM traffic of the conv kernels from PMC counters: two separate passes (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2).
# Run on the GPU box from the repo root; writes gpurun_out/pmc_traffic.json
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_$c -o b -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/pmc_$c.log 2>&1
done
python3 - <<'PY'
import csv, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f"gpurun_out/pmc_{c}/b_counter_collection.csv")):
        if r["Counter_Name"] == c:
            n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            n = n[:n.index("(")] if "(" in n else n
            agg[n].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "conv" in k or "roi_align" in k or "ard" in k or "sgd" in k:
            out.setdefault(k, {})[c] = {"launches": len(v), "avg_per_launch_KB": sum(v) / len(v)}
json.dump(out, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
for k, v in out.items():
    print(k[:60], {c: round(x["avg_per_launch_KB"] / 1024, 2) for c, x in v.items()}, "MB/launch (raw counter)")
PY
