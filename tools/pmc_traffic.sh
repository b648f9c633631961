#!/bin/bash
# HBM traffic of the conv kernels from PMC counters: two separate passes (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2).
# Run on the GPU box from the repo root; writes gpurun_out/pmc_traffic.json
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_$c -o b -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-math --no-kernel-timing --no-serialised-leg --no-single-batch-leg > gpurun_out/pmc_$c.log 2>&1
done
python3 - <<'PY'
import csv, collections, json
raw = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f"gpurun_out/pmc_{c}/b_counter_collection.csv")):
        if r["Counter_Name"] == c:
            n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            n = n[:n.index("(")] if "(" in n else n
            agg[n].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "conv" in k or "roi_align" in k or "ard" in k or "sgd" in k or "wino" in k:
            raw.setdefault(k, {})[c] = (len(v), sum(v) / len(v))
out = {"command": "tools/pmc_traffic.sh : rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-math --no-kernel-timing --no-serialised-leg --no-single-batch-leg",
       "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B for wide (16 B/lane) coalesced reads -> x2 (MI355X_MICROARCH.md 'HBM'); "
                     "unit KB; WRITE_SIZE as reported. Calibrated on ard_bwd_kernel (algorithmic 102.8 MB read / 51.4 MB written at B=4).",
       "kernels": {}}
for k, v in raw.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        f, w = v["FETCH_SIZE"][1], v["WRITE_SIZE"][1]
        out["kernels"][k] = {"launches": v["FETCH_SIZE"][0], "fetch_size_KB_raw": round(f, 1), "write_size_KB": round(w, 1),
                             "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
        print(k[:60], round((2 * f + w) / 1024, 1), "MB/launch (2*FETCH + WRITE)")
json.dump(out, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
PY
