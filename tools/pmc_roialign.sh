#!/bin/bash
# L1 / L2 counters of the ROIAlign kernels (north_star: ">= 40 % HBM roofline" -- or the counters that say which roof they sit under).
# Separate --pmc passes over tools/microbench.py's HBM-kernel section (2048 RoIs on a 4 x 38 x 63 x 1024 map, all bins and even bins);
# TCC has 4 counter slots per pass (FETCH_SIZE takes 3, WRITE_SIZE 2: MI355X_MICROARCH.md "rocprofv3 PMC slots").
# Run on the GPU box from the repo root:  bash tools/pmc_roialign.sh  ->  gpurun_out/pmc_roialign.json  (copy to profiles/rNN_pmc_roialign.json)
export TMPDIR=/tmp
R=$(pwd)
PASS_A="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"
PASS_B="FETCH_SIZE GRBM_GUI_ACTIVE"
PASS_C="WRITE_SIZE TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"
PASS_D="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"
mkdir -p gpurun_out
for P in A B C D; do
  rm -rf gpurun_out/pmc_roi_$P
  eval C=\$PASS_$P
  ( cd /tmp && timeout 240 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_roi_$P -o m -- python3 $R/tools/microbench.py --only nothing > $R/gpurun_out/pmc_roi_$P.log 2>&1 )
done
python3 - <<'PY'
import collections, csv, glob, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_roi_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        n = n[:n.index("(")] if "(" in n else n
        if "roi_" in n:
            agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"command": "tools/pmc_roialign.sh: rocprofv3 --pmc <pass> -- python3 tools/microbench.py --only nothing",
       "notes": "per-launch means over the microbenchmark's launches (all-bin and even-bin calls pooled per kernel name); GRBM_GUI_ACTIVE is summed over the 8 XCDs; "
                "FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them (gfx950: FETCH_SIZE under-reports wide reads by 2x, MI355X_MICROARCH.md)",
       "kernels": {}}
for k, v in sorted(agg.items()):
    a = {c: sum(x) / len(x) for c, x in v.items()}
    row = {c: round(x) for c, x in a.items()}
    cyc = a.get("GRBM_GUI_ACTIVE", 0) / 8.0
    row["launches_seen"] = max(len(x) for x in v.values())
    if a.get("TCC_HIT_sum") is not None and a.get("TCC_MISS_sum") is not None and (a["TCC_HIT_sum"] + a["TCC_MISS_sum"]) > 0:
        row["l2_hit_rate"] = round(a["TCC_HIT_sum"] / (a["TCC_HIT_sum"] + a["TCC_MISS_sum"]), 4)
    if cyc and a.get("TCC_REQ_sum"):
        row["l2_requests_per_cycle"] = round(a["TCC_REQ_sum"] / cyc, 2)
        row["l2_bytes_per_cycle_if_128B_requests"] = round(a["TCC_REQ_sum"] * 128 / cyc, 1)
    if a.get("TCP_TOTAL_CACHE_ACCESSES_sum") and a.get("TCP_TCC_READ_REQ_sum"):
        row["l1_miss_share(TCP->TCC reads / TCP accesses)"] = round(a["TCP_TCC_READ_REQ_sum"] / a["TCP_TOTAL_CACHE_ACCESSES_sum"], 4)
    if a.get("SQ_WAVE_CYCLES") and a.get("SQ_WAIT_ANY"):
        row["waves_parked_share"] = round(a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"], 4)
    out["kernels"][k] = row
    print(k, json.dumps(row))
json.dump(out, open("gpurun_out/pmc_roialign.json", "w"), indent=1)
PY
