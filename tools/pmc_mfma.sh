#!/bin/bash
# Matrix-pipe / LDS / wait counters of the step's conv kernels (north_star: "each choice evidenced by rocprof ... MFMA-busy").
# Two separate --pmc passes over the bench command (8 SQ slots per pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"); counter collection
# serialises the kernels, so these are each kernel's OWN figures, not the time-shared ones of the overlapped step.
# Run on the GPU box from the repo root:  bash tools/pmc_mfma.sh  ->  gpurun_out/pmc_mfma.json  (copy to profiles/rNN_pmc_mfma.json)
export TMPDIR=/tmp
R=$(pwd)
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-alt-math --no-kernel-timing --no-serialised-leg --no-single-batch-leg"
PASS_A="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
PASS_B="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
mkdir -p gpurun_out
for P in A B; do
  rm -rf gpurun_out/pmc_mfma_$P
  eval C=\$PASS_$P
  ( cd /tmp && rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_mfma_$P -o m -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_mfma_$P.log 2>&1 )
done
PASS_A="$PASS_A" PASS_B="$PASS_B" ARGS="$ARGS" python3 - <<'PY'
import collections, csv, glob, json, os
def load(tag):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    files = glob.glob(f"gpurun_out/pmc_mfma_{tag}/**/*counter_collection.csv", recursive=True)
    for f in files:
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            n = n[:n.index("(")] if "(" in n else n
            agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg
A, B = load("A"), load("B")
out = {"command": "tools/pmc_mfma.sh: rocprofv3 --pmc <pass> -- python3 bench.py " + os.environ["ARGS"],
       "pass_A": os.environ["PASS_A"].split(), "pass_B": os.environ["PASS_B"].split(),
       "units": "SQ_*_CYCLES of waves (WAVE / WAIT / ACTIVE) count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the "
                "1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs.  mfma_busy = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024): the share of "
                "SIMD-cycles the matrix pipe was busy while the kernel ran (kernels run alone under counter collection).",
       "kernels": {}}
def mean(v): return sum(v) / len(v) if v else None
for k in sorted(A):
    if not any(t in k for t in ("conv_", "wino_", "roi_align", "ard_", "sgd_")):
        continue
    a = {c: mean(v) for c, v in A[k].items()}
    b = {c: mean(v) for c, v in B.get(k, {}).items()}
    if not a.get("GRBM_GUI_ACTIVE") or not a.get("SQ_WAVE_CYCLES"):
        continue
    cyc = a["GRBM_GUI_ACTIVE"] / 8.0
    wc = a["SQ_WAVE_CYCLES"]
    row = {"launches": len(A[k]["GRBM_GUI_ACTIVE"]), "kernel_cycles": round(cyc),
           "mfma_busy": round(a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024.0), 4),
           "sq_busy": round(a.get("SQ_BUSY_CYCLES", 0.0) / a["GRBM_GUI_ACTIVE"], 4) if a.get("SQ_BUSY_CYCLES") else None,
           "wave_cycles_share": {"wait_any(parked: waitcnt/barrier)": round(a.get("SQ_WAIT_ANY", 0.0) / wc, 4),
                                 "wait_inst_any(issue stall)": round(a.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4),
                                 "wait_inst_lds(of which LDS issue)": round(a.get("SQ_WAIT_INST_LDS", 0.0) / wc, 4),
                                 "active_inst_any": round(a.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 4)}}
    if b:
        row["per_launch"] = {c: round(v) for c, v in b.items() if c.startswith("SQ_INSTS") or c.startswith("SQ_LDS")}
        if b.get("SQ_LDS_IDX_ACTIVE"):
            row["lds_bank_conflict_share"] = round(b.get("SQ_LDS_BANK_CONFLICT", 0.0) / b["SQ_LDS_IDX_ACTIVE"], 4)
    out["kernels"][k] = row
    print(f"{k[:64]:64s} n={row['launches']:4d} mfma_busy={row['mfma_busy']:.3f} wait_any={row['wave_cycles_share']['wait_any(parked: waitcnt/barrier)']:.3f} "
          f"wait_inst={row['wave_cycles_share']['wait_inst_any(issue stall)']:.3f} lds_issue={row['wave_cycles_share']['wait_inst_lds(of which LDS issue)']:.3f}")
json.dump(out, open("gpurun_out/pmc_mfma.json", "w"), indent=1)
PY
