#!/bin/bash
# PMC passes over a lab binary (one GEMM shape): tools/x6lab/pmc_lab.sh ./lab7 32768 2048 1024
export TMPDIR=/tmp
SCRIPT=$(realpath $1); shift
OUT=$(pwd)/gpurun_out/pmc_py
rm -rf $OUT; mkdir -p $OUT
PASS_A="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
PASS_B="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
PASS_C="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"
for P in A B C; do
  eval C=\$PASS_$P
  ( cd /tmp && timeout 200 rocprofv3 --pmc $C --output-format csv -d $OUT/$P -o m -- python3 $SCRIPT "$@" > $OUT/$P.log 2>&1 )
done
python3 - <<PY
import collections, csv, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    a = {c: sum(x) / len(x) for c, x in v.items()}
    if "GRBM_GUI_ACTIVE" not in a or "SQ_WAVE_CYCLES" not in a: continue
    cyc = a["GRBM_GUI_ACTIVE"] / 8; wc = a["SQ_WAVE_CYCLES"]
    print("%-60s cyc %8.0f mfma_busy %.3f | of wave cycles: wait_any %.3f wait_inst %.3f (lds %.3f) active %.3f | valu insts %.3g lds insts %.3g vmem %.3g | lds conflict %.3f | act valu %.3f lds %.3f vmem %.3f" % (
        k, cyc, a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024), a.get("SQ_WAIT_ANY", 0) / wc, a.get("SQ_WAIT_INST_ANY", 0) / wc, a.get("SQ_WAIT_INST_LDS", 0) / wc,
        a.get("SQ_ACTIVE_INST_ANY", 0) / wc, a.get("SQ_INSTS_VALU", 0), a.get("SQ_INSTS_LDS", 0), a.get("SQ_INSTS_VMEM_RD", 0),
        a.get("SQ_LDS_BANK_CONFLICT", 0) / max(a.get("SQ_LDS_IDX_ACTIVE", 1), 1), a.get("SQ_ACTIVE_INST_VALU", 0) / wc, a.get("SQ_ACTIVE_INST_LDS", 0) / wc, a.get("SQ_ACTIVE_INST_VMEM", 0) / wc))
PY
