#!/bin/bash
# usage: tools/pmc_conv.sh <variant> <layer-filter>   (GPU box) -- per-kernel PMC summary for one conv microbench shape
export TMPDIR=/tmp
V=$1; L=$2
rm -rf gpurun_out/pmc_$V
ABR_CONV_VARIANT=$V rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_$V -o m -- python3 tools/microbench.py --only $L > gpurun_out/pmc_$V.log 2>&1
python3 - <<PY
import csv,collections
rows=list(csv.DictReader(open("gpurun_out/pmc_$V/m_counter_collection.csv")))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    if "igemm" in k:
        a={c: sum(x)/len(x) for c,x in v.items()}
        wc=a["SQ_WAVE_CYCLES"]
        print("VAR $V", k[28:70], "mfma_util=%.3f wait_any=%.3f wait_inst=%.3f active=%.3f valu_insts=%.3g gui=%.3g" % (a["SQ_VALU_MFMA_BUSY_CYCLES"]/(a["GRBM_GUI_ACTIVE"]/8*1024), a["SQ_WAIT_ANY"]/wc, a["SQ_WAIT_INST_ANY"]/wc, a["SQ_ACTIVE_INST_ANY"]/wc, a["SQ_INSTS_VALU"], a["GRBM_GUI_ACTIVE"]/8))
PY
