#!/bin/bash
# Round profile: rocprofv3 kernel stats of the bench command + the PMC HBM-traffic passes.  Run on the GPU box from the repo
# root (gpurun -- 'bash tools/profile_bench.sh'); results land in gpurun_out/prof_*; copy the summaries into profiles/.
export TMPDIR=/tmp
R=$(pwd)
STEPS=${STEPS:-5}
WARMUP=${WARMUP:-2}
# --no-alt-math: without it bench.py appends 13 bf16x6 steps AFTER the timed region and every shared kernel's stats are polluted;
# --no-cpu-baseline: the oracle leg is host work.  The program itself follows `--` (no env / bash -c hop under the profiler).
ARGS="--steps $STEPS --warmup $WARMUP --no-cpu-baseline --no-alt-math --no-serialised-leg --no-single-batch-leg $EXTRA_ARGS"
mkdir -p gpurun_out
rm -rf gpurun_out/prof_stats
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o b -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_bench.jsonl 2> $R/gpurun_out/prof_bench.err )
STEPS=$STEPS WARMUP=$WARMUP ARGS="$ARGS" python3 - <<'PY'
import csv, json, os
rows = list(csv.DictReader(open("gpurun_out/prof_stats/b_kernel_stats.csv")))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n[:n.index("(")] if "(" in n else n
out = {"command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py " + os.environ["ARGS"],
       "steps_profiled": int(os.environ["STEPS"]) + int(os.environ["WARMUP"]), "total_kernel_ms": round(tot / 1e6, 2),
       "kernels": [{"kernel": short(r["Name"]), "calls": int(r["Calls"]), "total_ms": round(int(r["TotalDurationNs"]) / 1e6, 3),
                    "avg_us": round(float(r["AverageNs"]) / 1e3, 2), "pct": float(r["Percentage"])} for r in rows[:25]]}
json.dump(out, open("gpurun_out/prof_kernel_stats.json", "w"), indent=1)
print("total kernel ms", out["total_kernel_ms"], "top:", [(k["kernel"][:40], k["total_ms"]) for k in out["kernels"][:6]])
PY
cp gpurun_out/prof_stats/b_kernel_stats.csv gpurun_out/prof_kernel_stats.csv
bash tools/pmc_traffic.sh
tail -1 gpurun_out/prof_bench.jsonl | cut -c1-300
