export TMPDIR=/tmp
R=$(pwd)
mkdir -p gpurun_out; rm -rf gpurun_out/prof_ovl
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ovl -o b -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt-math --no-serialised-leg --no-kernel-timing > $R/gpurun_out/prof_ovl.jsonl 2> $R/gpurun_out/prof_ovl.err )
python3 tools/gpu_idle.py gpurun_out/prof_ovl/b_kernel_trace.csv 2>&1 | tail -25
python3 tools/fill_timeline.py gpurun_out/prof_ovl/b_kernel_trace.csv 2>&1 | tail -30
cut -c1-160 gpurun_out/prof_ovl.jsonl
