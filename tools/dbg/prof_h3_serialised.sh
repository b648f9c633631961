export TMPDIR=/tmp
R=$(pwd)
mkdir -p gpurun_out; rm -rf gpurun_out/prof_ser
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ser -o b -- python3 $R/bench.py --math f16x3 --steps 5 --warmup 2 --no-cpu-baseline --no-alt-math --no-serialised-leg --fold-streams > $R/gpurun_out/prof_ser.jsonl 2> $R/gpurun_out/prof_ser.err )
python3 tools/kstats.py gpurun_out/prof_ser/b_kernel_stats.csv 7 45
