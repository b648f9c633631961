#!/bin/bash
# Regenerates the round's evidence set on the GPU box (run from the repo root: bash tools/refresh_profiles.sh r04 [part]); every output lands in
# gpurun_out/refresh/ under its profiles/ name -- copy what is to be judged into profiles/.  Every step runs under `timeout`.
#   part 1: bench lines (headline, other geometries / tasks / arithmetics, per-rank B = 2 workloads), parity log, probes      (~8 min)
#   part 2: rocprofv3 kernel trace + PMC HBM traffic of the bench command, serialised-stream trace, per-shape tables, labs      (~10 min)
#   part 3: PMC matrix-pipe / LDS counters of the bench command (slow: counter collection replays every kernel)                (~15 min)
R=${1:-rXX}
PART=${2:-all}
O=gpurun_out/refresh
mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-alt-math"
if [ "$PART" = all ] || [ "$PART" = 1 ]; then
timeout 600 python bench.py > $O/${R}_bench_default.jsonl 2> $O/bench_default.err
timeout 300 python bench.py --image-size 800x1333 --steps 10 > $O/${R}_bench_800x1333.jsonl 2>/dev/null
timeout 300 python bench.py --task 10-5 --mosaic-squares --steps 20 > $O/${R}_bench_10-5_mosaic_squares.jsonl 2>/dev/null
timeout 300 python bench.py --task 10-5 --mosaic-squares --math bf16 --steps 20 --no-kernel-timing > $O/${R}_bench_10-5_mosaic_squares_bf16_backbone.jsonl 2>/dev/null
timeout 300 python bench.py --batch-pool 1 $B > $O/${R}_bench_single_batch.jsonl 2>/dev/null
timeout 300 python bench.py --share-frozen-prefix $B > $O/${R}_bench_shared_frozen_prefix.jsonl 2>/dev/null
timeout 300 python bench.py --math bf16x6 $B > $O/${R}_bench_bf16x6.jsonl 2>/dev/null
timeout 300 python tools/bench_eval.py 2>/dev/null | tail -1 > $O/${R}_bench_eval.jsonl
# the per-rank workloads of BASELINE configs[3] / configs[4] (8 ranks x batch 2) on ONE GPU: what the first scaling run is divided by
timeout 300 python bench.py --task 10-10 --batch-per-gpu 2 $B > $O/${R}_bench_10-10_b2.jsonl 2>/dev/null
timeout 300 python bench.py --task 10-5 --batch-per-gpu 2 --mosaic-squares --steps 20 > $O/${R}_bench_10-5_b2_mosaic_squares.jsonl 2>/dev/null
# main-stream timeline of the un-profiled step (events, no tracer) at B = 4 and B = 2
( timeout 200 python tools/step_marks.py; timeout 200 python tools/step_marks.py --batch-per-gpu 2 ) 2>&1 | grep -v amdgpu > $O/${R}_step_marks.txt
# is the host the bottleneck?  issue time against device time per batch size, and the host floor (every kernel short)
( timeout 300 python tools/dbg/host_time.py 2>&1 | grep "B ="; FLOOR_PROFILE=0 timeout 200 python tools/dbg/host_floor.py 2>&1 | grep floor ) > $O/${R}_host_time.txt
( echo "# tools/dbg/fc_time.py: predictor FCs, all tiles split along K (default), then ABR_IGEMM_FC_SPLIT=0"; timeout 200 python tools/dbg/fc_time.py 2>&1 | grep " x "
  ABR_IGEMM_FC_SPLIT=0 timeout 200 python tools/dbg/fc_time.py 2>&1 | grep " x " ) > $O/${R}_fc_split_k.txt
# gates of the arithmetic switch (round 5): conv fuzz in every arithmetic, soak (bit-identical revisits), unusual batches
( timeout 700 python tools/conv_fuzz.py --cases 2000 2>&1 | grep -v amdgpu | tail -15 ) > $O/${R}_conv_fuzz_2000.txt
( timeout 1200 python tools/detect_fuzz.py --cases 1000 2>&1 | grep -v amdgpu | tail -12 ) > $O/${R}_detect_fuzz_1000.txt
( timeout 400 python tools/soak.py --steps 600 2>&1 | grep -v amdgpu ) > $O/${R}_soak_600_steps.txt
( timeout 300 python tools/edge_steps.py 2>&1 | grep -v amdgpu | tail -15 ) > $O/${R}_edge_steps.txt
( timeout 300 python -m pytest tests/test_gpu_f16x3_admission.py tests/test_gpu_x6_admission.py -q -s -p no:cacheprovider 2>&1 | grep -E "ulp|passed|failed|inspected" | grep -v amdgpu ) > $O/${R}_admission_f16x3_and_bf16x6.txt
# parity evidence: the full-size golden / oracle comparisons with their measured numbers (losses, proposal match, worst max-rel / rel-L2 of the 52
# gradients per configuration and arithmetic) -- with -q alone the printed worst values are lost
( timeout 1500 python -m pytest tests/test_gpu_e2e_full_golden.py tests/test_gpu_e2e.py tests/test_gpu_e2e_golden.py tests/test_gpu_configs4_whole.py -q -s -p no:cacheprovider 2>&1 \
    | grep -E "^\[|worst|max-rel|l2-rel|passed|failed|^GPU |^oracle|proposal|present|distance" | grep -v amdgpu.ids ) > $O/${R}_fullsize_parity.log
( echo "# tools/topk_probe.py: proposal ranking alone (multi-workgroup phases, then ABR_TOPK_ONE_WG=1 = round 3's one workgroup per image)"
  timeout 120 python tools/topk_probe.py 2>&1 | grep -v amdgpu; ABR_TOPK_ONE_WG=1 timeout 120 python tools/topk_probe.py 2>&1 | grep -v amdgpu ) > $O/${R}_topk_probe.txt
fi
if [ "$PART" = all ] || [ "$PART" = 2 ]; then
# rocprofv3 kernel trace of the bench command (+ the PMC HBM-traffic passes)
STEPS=5 WARMUP=2 timeout 900 bash tools/profile_bench.sh > $O/profile_bench.log 2>&1
cp gpurun_out/prof_kernel_stats.csv $O/${R}_bench_kernel_stats.csv
cp gpurun_out/prof_kernel_stats.json $O/${R}_bench_kernel_stats.json
cp gpurun_out/prof_bench.jsonl $O/${R}_bench_under_rocprof.jsonl
[ -f gpurun_out/pmc_traffic.json ] && cp gpurun_out/pmc_traffic.json $O/${R}_pmc_traffic.json
# every stream folded into one: the kernels' own durations as a kernel trace sees them
rm -rf gpurun_out/prof_ser
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof_ser -o s -- python3 $OLDPWD/bench.py --steps 5 --warmup 2 $B --no-serialised-leg --no-single-batch-leg --fold-streams > $OLDPWD/$O/${R}_serialised_streams_bench.jsonl 2>/dev/null )
cp gpurun_out/prof_ser/s_kernel_stats.csv $O/${R}_serialised_streams_kernel_stats.csv 2>/dev/null
# per-shape table, HBM-bound kernels, ROIAlign L1 / L2 counters, main-loop labs
timeout 600 python tools/conv_breakdown.py --target-tf 400 2>&1 | grep -v amdgpu.ids > $O/${R}_conv_shapes_f16x3.txt
timeout 600 python tools/conv_breakdown.py --batch 2 --target-tf 400 2>&1 | grep -v amdgpu.ids > $O/${R}_conv_shapes_f16x3_b2.txt
timeout 600 python tools/microbench.py --only nothing 2>/dev/null > $O/${R}_microbench_hbm_kernels.txt
timeout 600 bash tools/pmc_roialign.sh > $O/pmc_roialign.log 2>&1; [ -f gpurun_out/pmc_roialign.json ] && cp gpurun_out/pmc_roialign.json $O/${R}_pmc_roialign.json
( echo "# tools/dbg/wino_time.sh: rocprofv3 kernel stats of the Winograd path, one 3x3 shape of the step per run (20 calls each)"; timeout 900 bash tools/dbg/wino_time.sh 2>&1 ) > $O/${R}_winograd_path_kernels.txt
( echo "# tools/x6lab/hlab.hip: two-term fp16 split with three products (f16x3) in the weights-direct loop vs the library bf16x6 loop; error vs float64 at 2048 outputs; two timing rounds"
  timeout 250 tools/x6lab/hlab ) > $O/${R}_x6lab_f16x3.txt 2>&1
fi
if [ "$PART" = all ] || [ "$PART" = 3 ]; then
timeout 1500 bash tools/pmc_mfma.sh > $O/pmc_mfma.log 2>&1
[ -f gpurun_out/pmc_mfma.json ] && cp gpurun_out/pmc_mfma.json $O/${R}_pmc_mfma.json
fi
ls -la $O
