#!/bin/bash
# tools/dbg/critical_probe.sh on the per-rank workload of the 8-GPU configurations (B = 2, task 10-10)
for r in 1 2; do
  for e in "-" "ABR_DBG_SLEEP_ROI_TARGETS=1200000" "ABR_DBG_SLEEP_ROI_TARGETS=2400000" "ABR_DBG_SLEEP_PROPOSALS=2400000"; do
    if [ "$e" = "-" ]; then ee=""; else ee="$e"; fi
    ms=$(env $ee timeout 300 python bench.py --task 10-10 --batch-per-gpu 2 --no-alt-math --no-cpu-baseline --no-kernel-timing --steps 40 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "round $r [$e] $ms"
  done
done
