#!/bin/bash
# A/B of environment settings inside ONE GPU session: tools/ab_env.sh <rounds> "VAR=a" "VAR=b OTHER=c" ...   ("-" = no setting)
R=${1:-2}; shift
for r in $(seq 1 $R); do
  for e in "$@"; do
    if [ "$e" = "-" ]; then ee=""; else ee="$e"; fi
    ms=$(env $ee timeout 300 python bench.py --no-alt-math --no-cpu-baseline --no-kernel-timing --steps ${STEPS:-40} 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "round $r [$e] $ms"
  done
done
