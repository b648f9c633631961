#!/bin/bash
# A/B of library builds inside ONE GPU session (boxes differ by several percent): tools/ab_bench.sh <rounds> <lib.so> [<lib.so> ...]
# ("cur" = the in-tree build).  Prints ms/step per build and round; the builds alternate within a round.
R=${1:-2}; shift
for r in $(seq 1 $R); do
  for l in "$@"; do
    if [ "$l" = cur ]; then unset ABR_IOD_HIP_LIB; else export ABR_IOD_HIP_LIB=$(pwd)/$l; fi
    ms=$(timeout 300 python bench.py --no-alt-math --no-cpu-baseline --no-kernel-timing --steps ${STEPS:-40} 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "round $r $l $ms"
  done
done
