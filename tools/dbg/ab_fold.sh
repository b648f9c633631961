#!/bin/bash
# overlapped and folded-stream step time per setting
for r in 1 2; do
for e in "A=1" "ABR_X6_TILE=2"; do
  ms=$(env $e timeout 300 python bench.py --no-alt-math --no-cpu-baseline --no-kernel-timing --steps 40 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  msf=$(env $e timeout 300 python bench.py --no-alt-math --no-cpu-baseline --no-kernel-timing --no-serialised-leg --fold-streams --steps 40 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "round $r [$e] overlapped $ms folded $msf"
done; done
