#!/bin/bash
# Own rate (every stream of the step folded into one) of the bf16x6 kernels under different environment settings, inside ONE GPU session:
#   tools/ser_rates.sh "VAR=a" "VAR=b" ...      ("-" = no setting; ABR_IOD_HIP_LIB=/path/to/other/libabr_iod_hip.so compares two builds)
for e in "$@"; do
 if [ "$e" = "-" ]; then ee=""; else ee="$e"; fi
 env $ee timeout 300 python bench.py --no-alt-math --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['roofline']['serialised']['kernels']
print('[$e]', 'step', d['ms_per_step'], 'ms; serialised', d['roofline']['serialised']['ms_per_step'], 'ms;', {n[:30]:(v['avg_launch_ms'],v['achieved']) for n,v in k.items() if 'x6' in n})
"
done
