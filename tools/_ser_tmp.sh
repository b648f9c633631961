for e in "$@"; do
 if [ "$e" = "-" ]; then ee=""; else ee="$e"; fi
 env $ee timeout 300 python bench.py --no-alt-math --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['roofline']['serialised']['kernels']
print('[$e]', d['ms_per_step'], d['roofline']['serialised']['ms_per_step'], {n[:24]:(v['avg_launch_ms'],v['achieved']) for n,v in k.items() if 'x6' in n})
"
done
