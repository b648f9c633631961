for r in 1 2 3; do
for a in "" "--no-kernel-timing"; do
ms=$(timeout 300 python bench.py --no-alt-math --no-cpu-baseline --no-serialised-leg $a 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
echo "round $r [$a] $ms"
done; done
