#!/bin/bash
# rocprofv3 kernel stats of the Winograd path, one 3x3 shape of the step per run (GPU box)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for s in layer2 layer3 layer4 rpn; do for wh in fwd wgrad; do
  rm -rf /tmp/wt; timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/wt -o wt --output-format csv -- python3 $R/tools/dbg/wino_time.py $s $wh 2>/dev/null | grep "MB"
  f=$(find /tmp/wt -name "*kernel_stats.csv" | head -1)
  echo "--- $s $wh"; python3 - "$f" <<'P'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if any(k in n for k in ('wino','conv_igemm','conv_wgrad','h3_','wgrad_reduce')):
        print("  %-70s calls %4s avg %8.1f us" % (n.replace('(anonymous namespace)::','')[:70], r['Calls'], float(r['AverageNs'])/1e3))
P
done; done
