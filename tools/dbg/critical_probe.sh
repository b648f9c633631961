#!/bin/bash
# Is a chain of the step on its critical path?  A spin kernel (torch.cuda._sleep) in front of (a) the target's proposal selection on its side stream,
# (b) the RoI targets on the main stream; same-session A/B of the step time.  A chain with slack absorbs the delay, a critical one passes it on 1:1.
python - <<'P' 2>&1 | grep -v amdgpu
import torch
for c in (100000, 200000, 400000):
    torch.cuda._sleep(c); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); torch.cuda._sleep(c); b.record(); torch.cuda.synchronize()
    print("torch.cuda._sleep(%d) = %.3f ms" % (c, a.elapsed_time(b)))
P
STEPS=40 bash tools/ab_env.sh 2 "-" "ABR_DBG_SLEEP_PROPOSALS=${1:-200000}" "ABR_DBG_SLEEP_ROI_TARGETS=${1:-200000}"
