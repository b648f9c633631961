#!/usr/bin/env python3
"""Overlapped vs serial training steps (tests/test_gpu_streams_equivalence.py), many times inside ONE process so that allocator and
cache state accumulates: prints every pair whose losses disagree.  GPU box only."""
import os
import sys

import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import test_gpu_streams_equivalence as T  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = 0
for i in range(n):
    for width, ov in (("tiny", T.TINY), ("full", T.TINY[8:])):
        p_on, l_on = T._run(True, ov)
        p_off, l_off = T._run(False, ov)
        rel = float((p_on - p_off).norm() / p_off.norm())
        worst = max(abs(a - b) / max(1.0, abs(b)) for a, b in zip(l_on, l_off))
        if worst > 1e-4 or rel > 1e-5:
            bad += 1
            print(f"MISMATCH iter {i} {width}: on {l_on} off {l_off} rel {rel:.2e}", flush=True)
print(f"{bad} mismatching pairs of {2 * n}; allocated {torch.cuda.memory_allocated() / 2**20:.0f} MiB reserved {torch.cuda.memory_reserved() / 2**20:.0f} MiB")
