"""GPU: random-shape sweep of the conv engine (tools/conv_fuzz.py): forward with a random fused epilogue, input gradient and weight gradient in
the three arithmetics against float64 at shapes the model never uses (odd extents, channel counts that are no tile multiple, 1-pixel maps,
strides).  Unsupported configurations must be refused with an error, never answered with wrong values.  (600 cases: profiles/r04_conv_fuzz.txt.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_random_conv_shapes_match_float64_or_are_refused():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "conv_fuzz.py"), "--cases", "120", "--seed", "7"], capture_output=True, text=True,
                       timeout=550, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "FAILURES: 0" in r.stdout, tail
    # the only refusals are the documented vector-width requirements of the backward kernels
    for line in r.stdout.split("refused (RuntimeError) configurations:")[-1].splitlines():
        if " x " in line:
            assert "multiple of 4" in line or "multiples of 4" in line, line
