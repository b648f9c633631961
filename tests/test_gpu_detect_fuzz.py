"""GPU: random / degenerate-input sweep of the detection kernels against the oracle (tools/detect_fuzz.py): ROIAlign forward bit-exact and
backward in both forms on RoIs that are empty, inverted, outside the map or larger than it (pooled sizes 1..9, sampling ratios 0..3), NMS keep
lists index-exact on box lists full of duplicates and exact IoU ties (thresholds 0..1, both comparison conventions, ragged counts incl. empty
images), and the sigmoid + top-k ranking under heavy score ties.  (1000 cases: profiles/r04_detect_fuzz.txt.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_random_and_degenerate_inputs_match_the_oracle():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "detect_fuzz.py"), "--cases", "80", "--seed", "11"], capture_output=True, text=True,
                       timeout=550, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "FAILURES: 0" in r.stdout, tail
