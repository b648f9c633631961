"""GPU: unusual-but-legal batches through the default training step at full width (tools/edge_steps.py): odd image extents, 1 / 7 images, 60
ground-truth boxes per image, a box covering the whole image, 8-pixel boxes, boxes on the borders, 224x320 and 1000x1666 images, portrait
images, all back to back with the next batch prefetched -- finite losses and gradients, equal to the step with every stream folded into one."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_unusual_batches_step_like_the_folded_step():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "edge_steps.py")], capture_output=True, text=True, timeout=550, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "FAILURES: 0" in r.stdout and r.stdout.count(" ok") >= 9, tail
