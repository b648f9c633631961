"""GPU: the n-group-major tile order of the weights-direct kernels (conv_igemm.hip, launch_x6w_np; ABR_X6_NGROUP) only changes WHICH workgroup
computes which output tile: outputs are bit-identical for every group width, including widths that do not divide the number of n-tile columns
(narrower last group), batched (Winograd) launches and the rule's own choice.  The setting is read once per process: one child process per value."""
import hashlib
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import hashlib, sys, torch
sys.path.insert(0, %r)
from abr_iod_amd import ops
MATH = {"bf16x6": ops.MATH_BF16X6, "f16x3": ops.MATH_F16X3}[sys.argv[1]]
g = torch.Generator(device="cuda").manual_seed(5)
out = []
ver = 900
for (M, N, K) in [(1000, 1280, 1024), (4096, 2048, 512), (777, 2048 + 96, 512), (2304, 1024, 2048)]:      # 1x1: 10 / 16 / 17 / 8 n-tile columns
    x = torch.randn(1, 1, M, K, device="cuda", generator=g); w = torch.randn(N, 1, 1, K, device="cuda", generator=g) * 0.05
    res = torch.randn(1, 1, M, N, device="cuda", generator=g)
    ver += 1
    out.append(ops.conv_forward(x, w, 1, 0, residual=res, relu=True, math=MATH, w_version=ver))
for (B, H, W, C, N) in [(2, 19, 23, 1024, 1024 + 128), (40, 4, 4, 512, 512)]:                           # Winograd: 36 batched GEMMs
    x = torch.randn(B, H, W, C, device="cuda", generator=g); w = torch.randn(N, 3, 3, C, device="cuda", generator=g) * 0.02
    ver += 1
    out.append(ops.conv_forward(x, w, 1, 1, relu=True, math=MATH, w_version=ver))
assert ops.x6_range_flags(reset=False) == 0
h = hashlib.sha256()
for t in out:
    assert torch.isfinite(t).all()
    h.update(t.cpu().numpy().tobytes())
print("DIGEST", h.hexdigest())
""" % ROOT


@pytest.mark.timeout(600)
@pytest.mark.parametrize("math", ["bf16x6", "f16x3"])
def test_outputs_do_not_depend_on_the_tile_order(math):
    digests = {}
    for setting in ("0", None, "1", "3", "4", "7"):
        env = dict(os.environ)
        env.pop("ABR_X6_NGROUP", None)
        if setting is not None:
            env["ABR_X6_NGROUP"] = setting
        r = subprocess.run([sys.executable, "-c", CHILD, math], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
        assert r.returncode == 0, (setting, (r.stdout + r.stderr)[-2000:])
        digests[setting] = [l for l in r.stdout.splitlines() if l.startswith("DIGEST")][-1]
    assert len(set(digests.values())) == 1, digests
