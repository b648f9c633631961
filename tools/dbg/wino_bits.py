"""Hashes of 3x3 stride-1 (Winograd) conv outputs in every arithmetic on seeded data, with and without the library's cached derived weights.
Run it under two builds of the library (ABR_IOD_HIP_LIB=<other .so>) and diff the outputs: the Winograd-domain weights' arithmetic
(conv_winograd.hip::wino_u_row) is spelled out so that builds agree bit for bit (profiles/r06_wino_bits_vs_round5_build.txt)."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from abr_iod_amd import ops
for (co, ci) in [(128, 128), (128, 64), (256, 256), (512, 512), (1024, 1024), (132, 64)]:
    g = torch.Generator(device="cuda").manual_seed(co * 7 + ci)
    x = torch.randn(2, 19, 23, ci, device="cuda", generator=g)
    w = torch.randn(co, 3, 3, ci, device="cuda", generator=g) * 0.05
    for name, m in (("f32", ops.MATH_F32), ("bf16x6", ops.MATH_BF16X6), ("f16x3", ops.MATH_F16X3)):
        for ver in (0, 11):
            if co % 32 and ver:
                continue
            y = ops.conv_forward(x, w, 1, 1, math=m, w_version=ver)
            torch.cuda.synchronize()
            print(co, ci, name, "ver", ver, hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:16])
