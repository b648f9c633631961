#!/usr/bin/env python3
"""Exploration: error of the fp32-MFMA and the bf16x6 contractions against float64 on adversarial operands (wide exponent spread
inside one reduction, tiny / huge magnitudes, cancellation, inf / nan).  Prints one line per case and arithmetic."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from abr_iod_amd import ops


def run(name, x, w):
    """x [M,K], w [N,K] fp32 cuda -> errors of y = x w^T, dW = y^T x"""
    M, K = x.shape; N = w.shape[0]
    y64 = x.double() @ w.double().t()
    scale = (x.double().abs() @ w.double().abs().t())            # sum |x||w| : the natural error scale of a dot product
    out = []
    for m, tag in ((ops.MATH_F32, "f32"), (ops.MATH_BF16X6, "x6 ")):
        y = ops.conv_forward(x.view(1, M, 1, K), w.view(N, 1, 1, K), 1, 0, math=m).view(M, N)
        e = (y.double() - y64)
        fin = torch.isfinite(y64) & torch.isfinite(scale) & (scale > 0)
        rel = (e[fin].abs() / scale[fin]).max().item() if fin.any() else float("nan")
        same_nonfinite = bool(((torch.isnan(y) == torch.isnan(y64.float())) & (torch.isinf(y) == torch.isinf(y64.float()))).all())
        out.append((tag, rel, same_nonfinite))
    print(f"{name:34s} " + "   ".join(f"{t}: max|err|/sum|x||w| = {r:.3e} nonfinite-pattern-ok={s}" for t, r, s in out), flush=True)


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    M, N, K = 512, 256, 1024
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
    ru = lambda *s: torch.rand(*s, device="cuda", generator=g)
    run("N(0,1)", rn(M, K), rn(N, K))
    for sp in (10, 20, 40, 60):
        run(f"spread 2^+-{sp} inside reduction", rn(M, K) * torch.exp2((ru(M, K) * 2 - 1) * sp), rn(N, K) * torch.exp2((ru(N, K) * 2 - 1) * sp))
    for e in (-60, -90, -100, -105, -110, -120, -126):
        run(f"x ~ 2^{e}", rn(M, K) * 2.0 ** e, rn(N, K))
    for e in (60, 100, 120):
        run(f"x ~ 2^{e}", rn(M, K) * 2.0 ** e, rn(N, K) * 2.0 ** -10)
    run("x,w ~ 2^-60 both", rn(M, K) * 2.0 ** -60, rn(N, K) * 2.0 ** -60)
    # cancellation: pairs (u, -u(1+d)) against equal x
    v = rn(M, K // 2); x = torch.stack([v, v], 2).reshape(M, K)
    u = rn(N, K // 2); w = torch.stack([u, -u * (1 + 2.0 ** -12 * rn(N, K // 2))], 2).reshape(N, K)
    run("cancellation (pairs u,-u(1+2^-12))", x, w)
    x = rn(M, K); x[3, 5] = float("inf"); x[7, 9] = float("nan"); x[11, 2] = -float("inf")
    run("inf / nan elements", x, rn(N, K))
    x = rn(M, K); x[3, 5] = 3.3e38
    run("near FLT_MAX element", x, rn(N, K) * 2.0 ** -20)
    x = rn(M, K) * 2.0 ** -130
    run("fp32-subnormal x", x, rn(N, K))


if __name__ == "__main__":
    main()
