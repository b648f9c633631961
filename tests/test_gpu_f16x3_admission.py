"""GPU: admission tests of the f16x3 arithmetic (ABR_MATH_F16X3, round 5): every contraction operand written x = s (h0 + h1) with
h0 = fp16(x / s), h1 = fp16(x / s - h0) and s = the power of two that puts the operand's largest magnitude in [2^14, 2^15) -- per tensor
for activations / gradients (its amax word), per output channel for weights -- three products h0 g0 + h0 g1 + h1 g0 on
v_mfma_f32_32x32x16_f16, fp32 accumulation (csrc/common.h, conv_igemm.hip, conv_wgrad.hip).  It mirrors tests/test_gpu_x6_admission.py with
the SAME bounds: this is what the arithmetic must show before it may produce the benchmark number.

Error model.  An element keeps a relative accuracy of 2^-22 while it is within 18 binades of its tensor's (weight row's) amax and an ABSOLUTE
accuracy of 2^-40 amax below that, random-signed.  A dot product therefore carries, beside the fp32 accumulation error every arithmetic of
this library has,   |err| <= 2^-22 sum|x||w|  +  2^-40 (amax_x sum_k |w_k| + amax_w sum_k |x_k|).
  * INSIDE the domain -- the second term is below the first, i.e. the operands' magnitudes, weighted by their partners, sit within ~16
    binades of their amax: N(0,1) data, exponents spread over 2^+-20 (40 binades) inside every reduction, tensors living anywhere between
    2^-105 and 2^120, sums that cancel to 2^-12 of their terms, ReLU-like zeros -- the error against float64, measured against the dot
    product's natural scale sum|x||w|, must stay within max(2 x the fp32 MFMA kernel's, 8 ulp) and below 32 ulp, exactly as demanded of
    bf16x6;
  * OUTSIDE it -- exponents spread over 2^+-40 / 2^+-60 inside every reduction (80 / 120 binades: what bf16x6 still takes), a reduction
    made only of elements far below the tensor's amax -- the error must stay within the ABSOLUTE bound above (tested with a factor 2), and
    the kernels count the elements below the 18 binades (ops.h3_range_stats: the trainer reads the fraction every few steps and
    leaves the arithmetic for bf16x6 above ABR_H3_MAX_SMALL_FRACTION); inf / nan operands raise ABR_X6_FLAG_NONFINITE, poison their own
    outputs, and the trainer switches the models to the fp32 MFMA kernels; an amax word that does not carry the epoch the caller names
    raises ABR_H3_FLAG_STALE (a plumbing bug, never data).
Covered contractions: 1x1 conv forward (= plain GEMM), 3x3 through the Winograd domain, the input gradient, the weight gradient; operand amax
taken from the producer's tag AND reduced by the library."""
import pytest
import torch

pytestmark = pytest.mark.gpu

EPS = 2.0 ** -24
M, N, K = 384, 256, 1024


def _gen(seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (lambda *s: torch.randn(*s, device="cuda", generator=g)), (lambda *s: torch.rand(*s, device="cuda", generator=g))


def _gemm(x, w, math):
    from abr_iod_amd import ops
    return ops.conv_forward(x.view(1, x.shape[0], 1, x.shape[1]).contiguous(), w.view(w.shape[0], 1, 1, w.shape[1]).contiguous(), 1, 0, math=math).view(x.shape[0], w.shape[0])


def _wgrad(x, gy, math):
    """dW[n,k] = sum_m gy[m,n] x[m,k]"""
    from abr_iod_amd import ops
    dw = torch.zeros(gy.shape[1], 1, 1, x.shape[1], device="cuda")
    ops.conv_wgrad(x.view(1, x.shape[0], 1, x.shape[1]).contiguous(), gy.view(1, gy.shape[0], 1, gy.shape[1]).contiguous(), dw, 1, 0, math=math)
    return dw.view(gy.shape[1], x.shape[1])


def _rel_err(y, y64, scale):
    ok = scale > 0
    return float(((y.double() - y64).abs()[ok] / scale[ok]).max())


ALL_CASES = ["N(0,1)", "exponent spread 2^+-20", "exponent spread 2^+-40", "exponent spread 2^+-60", "x ~ 2^-60", "x ~ 2^-100", "x ~ 2^-105",
             "x ~ 2^100", "x ~ 2^120", "x, w ~ 2^-60", "cancellation to 2^-12", "70 % exact zeros"]
OUT_OF_DOMAIN = ["exponent spread 2^+-40", "exponent spread 2^+-60"]
CASES = [c for c in ALL_CASES if c not in OUT_OF_DOMAIN]


def _case(name):
    rn, ru = _gen(ALL_CASES.index(name))
    if name == "N(0,1)":
        return rn(M, K), rn(N, K)
    if name.startswith("exponent spread"):   # exponents spread over 2^+-sp INSIDE every reduction, both operands
        sp = int(name.split("+-")[1])
        return rn(M, K) * torch.exp2((ru(M, K) * 2 - 1) * sp), rn(N, K) * torch.exp2((ru(N, K) * 2 - 1) * sp)
    if name.startswith("x ~ 2^"):             # magnitudes in [0.5, 1.5) * 2^e, random signs
        e = int(name.split("^")[1])
        return torch.sign(rn(M, K)) * (0.5 + ru(M, K)) * 2.0 ** e, rn(N, K) * (2.0 ** -10 if e > 0 else 1.0)
    if name == "x, w ~ 2^-60":
        return torch.sign(rn(M, K)) * (0.5 + ru(M, K)) * 2.0 ** -60, torch.sign(rn(N, K)) * (0.5 + ru(N, K)) * 2.0 ** -60
    if name == "cancellation to 2^-12":       # pairs (u, -u(1 + 2^-12 r)) against equal x
        v, u = rn(M, K // 2), rn(N, K // 2)
        return torch.stack([v, v], 2).reshape(M, K), torch.stack([u, -u * (1 + 2.0 ** -12 * rn(N, K // 2))], 2).reshape(N, K)
    if name == "70 % exact zeros":            # ReLU-like
        xz = rn(M, K)
        xz[xz < 0.5] = 0.0
        return xz, rn(N, K)
    raise KeyError(name)


@pytest.mark.parametrize("name", CASES)
def test_in_domain_error_matches_fp32_kernel(name):
    from abr_iod_amd import ops
    x, w = _case(name)
    ops.x6_range_flags(reset=True)
    y64 = x.double() @ w.double().t()
    scale = x.double().abs() @ w.double().abs().t()
    e32 = _rel_err(_gemm(x, w, ops.MATH_F32), y64, scale)
    e3 = _rel_err(_gemm(x, w, ops.MATH_F16X3), y64, scale)
    print(f"{name}: forward  f32 {e32 / EPS:.1f} ulp   f16x3 {e3 / EPS:.1f} ulp")
    assert e3 <= max(2.0 * e32, 8 * EPS), (name, e3, e32)
    assert e3 <= 32 * EPS, (name, e3)
    # the weight gradient reduces over the ROW axis: dW[n,k] = sum_m G[m,n] X[m,k]; X = columns of x, G = w's values re-shaped to [M, N]
    G, X = w.t()[:M].contiguous(), x[:, :N].contiguous()
    d64 = G.double().t() @ X.double()
    dscale = G.double().abs().t() @ X.double().abs()
    w32 = _rel_err(_wgrad(X, G, ops.MATH_F32), d64, dscale)
    w3 = _rel_err(_wgrad(X, G, ops.MATH_F16X3), d64, dscale)
    print(f"{name}: wgrad    f32 {w32 / EPS:.1f} ulp   f16x3 {w3 / EPS:.1f} ulp")
    assert w3 <= max(2.0 * w32, 8 * EPS), (name, w3, w32)
    assert ops.x6_range_flags(reset=True) == 0, name


@pytest.mark.parametrize("name", OUT_OF_DOMAIN)
def test_out_of_domain_spread_stays_within_the_absolute_bound(name):
    """80 / 120 binades inside every reduction: beyond what two fp16 terms under one scale can hold relative to each element; what is promised
    (and counted by the guard) is the absolute floor of 2^-40 amax per element"""
    from abr_iod_amd import ops
    x, w = _case(name)
    ops.x6_range_flags(reset=True)
    ops.h3_range_stats(reset=True)
    x64, w64 = x.double(), w.double()
    y64, scale = x64 @ w64.t(), x64.abs() @ w64.abs().t()
    floor = 2.0 ** -39 * (float(x.abs().max()) * w64.abs().sum(1)[None, :] + w64.abs().amax(1)[None, :] * x64.abs().sum(1)[:, None])
    err = (_gemm(x, w, ops.MATH_F16X3).double() - y64).abs()
    e32 = _rel_err(_gemm(x, w, ops.MATH_F32), y64, scale)
    print(f"{name}: forward  f32 {e32 / EPS:.1f} ulp   f16x3 {float((err / scale).max()) / EPS:.1f} ulp of sum|x||w|, {float((err / (floor + 1e-300)).max()):.3f} of the absolute floor")
    assert bool((err <= max(2.0 * e32, 8 * EPS) * scale + floor).all()), name
    G, X = w.t()[:M].contiguous(), x[:, :N].contiguous()
    d64, dscale = G.double().t() @ X.double(), G.double().abs().t() @ X.double().abs()
    dfloor = 2.0 ** -39 * (float(G.abs().max()) * X.double().abs().sum(0)[None, :] + float(X.abs().max()) * G.double().abs().sum(0)[:, None])
    derr = (_wgrad(X, G, ops.MATH_F16X3).double() - d64).abs()
    w32 = _rel_err(_wgrad(X, G, ops.MATH_F32), d64, dscale)
    print(f"{name}: wgrad    f32 {w32 / EPS:.1f} ulp   f16x3 {float((derr / dscale).max()) / EPS:.1f} ulp, {float((derr / (dfloor + 1e-300)).max()):.3f} of the absolute floor")
    assert bool((derr <= max(2.0 * w32, 8 * EPS) * dscale + dfloor).all()), name
    assert ops.x6_range_flags(reset=True) == 0
    small, seen = ops.h3_range_stats(reset=True)
    print(f"{name}: {small} of {seen} inspected operand elements more than 18 binades below their amax ({small / seen:.2f})")
    assert small / seen > 0.5


def test_in_domain_winograd_and_dgrad_paths():
    """3x3 stride-1 conv (Winograd-domain GEMMs: the split operands are B^T d B and G g G^T, each with its own amax / row scales) and its
    input gradient, wide spread; with and without a weight version (cached planes / planes packed into scratch per call)."""
    from abr_iod_amd import ops
    rn, ru = _gen(1)
    B, H, W, C, Co = 2, 20, 24, 128, 128
    for sp in (0, 30):
        x = rn(B, H, W, C) * torch.exp2((ru(B, H, W, C) * 2 - 1) * sp)
        w = rn(Co, 3, 3, C) / (9 * C) ** 0.5
        y64 = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), padding=1)
        s64 = torch.nn.functional.conv2d(x.double().abs().permute(0, 3, 1, 2), w.double().abs().permute(0, 3, 1, 2), padding=1)
        ops.x6_range_flags(reset=True)
        e = {}
        for m, tag, ver in ((ops.MATH_F32, "f32", 0), (ops.MATH_F16X3, "h3", 0), (ops.MATH_F16X3, "h3 cached", 11)):
            y = ops.conv_forward(x, w, 1, 1, math=m, w_version=ver).permute(0, 3, 1, 2)
            e[tag] = float(((y.double() - y64).abs() / s64).max())
        print(f"winograd 3x3, spread 2^+-{sp}: f32 {e['f32'] / EPS:.1f} ulp, f16x3 {e['h3'] / EPS:.1f} ulp (of sum|x||w|; includes the transforms' own fp32 rounding)")
        assert e["h3"] <= max(2.0 * e["f32"], 8 * EPS)
        assert e["h3 cached"] == e["h3"]
        assert ops.x6_range_flags(reset=True) == 0
        # input gradient of a 1x1 conv = forward with the transposed weight copy
        w1 = rn(Co, 1, 1, C)
        wt = ops.conv_dgrad_weights(w1, None)
        gy = rn(B, H, W, Co) * torch.exp2((ru(B, H, W, Co) * 2 - 1) * sp)
        g64 = gy.double().reshape(-1, Co) @ w1.double().view(Co, C)
        gs = gy.double().abs().reshape(-1, Co) @ w1.double().abs().view(Co, C)
        d = {}
        for m, tag in ((ops.MATH_F32, "f32"), (ops.MATH_F16X3, "h3")):
            gx = ops.conv_forward(gy, wt, 1, 0, math=m).reshape(-1, C)
            d[tag] = float(((gx.double() - g64).abs() / gs).max())
        assert d["h3"] <= max(2.0 * d["f32"], 8 * EPS), d
        # the Winograd weight gradient, with the forward pass's kept V and without
        gy3 = rn(B, H, W, Co) * 1e-3
        d64 = torch.zeros(Co, 3, 3, C, dtype=torch.float64, device="cuda")
        s64w = torch.zeros_like(d64)
        xp = torch.nn.functional.pad(x.double(), (0, 0, 1, 1, 1, 1))
        for r in range(3):
            for s_ in range(3):
                xs = xp[:, r:r + H, s_:s_ + W, :].reshape(-1, C)
                d64[:, r, s_, :] = gy3.double().reshape(-1, Co).t() @ xs
                s64w[:, r, s_, :] = gy3.double().abs().reshape(-1, Co).t() @ xs.abs()
        ew = {}
        for m, tag, keep in ((ops.MATH_F32, "f32", False), (ops.MATH_F16X3, "h3", False), (ops.MATH_F16X3, "h3 kept V", True)):
            v = ops.wino_v_alloc(x, w, 1, 1, m) if keep else None
            if keep:
                assert v is not None
                ops.conv_forward(x, w, 1, 1, math=m, wino_v=v, w_version=11)
            dw = torch.zeros(Co, 3, 3, C, device="cuda")
            ops.conv_wgrad(x, gy3, dw, 1, 1, math=m, wino_v=v)
            ew[tag] = float(((dw.double() - d64).abs() / s64w).max())
        print(f"winograd wgrad, spread 2^+-{sp}: f32 {ew['f32'] / EPS:.1f} ulp, f16x3 {ew['h3'] / EPS:.1f} ulp, with the kept V {ew['h3 kept V'] / EPS:.1f}")
        assert ew["h3"] <= max(2.0 * ew["f32"], 8 * EPS) and ew["h3 kept V"] <= max(2.0 * ew["f32"], 8 * EPS), ew
        assert ops.x6_range_flags(reset=True) == 0


def test_amax_from_the_producers_tag_equals_amax_reduced_by_the_library(monkeypatch):
    """A conv output carries its amax word (written by the producing kernel's epilogue); a consumer that uses the tag must give the same
    bits as one that lets the library reduce the tensor (same amax -> same scale -> same products)."""
    from abr_iod_amd import ops
    rn, _ = _gen(5)
    x = rn(2, 24, 20, 128)
    w1, w2 = rn(256, 1, 1, 128) / 11, rn(128, 3, 3, 256) / 48
    sc, bi = torch.rand(256, device="cuda") + 0.5, rn(256) * 0.1
    for math in (ops.MATH_F16X3, ops.MATH_F32):   # (an fp32-kernel producer tags its output too when asked to)
        h = ops.conv_forward(x, w1, 1, 0, scale=sc, bias=bi, relu=True, math=math, emit_amax=True)
        assert ops.amax_of(h)[0] is not None
        y_tag = ops.conv_forward(h, w2, 1, 1, math=ops.MATH_F16X3)
        h2 = h.clone()                       # no tag: amax_compute reduces it
        assert ops.amax_of(h2)[0] is None
        y_red = ops.conv_forward(h2, w2, 1, 1, math=ops.MATH_F16X3)
        assert torch.equal(y_tag, y_red)
        monkeypatch.setattr(ops, "H3_TAGS", False)    # no tags at all: the library reduces inside the call
        y_lib = ops.conv_forward(h2.clone(), w2, 1, 1, math=ops.MATH_F16X3)
        monkeypatch.setattr(ops, "H3_TAGS", True)
        assert torch.equal(y_tag, y_lib)
    # the strided scatter of a stride-2 1x1 conv's dgrad (rows on the even pixels of a zeroed tensor) carries its amax too, in the fresh pass and in
    # the second pass that adds the downsample branch's gradient into the same tensor: a consumer gives the same bits either way
    g1, g2 = rn(2, 12, 10, 256), rn(2, 12, 10, 128)
    wt1, wt2 = rn(64, 1, 1, 256) / 16, rn(64, 1, 1, 128) / 11
    ops.amax_compute(g1); ops.amax_compute(g2)
    gx = ops.conv_forward(g1, wt1, 1, 0, out_hw=(24, 20), out_stride=(2, 2), math=ops.MATH_F16X3)
    w_first = ops.amax_of(gx)[0]
    assert w_first is not None
    ops.conv_forward(g2, wt2, 1, 0, residual=gx, out=gx, out_hw=(24, 20), out_stride=(2, 2), math=ops.MATH_F16X3)
    assert ops.amax_of(gx)[0] not in (None, w_first)
    w3 = rn(64, 1, 1, 64) / 8
    y_tag = ops.conv_forward(gx, w3, 1, 0, math=ops.MATH_F16X3)
    y_red = ops.conv_forward(gx.clone(), w3, 1, 0, math=ops.MATH_F16X3)
    assert torch.equal(y_tag, y_red)
    other = torch.zeros_like(gx)        # a scatter into a tensor this geometry did not zero-fill gets no tag
    ops.conv_forward(g2, wt2, 1, 0, residual=other, out=other, out_hw=(24, 20), out_stride=(2, 2), math=ops.MATH_F16X3)
    assert ops.amax_of(other)[0] is None
    # an in-place write through torch invalidates the tag (the tensor's version moves)
    h = ops.conv_forward(x, w1, 1, 0, math=ops.MATH_F16X3)
    assert ops.amax_of(h)[0] is not None
    h.mul_(3.0)
    assert ops.amax_of(h)[0] is None
    assert ops.x6_range_flags(reset=True) == 0


def test_reduction_made_of_small_elements_only_is_bounded_absolutely_and_counted():
    """Rows of x 30 binades below the tensor's amax: every product of such a row carries an absolute error <= 2^-40 amax |w| (h1 is subnormal
    there) -- far outside the relative bound, exactly inside the absolute one -- and the guard counts the elements."""
    from abr_iod_amd import ops
    rn, ru = _gen(6)
    x, w = rn(M, K), rn(N, K)
    x[: M // 2] *= 2.0 ** -30
    ops.x6_range_flags(reset=True)
    ops.h3_range_stats(reset=True)
    y = _gemm(x, w, ops.MATH_F16X3).double()
    y64 = x.double() @ w.double().t()
    amax = float(x.abs().max())
    bound = (2.0 ** -40) * amax * w.double().abs().sum(1)[None, :] + 8 * EPS * (x.double().abs() @ w.double().abs().t())
    assert bool(((y - y64).abs() <= bound).all())
    rel_small = _rel_err(y[: M // 2].float(), y64[: M // 2], (x.double().abs() @ w.double().abs().t())[: M // 2])
    print(f"all-small rows: {rel_small / EPS:.0f} ulp relative to their own scale (bounded absolutely instead)")
    assert ops.x6_range_flags(reset=True) == 0
    small, seen = ops.h3_range_stats(reset=True)
    assert seen == M * K and M * K // 2 <= small <= M * K // 2 + M * K // 100, (small, seen)


def test_non_finite_operands_raise_the_flag_and_poison_their_outputs():
    from abr_iod_amd import ops
    rn, _ = _gen(3)
    x, w = rn(M, K), rn(N, K)
    x[3, 5] = float("inf"); x[7, 900] = float("nan"); x[11, 2] = -float("inf")
    ops.x6_range_flags(reset=True)
    y32 = _gemm(x, w, ops.MATH_F32)
    y3 = _gemm(x, w, ops.MATH_F16X3)
    assert ops.x6_range_flags(reset=True) & ops.X6_FLAG_NONFINITE
    bad = ~torch.isfinite(y32)
    assert bad.any(dim=1).nonzero().flatten().tolist() == [3, 7, 11]
    assert bool((~torch.isfinite(y3))[[3, 7, 11]].all())       # the rows that hold the non-finite elements are non-finite
    w2 = w.clone(); w2[5, 7] = float("nan")
    y3 = _gemm(rn(M, K), w2, ops.MATH_F16X3)
    assert ops.x6_range_flags(reset=True) & ops.X6_FLAG_NONFINITE
    assert bool((~torch.isfinite(y3))[:, 5].all())
    _wgrad(x[:, :N].contiguous(), rn(M, N), ops.MATH_F16X3)
    assert ops.x6_range_flags(reset=True) & ops.X6_FLAG_NONFINITE


def test_the_fp32_stem_hands_its_amax_on():
    """the stem conv runs on the fp32 MFMA kernel; asked to, its epilogue writes the output's amax word all the same, and the max-pooled copy
    inherits the tag (layer1's f16x3 convs then need no reduction pass over the pooled tensor)"""
    import ctypes
    from abr_iod_amd import ops
    rn, _ = _gen(9)
    x = rn(2, 64, 80, 4)
    ws = rn(64, 7, 7, 4) / 14
    y = ops.conv_forward(x, ws, 2, 3, relu=True, emit_amax=True)
    w, e = ops.amax_of(y)
    assert w is not None
    host = ctypes.c_uint64(0)
    torch.cuda.synchronize()
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    assert hip.hipMemcpy(ctypes.byref(host), ctypes.c_void_p(w), 8, 2) == 0   # hipMemcpyDeviceToHost
    assert (int(host.value) >> 32) == e
    assert torch.tensor([int(host.value) & 0xFFFFFFFF], dtype=torch.int64).to(torch.int32).view(torch.float32).item() == float(y.abs().max())
    p = ops.maxpool3x3s2(y)
    assert ops.amax_of(p) == ops.amax_of(y)
    w1 = rn(64, 1, 1, 64) / 8
    assert torch.equal(ops.conv_forward(p, w1, 1, 0, math=ops.MATH_F16X3), ops.conv_forward(p.clone(), w1, 1, 0, math=ops.MATH_F16X3))


def test_amax_words_come_in_blocks_and_never_collide():
    """ops.amax_new hands out the words of abr_h3_amax_alloc_block's blocks (one library call per 512 tensors); single allocations made by the
    library in between (abr_h3_amax_alloc: the Winograd path's V / M words) must neither reuse a block's words nor its epochs, also across a block
    boundary and from several host threads at once"""
    import ctypes as C
    import threading
    from abr_iod_amd import ops, _lib as L
    seen, lock = set(), threading.Lock()

    def take(n, single_every):
        got = []
        for i in range(n):
            got.append(ops.amax_new())
            if i % single_every == 0:
                w, e = C.c_void_p(0), C.c_uint32(0)
                L.check(L.lib().abr_h3_amax_alloc(C.byref(w), C.byref(e)), "h3_amax_alloc")
                got.append((int(w.value), int(e.value)))
        with lock:
            for g in got:
                assert g not in seen, g
                seen.add(g)
    threads = [threading.Thread(target=take, args=(700, 7 + t)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    words = {}
    for w, e in seen:
        assert e != 0 and w % 8 == 0
        words.setdefault(w, set()).add(e)
    assert len(seen) >= 4 * 700
    # within one pass of the ring (65536 words, far more than taken here) an address is handed out once
    assert all(len(es) == 1 for es in words.values())
    base = min(words)
    assert max(words) - base < 65536 * 8


def test_stream_wait_stream_orders_the_waiter_behind_the_signaller():
    """abr_stream_wait_stream(waiter, signaller) = torch's waiter.wait_stream(signaller): a consumer queued on the waiter afterwards sees what the
    signaller had queued before (a long chain of dependent kernels), repeatedly and in both directions"""
    from abr_iod_amd import _lib as L
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    x = torch.zeros(1 << 22, device="cuda")
    torch.cuda.synchronize()
    for it in range(20):
        s1, s2 = (a, b) if it % 2 == 0 else (b, a)
        with torch.cuda.stream(s1):
            for _ in range(30):
                x.add_(1.0)              # 30 dependent passes over 16 MB: the signaller is busy for a while
        L.check(L.lib().abr_stream_wait_stream(s2.cuda_stream, s1.cuda_stream), "stream_wait_stream")
        with torch.cuda.stream(s2):
            y = x.clone()                # must see all 30 additions of this round
            x.add_(0.0)                  # (keeps the next round's writes behind this read through the same ordering call)
        L.check(L.lib().abr_stream_wait_stream(s1.cuda_stream, s2.cuda_stream), "stream_wait_stream")
        s2.synchronize()
        assert float(y.min()) == float(y.max()) == 30.0 * (it + 1), it
    torch.cuda.synchronize()


def test_stale_amax_word_raises_the_flag():
    """a consumer told an epoch its amax word does not carry (the plumbing bug the epochs exist to catch)"""
    import ctypes as C
    from abr_iod_amd import _lib as L, ops
    rn, _ = _gen(8)
    x, w = rn(1, 64, 1, 64), rn(64, 1, 1, 64)
    word, epoch = ops.amax_new()
    L.check(L.lib().abr_h3_amax(L.ptr(x), x.numel(), word, epoch, L.stream()), "h3_amax")
    d = ops.conv_desc(x.shape, w.shape, 1, 0, math=ops.MATH_F16X3)
    out = torch.empty(1, 64, 1, 64, device="cuda")
    ops.x6_range_flags(reset=True)
    d.x_amax, d.x_amax_epoch = word, epoch
    L.check(L.lib().abr_conv_forward(C.byref(d), L.ptr(x), L.ptr(w), L.ptr(out), L.stream()), "conv_forward")
    assert ops.x6_range_flags(reset=True) & ops.H3_FLAG_STALE == 0
    d.x_amax_epoch = epoch + 1
    L.check(L.lib().abr_conv_forward(C.byref(d), L.ptr(x), L.ptr(w), L.ptr(out), L.stream()), "conv_forward")
    assert ops.x6_range_flags(reset=True) & ops.H3_FLAG_STALE


@pytest.mark.parametrize("trip", ["nonfinite", "small", "clean"])
def test_trainer_guard_on_f16x3(trip, monkeypatch):
    """inf / nan operands: both models leave the f16x3 arithmetic for the fp32 MFMA kernels.  A large share of operand elements far below
    their tensor's amax: both models move to bf16x6.  Ordinary data (a fraction of 1e-3): nothing happens."""
    import logging
    import os
    from abr_iod_amd import ops
    from abr_iod_amd.engine import trainer
    from abr_iod_amd.engine.synthetic import build_models, make_cfgs
    monkeypatch.setattr(trainer, "H3_STATS_EVERY", 1)
    tiny = ["MODEL.RESNETS.STEM_OUT_CHANNELS", 16, "MODEL.RESNETS.RES2_OUT_CHANNELS", 32, "MODEL.RESNETS.WIDTH_PER_GROUP", 8,
            "MODEL.RESNETS.BACKBONE_OUT_CHANNELS", 128]
    os.environ["ABR_CONV_MATH"] = "f16x3"
    try:
        cfg_s, cfg_t = make_cfgs("15-5", overrides=tiny)
        ms, mt = build_models(cfg_s, cfg_t, seed=0)
    finally:
        os.environ.pop("ABR_CONV_MATH", None)
    assert mt.conv_math == ms.conv_math == "f16x3"
    assert all(m.math == ops.MATH_F16X3 for m in mt.modules() if hasattr(m, "math"))
    ops.x6_range_flags(reset=True)
    ops.h3_range_stats(reset=True)
    trainer.trainer_state(mt).x6_watch = None
    for _ in range(3):                       # clean steps: nothing happens
        trainer._x6_guard(ms, mt)
        torch.cuda.synchronize()
    assert mt.conv_math == "f16x3"
    rn, _ = _gen(4)
    bad = rn(M, K)
    if trip == "nonfinite":
        bad[5, 7] = float("inf")
    elif trip == "small":
        bad[: M // 2] *= 2.0 ** -25
    _gemm(bad, rn(N, K), ops.MATH_F16X3)
    records = []
    h = logging.Handler(); h.emit = records.append
    log = logging.getLogger("h3test." + trip); log.addHandler(h); log.setLevel(logging.INFO)
    for _ in range(4):                       # the polls are asynchronous: a flag / a count is seen one or two steps later
        trainer._x6_guard(ms, mt, log)
        torch.cuda.synchronize()
    if trip == "clean":
        assert mt.conv_math == ms.conv_math == "f16x3" and not records
        assert trainer.trainer_state(mt).h3_small_fraction < 0.01
        return
    if trip == "small":
        assert mt.conv_math == ms.conv_math == "bf16x6"
        assert all(m.math == ops.MATH_BF16X6 for m in mt.modules() if hasattr(m, "math"))
        assert len(records) == 1 and "18 binades" in records[0].getMessage() and records[0].levelno == logging.WARNING
        return
    assert mt.conv_math == ms.conv_math == "f32"
    assert all(m.math == ops.MATH_F32 for m in mt.modules() if hasattr(m, "math"))
    assert len(records) == 1 and "range guard tripped" in records[0].getMessage()
    ops.x6_range_flags(reset=True)
