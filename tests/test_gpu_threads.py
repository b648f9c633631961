"""GPU: the C-ABI library called from several host threads at once, each thread on a HIP stream of its own.

The library keeps grow-only scratch per stream (Winograd workspace, split-M partial tiles, split-K tickets, top-k keys) in maps that are
created on FIRST use; ctypes drops the GIL around every call, the training step itself runs its forward on the Python thread and its
backward on autograd's, and a host may run a data path or a second model next to it.  First use from several threads at once must
neither corrupt those maps nor mix up scratch between streams: every thread's results must equal the single-threaded results bit for bit
(same kernels, same per-launch summation order).  Scope: the LIBRARY boundary.  The Python trainer keeps its stream choreography in
process-wide state (one trainer per process = the reference's one process per GPU, scripts/run_SI.sh:6); DESIGN.md section 7 says so."""
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu


def _work(seed, dev_stream, out, errors, barrier, rounds, math="bf16x6"):
    try:
        from abr_iod_amd import ops
        X6 = {"bf16x6": ops.MATH_BF16X6, "f16x3": ops.MATH_F16X3}[math]   # (f16x3: amax words from the shared ring, allocated by every thread at once)
        g = torch.Generator(device="cuda").manual_seed(seed)
        with torch.cuda.stream(dev_stream):
            x = torch.randn(2, 20, 24, 128, device="cuda", generator=g)
            w3 = torch.randn(128, 3, 3, 128, device="cuda", generator=g) * 0.03      # Winograd path (scratch per stream)
            w1 = torch.randn(256, 1, 1, 128, device="cuda", generator=g) * 0.05
            gy = torch.randn(2, 20, 24, 128, device="cuda", generator=g)
            logits = torch.randn(2, 30 * 40, 76, device="cuda", generator=g) * 3
            res = []
            barrier.wait()                     # every thread makes its FIRST library calls at the same moment
            for r in range(rounds):
                ver = 1000 * (seed + 1) + r + 1
                y3 = ops.conv_forward(x, w3, 1, 1, relu=True, math=X6, w_version=ver)
                y1 = ops.conv_forward(y3, w1, 1, 0, math=X6, w_version=ver)
                dw = torch.zeros_like(w3)
                ops.conv_wgrad(x, gy, dw, 1, 1, math=X6)                                  # split-M partials + reduction (scratch per stream)
                sc, idx = ops.topk_sigmoid(logits, 15, 2000)                              # per-stream key / histogram scratch
                res.append((y3, y1, dw, sc, idx))
            dev_stream.synchronize()
        out[seed] = res
    except Exception as e:   # surfaced by the main thread
        errors.append((seed, repr(e)))


@pytest.mark.timeout(300)
@pytest.mark.parametrize("math", ["bf16x6", "f16x3"])
def test_library_calls_from_concurrent_threads_match_serial(math):
    from abr_iod_amd import ops
    n_threads, rounds = 4, 6
    ops.conv_cache_clear()
    # serial reference: the same work, one thread after the other (fresh streams -> first use of every per-stream map entry as well)
    ref, errors = {}, []
    one = threading.Barrier(1)
    for t in range(n_threads):
        _work(t, torch.cuda.Stream(), ref, errors, one, rounds, math)
    assert not errors, errors
    ops.conv_cache_clear()
    got = {}
    barrier = threading.Barrier(n_threads)
    threads = [threading.Thread(target=_work, args=(t, torch.cuda.Stream(), got, errors, barrier, rounds, math)) for t in range(n_threads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=240)
        assert not th.is_alive(), "a worker thread hung"
    assert not errors, errors
    torch.cuda.synchronize()
    for t in range(n_threads):
        for r in range(rounds):
            for a, b, name in zip(got[t][r], ref[t][r], ("wino conv", "1x1 conv", "wgrad", "topk scores", "topk idx")):
                assert torch.equal(a, b), (t, r, name)
