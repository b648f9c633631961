"""GPU: the opt-in shared frozen prefix (engine/trainer.py::SHARE_FROZEN_PREFIX) -- the source and the target model hold identical frozen stem +
layer1 weights in the reference's setup (both load the same checkpoint, FREEZE_CONV_BODY_AT = 2: tools/train_incremental.py:180-215), so that
prefix is computed once per batch and fed to both.  It must (a) change nothing in the step's results and (b) switch itself off as
soon as any frozen tensor of the two models differs."""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(share, steps=3, perturb=False):
    from abr_iod_amd import ops
    from abr_iod_amd.engine import train_step, trainer
    from abr_iod_amd.engine.synthetic import build_models, make_cfgs, synthetic_batch
    from abr_iod_amd.solver.build import make_lr_scheduler, make_optimizer
    cfg_s, cfg_t = make_cfgs("15-5", dist_type="id", feat="ard", alpha=0.5, beta=1.0, gamma=1.0)
    ms, mt = build_models(cfg_s, cfg_t, seed=0)
    if perturb:
        with torch.no_grad():
            ms.backbone.body.layer1[1].conv2.weight[3, 0, 1, 1] += 1e-3
        from abr_iod_amd.modeling.backbone import resnet
        resnet._STATIC_VERSION[0] += 1
    opt = make_optimizer(cfg_t, mt); sch = make_lr_scheduler(cfg_t, opt)
    batches = [synthetic_batch(2, 320, 480, seed=10 + i) for i in range(steps + 1)]
    old = trainer.SHARE_FROZEN_PREFIX[0]
    trainer.SHARE_FROZEN_PREFIX[0] = share
    try:
        ops._sample_calls[0] = 0
        random.seed(0); torch.manual_seed(0)
        out = []
        for i in range(steps):
            im, tg = batches[i]
            ld, total = train_step(ms, mt, im, tg, opt, sch, cfg_t, next_images=batches[i + 1][0])
            out.append({k: float(v.detach()) for k, v in ld.items()})
        torch.cuda.synchronize()
        shared = trainer.trainer_state(mt).share_ok if share else False
        return out, mt.flat.params.clone(), shared
    finally:
        trainer.SHARE_FROZEN_PREFIX[0] = old


def _close(a, b, tol=2e-6):
    return all(abs(x[k] - y[k]) <= tol * max(1.0, abs(x[k])) for x, y in zip(a, b) for k in x)


def test_frozen_prefix_of_both_models_is_the_same_tensor_bit_for_bit():
    """what sharing relies on: identical frozen weights + deterministic kernels => identical stem / layer1 outputs"""
    from abr_iod_amd.engine import trainer
    from abr_iod_amd.engine.synthetic import build_models, make_cfgs, synthetic_batch
    cfg_s, cfg_t = make_cfgs("15-5", dist_type="id", feat="ard", alpha=0.5, beta=1.0, gamma=1.0)
    ms, mt = build_models(cfg_s, cfg_t, seed=0)
    assert trainer.frozen_prefix_shareable(ms, mt)
    im, _ = synthetic_batch(2, 320, 480, seed=3)
    with torch.no_grad():
        xs, ds = ms.prefetch_frozen(im)
        xt, dt = mt.prefetch_frozen(im)
    assert torch.equal(xs, xt) and len(ds) == len(dt) == 1 and torch.equal(ds[0], dt[0])


def test_shared_frozen_prefix_changes_nothing():
    a, pa, sa = _run(False)
    b, pb, sb = _run(True)
    assert sb and not sa
    # the step is not bit-reproducible from run to run (fp32 atomics in the loss reductions), with or without sharing: same tolerance as two plain runs
    assert _close(a, b), (a, b)
    assert (pa - pb).abs().max().item() <= 1e-6 * pa.abs().max().item()


def test_shared_frozen_prefix_switches_itself_off_when_the_weights_differ():
    b, pb, sb = _run(True, steps=2, perturb=True)
    assert not sb
    a, pa, _ = _run(False, steps=2, perturb=True)
    assert _close(a, b), (a, b)
