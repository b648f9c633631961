"""CPU: WarmupMultiStepLR reproduces the reference's learning-rate sequence (tests/golden/lr_schedule.json, written by
tests/golden/make_golden_lr.py from maskrcnn_benchmark/solver/lr_scheduler.py:10-52 stepped as tools/train_incremental.py:146-147
does): linear and constant warm-up, two milestones, and the configs/voc schedule around its warm-up end and its milestone."""
import json
import os

import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lr_schedule.json")


class _Opt(object):
    """the two kinds of param group FusedSGD holds (solver/build.py): a weight group and a bias group (lr x BIAS_LR_FACTOR)"""

    def __init__(self, base_lr):
        self.param_groups = [{"lr": base_lr, "initial_lr": base_lr}, {"lr": 2 * base_lr, "initial_lr": 2 * base_lr}]


@pytest.mark.parametrize("case", ["short", "const", "voc"])
def test_lr_sequence_equals_reference(case):
    from abr_iod_amd.solver.lr_scheduler import WarmupMultiStepLR
    g = json.load(open(GOLDEN))[case]
    base_lr, milestones, gamma, wf, wi, method = g["args"]
    opt = _Opt(base_lr)
    sch = WarmupMultiStepLR(opt, milestones, gamma, warmup_factor=wf, warmup_iters=wi, warmup_method=method)
    want = {int(k): v for k, v in g["lr"].items()}
    for it in range(max(want) + 1):
        if it in want:   # the lr the optimizer step of iteration `it` uses
            got = [opt.param_groups[0]["lr"], opt.param_groups[1]["lr"]]
            assert got == pytest.approx(want[it], rel=1e-12, abs=0), (it, got, want[it])
        sch.step()
    assert sch.get_last_lr() == [opt.param_groups[0]["lr"], opt.param_groups[1]["lr"]]


def test_scheduler_state_round_trip_has_torch_bookkeeping():
    from abr_iod_amd.solver.lr_scheduler import WarmupMultiStepLR
    opt = _Opt(0.01)
    sch = WarmupMultiStepLR(opt, (30, 40), 0.1, warmup_iters=10)
    for _ in range(17):
        sch.step()
    sd = sch.state_dict()
    # keys torch's _LRScheduler.state_dict() carries in a reference checkpoint (utils/checkpoint.py:41-45 stores scheduler.state_dict())
    for k in ("milestones", "gamma", "warmup_factor", "warmup_iters", "warmup_method", "base_lrs", "last_epoch", "_step_count", "_last_lr"):
        assert k in sd, k
    assert sd["last_epoch"] == 17 and sd["_step_count"] == 18
    opt2 = _Opt(0.01)
    sch2 = WarmupMultiStepLR(opt2, (30, 40), 0.1, warmup_iters=10)
    sch2.load_state_dict(sd)
    sch.step(); sch2.step()
    assert [g["lr"] for g in opt.param_groups] == [g["lr"] for g in opt2.param_groups]
    with pytest.raises(ValueError):   # lr_scheduler.py:21-25
        WarmupMultiStepLR(_Opt(0.01), (40, 30))
    with pytest.raises(ValueError):   # :27-31
        WarmupMultiStepLR(_Opt(0.01), (30, 40), warmup_method="cosine")
