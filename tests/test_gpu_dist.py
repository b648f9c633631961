"""GPU, one rank: the data-parallel step through RCCL itself (backend "nccl" on ROCm).  A 1-GPU box cannot show scaling, but it
does exercise what bench.py --gpus N does on every rank: process-group initialisation over 127.0.0.1, the all-reduce of the flat
gradient buffer in three buckets -- the RoI-head and RPN buckets from gradient hooks during backward, ordered after the side-stream
weight gradients (solver/grad_reducer.py), the backbone bucket in optimizer.step() -- and the 1/world scaling in the update; with
one rank the result must equal the run without the collective."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_single_rank_rccl_step_equals_local_step():
    from abr_iod_amd.engine import train_step
    from abr_iod_amd.engine.synthetic import build_models, make_cfgs, synthetic_batch
    from abr_iod_amd.solver.build import make_lr_scheduler, make_optimizer
    tiny = ["MODEL.RESNETS.STEM_OUT_CHANNELS", 16, "MODEL.RESNETS.RES2_OUT_CHANNELS", 32, "MODEL.RESNETS.WIDTH_PER_GROUP", 8,
            "MODEL.RESNETS.BACKBONE_OUT_CHANNELS", 128]
    images, targets = synthetic_batch(2, 160, 224, seed=1)

    def run(collective):
        import random
        cfg_s, cfg_t = make_cfgs("15-5", dist_type="id", feat="ard", alpha=0.5, overrides=tiny)
        ms, mt = build_models(cfg_s, cfg_t, seed=0)
        opt = make_optimizer(cfg_t, mt)
        opt.force_all_reduce = collective
        sent, inner = [], opt.reducer.reduce_bucket_async
        opt.reducer.reduce_bucket_async = lambda name: (sent.append((name, name in opt.reducer._done)), inner(name))
        sch = make_lr_scheduler(cfg_t, opt)
        torch.manual_seed(3); random.seed(3)
        for _ in range(3):
            train_step(ms, mt, images, targets, opt, sch, cfg_t)
        torch.cuda.synchronize()
        if collective:  # the trainer's gradient hooks sent the RoI-head bucket, then the RPN bucket, DURING each backward pass
            first = [name for name, already in sent if not already]
            assert first == ["roi_heads", "rpn"] * 3, sent
        else:
            assert sent == []
        return mt.flat.params.clone()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:{}".format(_free_port()), rank=0, world_size=1)
    try:
        with_rccl = run(True)
    finally:
        dist.destroy_process_group()
    local = run(False)
    assert torch.isfinite(with_rccl).all()
    # atomically accumulated weight gradients make two runs differ in the last bits; the collective adds nothing on top
    assert float((with_rccl - local).norm() / local.norm()) < 1e-5


def _two_rank_worker(rank, world, port, out):
    import random
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)   # gloo moves CUDA tensors too: a real 2-rank exchange on one GPU
    torch.cuda.set_device(0)
    from abr_iod_amd.engine import train_step
    from abr_iod_amd.engine.synthetic import build_models, make_cfgs, synthetic_batch
    from abr_iod_amd.solver.build import make_lr_scheduler, make_optimizer
    tiny = ["MODEL.RESNETS.STEM_OUT_CHANNELS", 16, "MODEL.RESNETS.RES2_OUT_CHANNELS", 32, "MODEL.RESNETS.WIDTH_PER_GROUP", 8,
            "MODEL.RESNETS.BACKBONE_OUT_CHANNELS", 128]
    cfg_s, cfg_t = make_cfgs("15-5", dist_type="id", feat="ard", alpha=0.5, overrides=tiny)
    ms, mt = build_models(cfg_s, cfg_t, seed=0)                     # same weights on both ranks
    opt = make_optimizer(cfg_t, mt); sch = make_lr_scheduler(cfg_t, opt)
    assert opt.reducer.active and opt.world_size == 2
    sent, inner = [], opt.reducer.reduce_bucket_async
    opt.reducer.reduce_bucket_async = lambda name: (sent.append(name), inner(name))
    images, targets = synthetic_batch(1, 160, 224, seed=10 + rank)  # each rank its own image
    torch.manual_seed(3 + rank); random.seed(3 + rank)
    grads = []
    for _ in range(2):
        train_step(ms, mt, images, targets, opt, sch, cfg_t)
        torch.cuda.synchronize()
        grads.append(mt.flat.grads.clone().cpu().numpy())           # after step(): the all-reduced (summed) gradient
    out.put((rank, mt.flat.params.detach().cpu().numpy(), grads, sent))   # (numpy: nothing shared by file descriptor)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_exchange_gradients_through_the_hooks():
    """Two processes, one image each, a real exchange (gloo over CUDA tensors, both ranks on this GPU): the hook-issued bucket
    all-reduces plus the one in step() must leave both ranks with the SAME summed gradient and the same parameters."""
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=500) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, p0, g0, s0), (_, p1, g1, s1) = got
    # the hooks fired during backward on both ranks: pooled-input hook -> roi_heads; feature-map hook -> roi_heads (already sent), rpn
    assert s0[:3] == ["roi_heads", "roi_heads", "rpn"] and s1 == s0
    import numpy as np
    for a, b in zip(g0, g1):
        assert np.isfinite(a).all() and float(np.abs(a).max()) > 0
        assert np.array_equal(a, b)                                               # every element reduced exactly once, same sum everywhere
    assert np.array_equal(p0, p1)
