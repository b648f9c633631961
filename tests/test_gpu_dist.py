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


@pytest.mark.parametrize("backend", ["torch", "abr"])
def test_single_rank_rccl_step_equals_local_step(backend):
    """backend "torch": torch.distributed.all_reduce under "nccl" (= RCCL); "abr": the library's own communicator and abr_allreduce_flat
    (csrc/comm.hip), one RCCL group per bucket on the reducer's stream."""
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
        opt.reducer.backend = backend
        opt.reducer.measure = collective
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
            d = opt.reducer.describe()
            assert [r["bucket"] for r in d] == ["roi_heads", "rpn", "backbone"] and all(r["backend"] == backend for r in d)
            waits = [r["main_stream_wait_ms_cumulative"] for r in d]
            assert all(w >= 0.0 for w in waits) and waits == sorted(waits), waits      # measured, cumulative in issue order
            opt.reducer.close()
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


def test_allreduce_flat_abi_single_rank():
    """abr_comm_* / abr_allreduce_flat straight through the C ABI: a one-rank communicator on this GPU, ranges summed in place (= unchanged),
    ranges outside the list untouched, bad ranges refused; include/abr_iod_hip.h section 8."""
    import ctypes as C
    from abr_iod_amd import _lib as L
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    lib = L.lib()
    assert lib.abr_comm_rccl_version() >= 20000
    idb = C.create_string_buffer(128)
    L.check(lib.abr_comm_unique_id(C.cast(idb, C.c_void_p)), "comm_unique_id")
    assert any(idb.raw)
    comm = C.c_void_p()
    L.check(lib.abr_comm_init(1, 0, C.cast(idb, C.c_void_p), C.byref(comm)), "comm_init")
    try:
        info = (C.c_int32 * 3)()
        L.check(lib.abr_comm_info(comm, C.cast(info, C.c_void_p)), "comm_info")
        assert list(info) == [1, 0, torch.cuda.current_device()]
        g = torch.Generator(device="cuda").manual_seed(0)
        buf = torch.randn(1 << 20, device="cuda", generator=g)
        want = buf.clone()
        ranges = (C.c_int64 * 6)(0, 1000, 5000, 5000, 70000, 1 << 20)      # (the middle range is empty)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        L.check(lib.abr_allreduce_flat(comm, buf.data_ptr(), C.cast(ranges, C.c_void_p), 3, side.cuda_stream), "allreduce_flat")
        side.synchronize()
        assert torch.equal(buf, want)
        bad = (C.c_int64 * 2)(10, 5)
        assert lib.abr_allreduce_flat(comm, buf.data_ptr(), C.cast(bad, C.c_void_p), 1, side.cuda_stream) == -1
        assert b"range 0" in lib.abr_last_error()
    finally:
        L.check(lib.abr_comm_destroy(comm), "comm_destroy")


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
    images, targets = synthetic_batch(1, 160, 224, seed=10 + rank)  # each rank its own image

    # (a) this rank's LOCAL, un-reduced gradient of the first step: same weights, same draws (recorded), exchange switched off
    ms, mt = build_models(cfg_s, cfg_t, seed=0)
    opt = make_optimizer(cfg_t, mt); sch = make_lr_scheduler(cfg_t, opt)
    opt.reducer.reduce_bucket_async = lambda name: None
    opt.reducer.finish = lambda: None
    torch.manual_seed(3 + rank); random.seed(3 + rank)
    train_step(ms, mt, images, targets, opt, sch, cfg_t)
    torch.cuda.synchronize()
    g_local = mt.flat.grads.clone().cpu().numpy()
    ev_rpn, ev_box = mt.rpn.loss_evaluator, mt.roi_heads.box.loss_evaluator
    pos, samp = ev_rpn.last_sampled
    draws = ((pos[pos >= 0].clone(), samp[samp >= 0].clone()), [t[t >= 0].clone() for t in ev_box.last_sampled_inds], [list(s) for s in ms.last_soften_indices])

    # (b) the real thing: two steps with the hook-issued exchange; the first replays the draws of (a)
    ms, mt = build_models(cfg_s, cfg_t, seed=0)                     # same weights on both ranks
    opt = make_optimizer(cfg_t, mt); sch = make_lr_scheduler(cfg_t, opt)
    assert opt.reducer.active and opt.world_size == 2
    sent, inner = [], opt.reducer.reduce_bucket_async
    opt.reducer.reduce_bucket_async = lambda name: (sent.append(name), inner(name))
    p_before = mt.flat.params.detach().clone()
    n = mt.flat.n_trainable
    lr = torch.zeros(n, device="cuda"); wd = torch.zeros(n, device="cuda")   # per element, at iteration 0 (warm-up factor included)
    for g_ in opt.param_groups:
        a, b = g_["range"]
        lr[a:b] = g_["lr"]; wd[a:b] = g_["weight_decay"]
    grads = []
    for it in range(2):
        if it == 0:
            mt.rpn.loss_evaluator.inject_sampled, mt.roi_heads.box.loss_evaluator.inject_sampled_inds, ms.inject_soften_indices = draws
        else:
            mt.rpn.loss_evaluator.inject_sampled = mt.roi_heads.box.loss_evaluator.inject_sampled_inds = ms.inject_soften_indices = None
        train_step(ms, mt, images, targets, opt, sch, cfg_t)
        torch.cuda.synchronize()
        grads.append(mt.flat.grads.clone().cpu().numpy())           # after step(): the all-reduced (summed) gradient
        if it == 0:
            p_after1 = mt.flat.params.detach().clone()
    # SGD's first step on the AVERAGED gradient (solver/build.py:7-21 groups: weights lr/wd, biases 2*lr / 0), restated in torch
    expect = p_before[:n] - lr * (torch.from_numpy(grads[0]).cuda() / world + wd * p_before[:n])
    upd_err = float((p_after1[:n] - expect).norm() / (p_after1[:n] - p_before[:n]).norm())
    out.put((rank, mt.flat.params.detach().cpu().numpy(), grads, sent, g_local, upd_err))   # (numpy: nothing shared by file descriptor)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_exchange_gradients_through_the_hooks():
    """Two processes, one image each, a real exchange (gloo over CUDA tensors, both ranks on this GPU): the hook-issued bucket
    all-reduces plus the one in step() must leave both ranks with the SAME gradient, that gradient must be g_rank0 + g_rank1 (each
    element reduced exactly once: a bucket sent twice or a wrong range would be identical on both ranks and still wrong), and the
    update must be SGD on the gradient averaged over the ranks."""
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from e2e_common import run_ranks
    got = sorted(run_ranks(ctx, _two_rank_worker, [(r, world, port) for r in range(world)], timeout=200), key=lambda t: t[0])
    (_, p0, g0, s0, l0, u0), (_, p1, g1, s1, l1, u1) = got
    # the hooks fired during backward on both ranks: pooled-input hook -> roi_heads; feature-map hook -> roi_heads (already sent), rpn
    assert s0[:3] == ["roi_heads", "roi_heads", "rpn"] and s1 == s0
    import numpy as np
    for a, b in zip(g0, g1):
        assert np.isfinite(a).all() and float(np.abs(a).max()) > 0
        assert np.array_equal(a, b)                                               # the same sum everywhere
    assert np.array_equal(p0, p1)
    want = l0 + l1                                                                # every element reduced exactly ONCE
    rel = float(np.linalg.norm(g0[0] - want) / np.linalg.norm(want))
    print("exchanged gradient vs g_rank0 + g_rank1: rel-L2", rel)
    assert np.array_equal(g0[0], want), rel                                       # one fp32 add per element, and (round 5) no atomically accumulated sum left in the step
    assert float(np.abs(g0[0] - want).max()) <= 1e-4 * float(np.abs(want).max())
    assert u0 < 1e-4 and u1 < 1e-4, (u0, u1)                                      # update == SGD on the averaged gradient


# ---------------------------------------------------------------------------------------------------------------------------------
# 1 rank x 2 images  ==  2 ranks x 1 image  on the REAL (full-width) model -- SURVEY.md §4: "distributed correctness = same loss /
# gradients for 1 vs N ranks with a fixed sampler".  Every random draw of the single-process run (RPN sampler, box-head sampler, the 64
# soften picks) is replayed on the rank that owns the image.  Reference: DistributedDataParallel averages the per-rank gradients
# (tools/train_incremental.py:231-235); here: sum all-reduce of the flat gradient + 1/world in the SGD kernel.
EQ_OVERRIDES = ["MODEL.RPN.PRE_NMS_TOP_N_TRAIN", 600, "MODEL.RPN.POST_NMS_TOP_N_TRAIN", 100, "MODEL.RPN.PRE_NMS_TOP_N_TEST", 300,
                "MODEL.RPN.POST_NMS_TOP_N_TEST", 150, "MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE", 48, "MODEL.RPN.BATCH_SIZE_PER_IMAGE", 64]
EQ_H, EQ_W = 160, 224


def _eq_setup(image_ids):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from e2e_common import clamp_targets
    from abr_iod_amd.engine.synthetic import build_models, make_cfgs, synthetic_batch
    from abr_iod_amd.solver.build import make_lr_scheduler, make_optimizer
    cfg_s, cfg_t = make_cfgs("15-5", dist_type="id", feat="ard", alpha=0.5, beta=1.0, gamma=1.0, overrides=EQ_OVERRIDES)   # full width
    ms, mt = build_models(cfg_s, cfg_t, seed=0)
    with torch.no_grad():   # target != source, identically in every process (same device generator, same seed)
        g = torch.Generator(device="cuda").manual_seed(5)
        n = mt.flat.n_trainable
        mt.flat.params[:n].mul_(1.0 + 0.05 * torch.randn(n, device="cuda", generator=g))
    from abr_iod_amd.modeling.backbone.resnet import bump_param_version
    bump_param_version()
    images, targets = synthetic_batch(2, EQ_H, EQ_W, seed=3, max_boxes=3)
    clamp_targets(targets, EQ_W, EQ_H)
    images, targets = images[image_ids], [targets[i] for i in image_ids]
    opt = make_optimizer(cfg_t, mt)
    return cfg_t, ms, mt, images, targets, opt, make_lr_scheduler(cfg_t, opt)


def _eq_rank_worker(rank, world, port, draws, out):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)   # a real 2-rank exchange (gloo moves CUDA tensors) on the one GPU
    torch.cuda.set_device(0)
    from abr_iod_amd.engine import train_step
    cfg_t, ms, mt, images, targets, opt, sch = _eq_setup([rank])
    assert opt.reducer.active and opt.world_size == world
    pos, samp, head, soften = draws[rank]
    mt.rpn.loss_evaluator.inject_sampled = (torch.tensor(pos).cuda(), torch.tensor(samp).cuda())
    mt.roi_heads.box.loss_evaluator.inject_sampled_inds = [torch.tensor(head).cuda()]
    ms.inject_soften_indices = [soften]
    ld, total = train_step(ms, mt, images, targets, opt, sch, cfg_t)
    torch.cuda.synchronize()
    props = mt.roi_heads.box.loss_evaluator.last_input_proposals[0].bbox.cpu().numpy()
    out.put((rank, mt.flat.grads.cpu().numpy(), mt.flat.params.detach().cpu().numpy(), {k: float(v) for k, v in ld.items()}, props))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(360)
def test_two_ranks_one_image_each_equal_one_rank_two_images():
    import numpy as np
    import torch.multiprocessing as mp
    from abr_iod_amd.engine import train_step
    # ---- one process, both images; the samplers draw freely and the draws are recorded
    cfg_t, ms, mt, images, targets, opt, sch = _eq_setup([0, 1])
    assert not opt.reducer.active
    p_before = mt.flat.params.detach().clone()
    ld, total = train_step(ms, mt, images, targets, opt, sch, cfg_t)
    torch.cuda.synchronize()
    g_single = mt.flat.grads.cpu().numpy()
    p_single = mt.flat.params.detach().cpu().numpy()
    assert float((mt.flat.params - p_before).abs().max()) > 0
    ev_rpn, ev_box = mt.rpn.loss_evaluator, mt.roi_heads.box.loss_evaluator
    n = ev_rpn.last_targets[0][0].numel()                      # anchors per image
    pos_all, samp_all = (t.cpu() for t in ev_rpn.last_sampled)
    pos_all, samp_all = pos_all[pos_all >= 0], samp_all[samp_all >= 0]
    draws, single_props = [], []
    for i in range(2):
        sel = lambda t: (t[(t >= i * n) & (t < (i + 1) * n)] - i * n).tolist()
        draws.append((sel(pos_all), sel(samp_all), [v for v in ev_box.last_sampled_inds[i].cpu().tolist() if v >= 0], list(ms.last_soften_indices[i])))
        single_props.append(ev_box.last_input_proposals[i].bbox.cpu().numpy())
        assert len(draws[i][1]) == 64 and len(draws[i][2]) == 48 and len(draws[i][3]) == 64   # equal shares: mean of means == global mean
    ld_single = {k: float(v) for k, v in ld.items()}

    # ---- two ranks, one image each, the same draws
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    from e2e_common import run_ranks
    got = sorted(run_ranks(ctx, _eq_rank_worker, [(r, world, port, draws) for r in range(world)], timeout=260), key=lambda t: t[0])
    (_, g0, p0, ld0, props0), (_, g1, p1, ld1, props1) = got
    # the per-image proposal lists the injected indices refer to are the same lists
    for w, sgl in ((props0, single_props[0]), (props1, single_props[1])):
        assert w.shape == sgl.shape, (w.shape, sgl.shape)
        assert np.array_equal(w, sgl), (float(np.abs(w - sgl).max()), np.nonzero(np.abs(w - sgl).max(1))[0][:10])
    assert np.array_equal(g0, g1) and np.array_equal(p0, p1)              # both ranks hold the same sum and the same parameters
    # losses: mean over ranks of the per-rank losses == the 2-image losses (engine/trainer.py:15-37 reduce_loss_dict semantics)
    for k in ld_single:
        assert abs(0.5 * (ld0[k] + ld1[k]) - ld_single[k]) <= 1e-5 * max(1.0, abs(ld_single[k])), (k, ld0[k], ld1[k], ld_single[k])
    # gradients: (g_rank0 + g_rank1) / world == gradient of the 2-image batch, to fp32 reduction order
    g_mean = 0.5 * g0
    rel = float(np.linalg.norm(g_mean - g_single) / np.linalg.norm(g_single))
    print("1x2 vs 2x1: gradient rel-L2", rel, " max-abs", float(np.abs(g_mean - g_single).max()), " |g|max", float(np.abs(g_single).max()))
    assert rel < 1e-6, rel   # (measured 2.0e-7: what the different per-rank reduction shapes -- one image's rows vs two -- explain; 2e-5 before round 5)
    # and the SGD update (1/world folded into the kernel) lands on the same parameters
    relp = float(np.linalg.norm(p0 - p_single) / np.linalg.norm(p_single - p_before.cpu().numpy()))
    print("parameter update rel-L2 difference", relp)
    assert relp < 2e-5, relp


# ---------------------------------------------------------------------------------------------------------------------------------
# FOUR ranks on the one GPU whose steps differ in everything the exchange must not depend on: which gradient hooks
# fire, whether the rank has a source model at all, how many images it holds.  solver/grad_reducer.py promises the SAME collective sequence
# on every rank whatever its local state; here the sequence each rank really issued is recorded and compared, and the exchanged gradient
# must be the sum of the four local gradients, element for element.
#   rank 0  ARD + ID step, hooks armed                                   (the benchmark's step)
#   rank 1  the same with the overlap switched off: NO hook sends, everything leaves from optimizer.step()
#   rank 2  finetune-shaped step: no source model, no second RoI pass, no distillation terms (one pooled input instead of two / a joint one)
#   rank 3  ARD + ID step on TWO images
UNEVEN_MODES = ("ard", "ard-no-overlap", "finetune", "ard-two-images")


def _uneven_rank_worker(rank, world, port, out):
    import random
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    mode = UNEVEN_MODES[rank]
    from abr_iod_amd.engine import train_step
    from abr_iod_amd.engine.synthetic import build_models, make_cfgs, synthetic_batch
    from abr_iod_amd.solver import grad_reducer
    from abr_iod_amd.solver.build import make_lr_scheduler, make_optimizer
    tiny = ["MODEL.RESNETS.STEM_OUT_CHANNELS", 16, "MODEL.RESNETS.RES2_OUT_CHANNELS", 32, "MODEL.RESNETS.WIDTH_PER_GROUP", 8,
            "MODEL.RESNETS.BACKBONE_OUT_CHANNELS", 128]
    if mode == "finetune":
        cfg_s, cfg_t = make_cfgs("15-5", dist_type="l2", feat="no", alpha=0.0, beta=0.0, gamma=0.0, overrides=tiny)
    else:
        cfg_s, cfg_t = make_cfgs("15-5", dist_type="id", feat="ard", alpha=0.5, overrides=tiny)
    images, targets = synthetic_batch(2 if mode == "ard-two-images" else 1, 160, 224, seed=20 + rank)
    if mode == "ard-no-overlap":
        grad_reducer.OVERLAP = False

    def models():
        ms, mt = build_models(cfg_s, cfg_t, seed=0)          # the same target weights on every rank (the finetune target has the same 21-class head)
        return (None if mode == "finetune" else ms), mt

    # (a) the LOCAL gradient of the step, exchange switched off; the draws are recorded and replayed in (b)
    ms, mt = models()
    opt = make_optimizer(cfg_t, mt); sch = make_lr_scheduler(cfg_t, opt)
    opt.reducer.reduce_bucket_async = lambda name: None
    opt.reducer.finish = lambda: None
    opt.reducer.begin = lambda: None
    torch.manual_seed(3 + rank); random.seed(3 + rank)
    real_all_reduce = dist.all_reduce
    dist.all_reduce = lambda t, *a, **k: None                 # (the range guard's flag exchange too: phase (a) is a local run)
    try:
        train_step(ms, mt, images, targets, opt, sch, cfg_t)
    finally:
        dist.all_reduce = real_all_reduce
    torch.cuda.synchronize()
    g_local = mt.flat.grads.clone().cpu().numpy()
    ev_rpn, ev_box = mt.rpn.loss_evaluator, mt.roi_heads.box.loss_evaluator
    pos, samp = ev_rpn.last_sampled
    draws = ((pos[pos >= 0].clone(), samp[samp >= 0].clone()), [t[t >= 0].clone() for t in ev_box.last_sampled_inds],
             [list(s) for s in ms.last_soften_indices] if ms is not None else None)

    # (b) the exchanged step, every collective this rank issues logged as (dtype, elements, offset inside its storage, reduce op)
    ms, mt = models()
    opt = make_optimizer(cfg_t, mt); sch = make_lr_scheduler(cfg_t, opt)
    assert opt.reducer.active and opt.world_size == world
    issued = []

    def logged_all_reduce(t, op=dist.ReduceOp.SUM, *a, **k):
        issued.append((str(t.dtype), int(t.numel()), int(t.storage_offset()) if t.numel() > 1 else -1, str(op)))
        return real_all_reduce(t, op, *a, **k)
    dist.all_reduce = logged_all_reduce
    grad_reducer.dist.all_reduce = logged_all_reduce
    hook_sends = []
    inner = opt.reducer.reduce_bucket_async
    opt.reducer.reduce_bucket_async = lambda name: (hook_sends.append(name), inner(name))
    mt.rpn.loss_evaluator.inject_sampled, mt.roi_heads.box.loss_evaluator.inject_sampled_inds = draws[0], draws[1]
    if ms is not None:
        ms.inject_soften_indices = draws[2]
    train_step(ms, mt, images, targets, opt, sch, cfg_t)
    torch.cuda.synchronize()
    out.put((rank, mode, issued, hook_sends, g_local, mt.flat.grads.cpu().numpy(), mt.flat.params.detach().cpu().numpy(),
             [(b, by, w) for b, by, w in opt.reducer.last_issue], int(mt.flat.n_trainable)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(420)
def test_four_ranks_with_uneven_hook_firing_issue_one_collective_sequence():
    import numpy as np
    import torch.multiprocessing as mp
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from e2e_common import run_ranks
    world, port = 4, _free_port()
    ctx = mp.get_context("spawn")
    got = sorted(run_ranks(ctx, _uneven_rank_worker, [(r, world, port) for r in range(world)], timeout=360), key=lambda t: t[0])
    seqs = [g[2] for g in got]
    for r in range(1, world):
        assert seqs[r] == seqs[0], "rank {} ({}) issued {} but rank 0 issued {}".format(r, got[r][1], seqs[r], seqs[0])
    # three gradient ranges (roi_heads, rpn, backbone) in BUCKET_ORDER, float32, sum -- whatever the hooks did
    fl = [s for s in seqs[0] if s[0] == "torch.float32" and s[1] > 1]
    assert len(fl) >= 3 and sum(s[1] for s in fl) == got[0][8], (fl, got[0][8])     # every trainable element in exactly one range
    # the ISSUE POINTS differ, as intended: rank 0 sent two buckets from hooks, rank 1 none
    where = {g[1]: [w for _, _, w in g[7]] for g in got}
    assert where["ard"][:2] == ["backward-hook", "backward-hook"] and where["ard"][2] == "optimizer.step", where
    assert where["ard-no-overlap"] == ["optimizer.step"] * 3, where
    # every rank holds the same gradient, and it is the sum of the four local ones: each element reduced exactly once
    for g in got[1:]:
        assert np.array_equal(g[5], got[0][5]) and np.array_equal(g[6], got[0][6])
    want = got[0][4].astype(np.float64) + got[1][4] + got[2][4] + got[3][4]
    have = got[0][5].astype(np.float64)
    assert np.isfinite(have).all() and float(np.abs(have).max()) > 0
    err = float(np.abs(have - want).max()) / float(np.abs(want).max())
    print("4 uneven ranks: exchanged gradient vs the sum of the local ones, max-abs / max", err, " issue points", where)
    assert err <= 1e-6, err            # (gloo's reduction order over four ranks is its own: fp32 rounding of a 4-term sum)
