"""CPU, world_size 2, gloo: the data-parallel exchange used by FusedSGD/bench.py -- ONE sum all-reduce of the flat gradient
buffer + 1/world scaling folded into the update -- equals single-process training on the concatenated batch, and
reduce_loss_dict matches the reference's semantics (engine/trainer.py:15-37).  The HIP SGD kernel itself is GPU-only, so the
update rule is restated here in torch on the same flat layout; the GPU kernel is checked against torch.optim.SGD in
tests/test_gpu_e2e.py and tests/test_gpu_ops.py."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from abr_iod_amd.engine.trainer import reduce_loss_dict
    from abr_iod_amd.modeling._flat import flatten_parameters
    from abr_iod_amd.utils.comm import get_rank, get_world_size

    assert get_world_size() == world and get_rank() == rank
    torch.manual_seed(0)  # same init on every rank (what loading the same checkpoint does)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))
    flat = flatten_parameters(model)
    assert flat.n_trainable >= sum(p.numel() for p in model.parameters())
    for p in model.parameters():  # parameters and grads are views of the flat buffers
        assert p.data_ptr() >= flat.params.data_ptr() and p.grad.data_ptr() >= flat.grads.data_ptr()
    g = torch.Generator().manual_seed(123)
    x_all, y_all = torch.randn(8, 8, generator=g), torch.randn(8, 3, generator=g)
    shard = slice(rank * 4, rank * 4 + 4)  # DistributedSampler: contiguous rank slice (data/samplers/distributed.py:42-60)
    lr, wd, mu = 0.1, 1e-4, 0.9
    mom = torch.zeros_like(flat.grads)
    losses = []
    for step in range(3):
        flat.zero_grad()
        loss = ((model(x_all[shard]) - y_all[shard]) ** 2).mean()
        grads = torch.autograd.grad(loss, list(model.parameters()))
        for p, gr in zip(model.parameters(), grads):
            p.grad.add_(gr)                       # what the wgrad kernels do: accumulate into the flat views
        dist.all_reduce(flat.grads, op=dist.ReduceOp.SUM)      # FusedSGD.all_reduce_grads
        d = flat.grads / world + wd * flat.params[: flat.n_trainable]
        mom = d if step == 0 else mu * mom + d
        flat.params[: flat.n_trainable].sub_(lr * mom)
        losses.append(loss.detach())
    red = reduce_loss_dict({"b": losses[-1].clone(), "a": losses[0].clone()})
    if rank == 0:
        out.put((flat.params.detach().numpy().copy(), {k: float(v) for k, v in red.items()}, [float(l) for l in losses]))   # numpy: nothing shared by file descriptor
    else:
        out.put((None, None, [float(l) for l in losses]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_flat_allreduce_matches_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from e2e_common import run_ranks
    got = run_ranks(ctx, _worker, [(r, world, port) for r in range(world)], timeout=150)
    params0, red, _ = next(g for g in got if g[0] is not None)
    per_rank_losses = [g[2] for g in got]

    # single process on the concatenated batch with torch.optim.SGD
    sys.path.insert(0, ROOT)
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))
    g = torch.Generator().manual_seed(123)
    x_all, y_all = torch.randn(8, 8, generator=g), torch.randn(8, 3, generator=g)
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    for step in range(3):
        opt.zero_grad()
        # per-rank losses are means over 4 samples; DDP averages gradients -> equals the mean over all 8
        (((model(x_all) - y_all) ** 2).mean()).backward()
        opt.step()
    ref = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    from abr_iod_amd.modeling._flat import flatten_parameters
    flat = flatten_parameters(model)  # same layout as in the workers
    assert torch.allclose(torch.from_numpy(params0), flat.params, rtol=1e-5, atol=1e-6)
    # reduce_loss_dict: rank 0 holds the mean over ranks, keys sorted (engine/trainer.py:27-36)
    assert abs(red["a"] - sum(l[0] for l in per_rank_losses) / world) < 1e-6
    assert abs(red["b"] - sum(l[-1] for l in per_rank_losses) / world) < 1e-6


def _eval_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from abr_iod_amd.engine.inference import _accumulate_predictions_from_multiple_gpus
    from abr_iod_amd.structures.bounding_box import BoxList
    mine = {}
    for i in range(rank, 6, world):  # each rank evaluated every world-th image
        b = BoxList(torch.full((i + 1, 4), float(i)), (100 + i, 50), mode="xyxy")
        b.add_field("scores", torch.arange(i + 1, dtype=torch.float32))
        mine[i] = b
    merged = _accumulate_predictions_from_multiple_gpus(mine)
    out.put((rank, None if merged is None else [(len(p), p.size, float(p.bbox[0, 0]), p.get_field("scores").tolist()) for p in merged]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_eval_predictions_merge_across_ranks():
    """Sharded evaluation (one process per GPU): rank 0 ends up with every image's detections ordered by image id, the other
    ranks with None (engine/inference.py:143-160)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from e2e_common import run_ranks
    got = dict(x for x in run_ranks(ctx, _eval_worker, [(r, world, port) for r in range(world)], timeout=150))
    assert got[1] is None
    assert got[0] == [(i + 1, (100 + i, 50), float(i), [float(v) for v in range(i + 1)]) for i in range(6)]


SEGMENTS = [  # the layout flatten_parameters produces: fused-head groups first, then named_parameters order (trainable region)
    ("rpn.head.cls_logits.weight", 0, 128, False), ("roi_heads.box.predictor.cls_score.weight", 128, 320, False),
    ("backbone.body.layer2.0.conv1.weight", 320, 448, False), ("backbone.body.layer3.5.conv3.weight", 448, 704, False),
    ("rpn.head.conv.weight", 704, 960, False), ("rpn.head.conv.bias", 960, 1024, True),
    ("roi_heads.box.feature_extractor.head.layer4.0.conv1.weight", 1024, 1536, False),
    ("roi_heads.box.feature_extractor.head.layer4.2.conv3.weight", 1536, 2048, False),
]


def test_grad_buckets_cover_the_flat_buffer_once():
    sys.path.insert(0, ROOT)
    from abr_iod_amd.solver.grad_reducer import BUCKET_ORDER, make_buckets
    b = make_buckets(SEGMENTS)
    assert b["roi_heads"] == [(128, 320), (1024, 2048)] and b["rpn"] == [(0, 128), (704, 1024)] and b["backbone"] == [(320, 704)]
    covered = sorted(r for name in BUCKET_ORDER for r in b[name])
    assert covered[0][0] == 0 and covered[-1][1] == 2048 and all(x[1] == y[0] for x, y in zip(covered, covered[1:]))


def _reducer_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from abr_iod_amd.solver.grad_reducer import GradReducer
    results = []
    for early in ((), ("roi_heads",), ("roi_heads", "rpn"), ("rpn", "roi_heads", "rpn")):
        grads = torch.arange(2048, dtype=torch.float32) * (rank + 1)
        red = GradReducer(grads, SEGMENTS)
        assert red.active
        red.begin()
        for name in early:              # what the trainer's gradient hooks do during backward
            red.reduce_bucket_async(name)
        red.finish()                    # FusedSGD.all_reduce_grads
        red.begin()                     # the next step starts clean
        results.append(grads.clone().numpy())   # (numpy: a torch tensor on a queue is shared by file descriptor and the worker may exit first)
    out.put((rank, results))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_bucketed_overlapped_allreduce_equals_one_allreduce():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from e2e_common import run_ranks
    got = run_ranks(ctx, _reducer_worker, [(r, world, port) for r in range(world)], timeout=150)
    want = torch.arange(2048, dtype=torch.float32) * 3   # rank 0 holds 1x, rank 1 holds 2x: every element summed exactly once
    for rank, results in got:
        for r in results:
            assert torch.equal(torch.from_numpy(r), want)



def test_modelled_ring_time_of_the_gradient_buckets():
    """solver/grad_reducer.py::modelled_ring_allreduce_ms (printed per bucket in bench.py's N > 1 line as a PLANNING figure): zero for one rank, 2 (w - 1) / w
    of the bytes over min(7, w - 1) links of 153 GB/s plus 8 us per ring step otherwise; monotone in bytes; the 33 MB backbone bucket at w = 8 is ~0.17 ms."""
    from abr_iod_amd.solver.grad_reducer import modelled_ring_allreduce_ms as f
    assert f(33_000_000, 1) == 0.0
    t8 = f(33_000_000, 8)
    want = 2 * 7 / 8 * 33e6 / (153e9 * 7) * 1e3 + 2 * 7 * 8e-3
    assert abs(t8 - want) < 1e-9 and 0.1 < t8 < 0.3
    assert f(66_000_000, 8) > t8 > f(33_000_000, 2) * 0 and f(33_000_000, 2) > f(1_000_000, 2)
