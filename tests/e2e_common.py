"""Shared by the end-to-end parity tests and the golden generators: the BASELINE.json configurations as the reference's
launch scripts spell them, and the deterministic (CPU-generator) weight perturbation that makes target != source.

    name        BASELINE.json  reference command line
    finetune    configs[1]     tools/train_incremental.py -t 15-5 (no --feat/--dist_type/-alpha/-beta: argparse defaults
                               feat 'no', dist_type 'l2', alpha 0, beta 0 -- train_incremental.py:334-365; scripts/run_SI.sh:26 "Finetune")
    15-5        configs[2]     --feat ard -gamma 1.0 --dist_type id -alpha 0.5 -beta 1.0        (scripts/run_SI.sh:24-25)
    10-10       configs[3]     --feat ard -gamma 1.0 --dist_type id -alpha 0.1 -beta 0.5        (scripts/run_SI.sh:30-32)
    10-5        configs[4]     --feat ard -gamma 1.0 --dist_type id -alpha 1.0 -beta 1.0, step 1 (scripts/run_MI.sh:11-21)
"""
import torch

#            task,    dist_type, feat, alpha, beta, gamma, label_range (ids of the NEW classes), n_old
CONFIGS = {
    "finetune": ("15-5", "l2", "no", 0.0, 0.0, 0.0, (16, 21), 15),
    "15-5": ("15-5", "id", "ard", 0.5, 1.0, 1.0, (16, 21), 15),
    "10-10": ("10-10", "id", "ard", 0.1, 0.5, 1.0, (11, 21), 10),
    "10-5": ("10-5", "id", "ard", 1.0, 1.0, 1.0, (11, 16), 10),
}


def needs_source(name):
    _, _, feat, alpha, _, _, _, _ = CONFIGS[name]
    return alpha > 0 or feat == "ard"


def perturb_trainable(sd, trainable_names, seed=5, rel=0.05):
    """In place on a reference-layout state_dict (CPU tensors): every trainable tensor *= 1 + rel * N(0,1), drawn from a CPU
    generator in `trainable_names` order -- identical in the build container (golden generators) and on the GPU box (tests)."""
    g = torch.Generator().manual_seed(seed)
    for name in trainable_names:
        v = sd[name]
        v.mul_(1.0 + rel * torch.randn(v.shape, generator=g, dtype=torch.float32).to(v.device))
    return sd


def clamp_targets(targets, w, h, min_side=8):
    """keep synthetic GT inside a (small) image and at least `min_side` wide / tall"""
    for t in targets:
        t.bbox[:, 0::2].clamp_(max=w - 1)
        t.bbox[:, 1::2].clamp_(max=h - 1)
        t.bbox[:, 2] = torch.max(t.bbox[:, 2], t.bbox[:, 0] + min_side).clamp(max=w - 1)
        t.bbox[:, 3] = torch.max(t.bbox[:, 3], t.bbox[:, 1] + min_side).clamp(max=h - 1)
    return targets


def match_fraction(a, b, atol=1e-2):
    """fraction of the rows of `a` [n,4] that have a row of `b` [m,4] within atol (max-abs): proposal lists are compared as SETS,
    because at 12000 -> 2000 boxes a near-tie in the fp32 objectness ranking or an IoU within rounding of the NMS threshold
    inserts / drops single boxes and shifts every later position (seen between two CPU runs of the same arithmetic as well)."""
    import numpy as np
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    if len(a) == 0:
        return 1.0
    hit = 0
    for lo in range(0, len(a), 256):
        d = np.abs(a[lo:lo + 256, None, :] - b[None, :, :]).max(-1)
        hit += int((d.min(1) <= atol).sum())
    return hit / len(a)


def run_ranks(ctx, target, argsets, timeout):
    """start one process per argument tuple, collect one queue item per rank, and ALWAYS reap the children: a rank that died or wedged in
    a collective must fail the test, not leave orphans that block the interpreter's exit (daemonic + terminate/kill in `finally`)"""
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=tuple(a) + (q,), daemon=True) for a in argsets]
    for p in procs:
        p.start()
    try:
        import queue
        import time
        got, deadline = [], time.time() + timeout
        while len(got) < len(procs):
            try:
                got.append(q.get(timeout=2.0))
            except queue.Empty:
                # a rank that DIED (exception, abort) never reports: fail now with its exit code instead of waiting out the other rank's collective
                dead = [(i, p.exitcode) for i, p in enumerate(procs) if p.exitcode not in (None, 0)]
                assert not dead, "rank(s) exited without a result: {}".format(dead)
                assert time.time() < deadline, "no result from {} of {} ranks after {} s (alive: {})".format(
                    len(procs) - len(got), len(procs), timeout, [p.is_alive() for p in procs])
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0, p.exitcode
        return got
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
                p.join(5)
                if p.is_alive():
                    p.kill()
