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


def oracle_full_size_step(g, name, sd_s, sd_t, images, with_grad=True, H=600, W=1000):
    """The torch-CPU oracle's full-size step on the REFERENCE's draws stored in fixture `g` (tests/golden/e2e_full_<name>.npz:
    RPN sampler lists, proposal lists, box-head sampler lists, the 64 soften picks): the six losses of
    train_incremental.py:82-116 and -- with_grad -- autograd's gradient of their weighted sum w.r.t. the 52 trainable tensors
    (ROIAlign backward from oracle.c).  Returns (losses dict, total, RefModel of the target)."""
    import numpy as np
    from oracle import ops as O
    from oracle import torch_ref as R
    from oracle.model_ref import RefModel
    _, dist_type, _, alpha, beta, gamma, _, n_old = CONFIGS[name]
    mt = RefModel(sd_t) if with_grad else RefModel(sd_t, trainable_prefixes=())
    ms = RefModel(sd_s, trainable_prefixes=()) if sd_s is not None else None
    ctx = torch.enable_grad() if with_grad else torch.no_grad()
    with ctx:
        ft = mt.backbone(images)
        obj, reg = mt.rpn_head(ft)
        fh, fw = ft.shape[-2:]
        anchors, vis = O.grid_anchors(O.cell_anchors(), fh, fw, 16, (H, W))
        n = anchors.shape[0]
        NB = int(g["batch"]) if "batch" in g else 2
        gts = [g[f"gt{i}"] for i in range(NB)]
        labs, tgts, posm, negm = [], [], torch.zeros(NB, n, dtype=torch.bool), torch.zeros(NB, n, dtype=torch.bool)
        for i in range(NB):
            lab, tgt, _ = R.rpn_prepare_targets(anchors, vis, gts[i])
            labs.append(torch.from_numpy(lab)); tgts.append(torch.from_numpy(tgt))
            posm[i, torch.from_numpy(g[f"rpn_pos{i}"]).long()] = True
            negm[i, torch.from_numpy(g[f"rpn_neg{i}"]).long()] = True
        lo, lb = R.rpn_loss(obj, reg, torch.stack(labs), torch.stack(tgts), posm, negm)
        rois, labels, rts = [], [], []
        for i in range(NB):
            boxes = g[f"tgt_props{i}"]
            m = O.matcher(O.box_iou(gts[i], boxes), 0.5, 0.5, False)
            lab = g[f"gt_labels{i}"][np.clip(m, 0, None)].astype(np.int64)
            lab[m == -1] = 0; lab[m == -2] = -1
            tgt = O.box_encode(gts[i][np.clip(m, 0, None)], boxes, (10.0, 10.0, 5.0, 5.0))
            sel = g[f"head_sel{i}"].astype(np.int64)
            rois.append(np.concatenate([np.full((len(sel), 1), i, np.float32), boxes[sel]], 1))
            labels.append(lab[sel]); rts.append(tgt[sel])
        _, logits, boxreg = mt.box_head(ft, torch.from_numpy(np.concatenate(rois)))
        lc, lbox = R.box_head_loss(logits, boxreg, torch.from_numpy(np.concatenate(labels)), torch.from_numpy(np.concatenate(rts)), dist_type, n_old)
        losses = dict(loss_classifier=lc, loss_box_reg=lbox, loss_objectness=lo, loss_rpn_box_reg=lb)
        total = lc + lbox + lo + lb
        if ms is not None:
            with torch.no_grad():
                fs = ms.backbone(images)
                rois64 = torch.from_numpy(np.concatenate([np.concatenate([np.full((64, 1), i, np.float32), g[f"src_top128_{i}"][g[f"soften_sel{i}"]]], 1)
                                                          for i in range(NB)]))
                ps, zs, bs = ms.box_head(fs, rois64)
            pt, zt, bt = mt.box_head(ft, rois64)
            k_all = zt.shape[1]
            l_id = R.roi_distillation_loss(zs, bs.view(-1, n_old + 1, 4), zt, bt.view(-1, k_all, 4), dist_type)
            l_ard = R.ard_loss(ps, pt, gamma)
            losses["loss_id"], losses["loss_ard"] = l_id, l_ard
            total = total + alpha * l_id + beta * l_ard
        if with_grad:
            total.backward()
    return {k: float(v.detach()) for k, v in losses.items()}, float(total.detach()), mt
