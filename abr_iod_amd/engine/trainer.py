"""Incremental trainer loop.

Mirror of tools/train_incremental.py:55-181 (`do_train`) and maskrcnn_benchmark/engine/trainer.py:15-37 (`reduce_loss_dict`).
`train_step` is the body of one loop iteration (train_incremental.py:77-147) factored out so that the benchmark and the
smoke test run exactly the code the loop runs.

Work the reference does whose results are never used is skipped, with identical outputs:
  * with DIST.ALPHA == 0 and DIST.FEAT != 'ard' (the finetune configs) the source pass and the second RoI pass feed nothing;
  * `model_source.roi_heads.box.loss_evaluator.subsample(soften_proposal, targets)` (train_incremental.py:86) is dead: its result
    is overwritten with None at :102.  It is still executed when `faithful_rng=True` because it consumes device RNG.
Data parallelism: one process per GPU; gradients are summed over ranks by three large RCCL all-reduces of the flat gradient buffer
(DistributedDataParallel in the reference, train_incremental.py:231-235): the RoI-head and RPN buckets leave during backward from
gradient hooks (`_arm_overlap`, solver/grad_reducer.py), the backbone bucket inside optimizer.step().
"""
import datetime
import logging
import time

import torch
import torch.distributed as dist

from ..distillation.distillation import (calculate_attentive_roi_feature_distillation, calculate_feature_distillation_loss,
                                         calculate_roi_distillation_losses, calculate_rpn_distillation_loss)
import os

from ..utils.comm import get_world_size
from .. import ops as _ops

# (ABR_JOINT_ROI, below) Distillation RoIs through layer4 together with the detection RoIs (GeneralizedRCNN.forward_joint / forward_finish(...,
# soften_proposals=)).  Round 1 (fp32 kernels, 2 workgroups per CU) measured it 0.5 ms slower; with the bf16x6 weights-direct kernels (3 workgroups
# per CU: 288 m-tiles x 16 n-tiles = exactly 6 rounds) the second pass's M = 4096-row launches at ~half the big pass's rate cost more than they
# fill: ON by default since round 4 (-0.25 ms at B = 4, -0.15 ms at B = 2, same-session A/Bs in MEASUREMENTS.md).
SOURCE_OVERLAP = os.environ.get("ABR_SOURCE_OVERLAP", "1") != "0"
# the frozen source model's backbone + RPN head on a stream of their own, NEXT to the target's forward (its many small layer1-3 kernels
# leave CUs idle that the other model's kernels fill, as the dgrad / wgrad pair does in the backward pass)
SOURCE_STREAM = os.environ.get("ABR_SOURCE_STREAM", "1") != "0"
# ... and its 64-RoI head pass too (next to the target's 2048-RoI pass); 0: on the main stream between the target's two forward halves
SOURCE_HEAD_STREAM = os.environ.get("ABR_SOURCE_HEAD_STREAM", "1") != "0"
# software pipelining across steps: when the caller names the NEXT batch (`train_step(..., next_images=)`; `do_train` looks one batch ahead),
# the frozen source model's backbone + RPN head for it are enqueued on the source stream before this step's backward pass and run next to it
# -- their result does not depend on this step's update.  Same work per step, same results; 0 = off.
EARLY_SECOND_PASS = os.environ.get("ABR_EARLY_SECOND_PASS", "1") != "0"
PIPELINE_TARGET_FROZEN = os.environ.get("ABR_PIPELINE_TARGET_FROZEN", "1") != "0"
PIPELINE_SOURCE = os.environ.get("ABR_PIPELINE_SOURCE", "1") != "0"
# the next batch's source-model prefetch right behind the current batch's head pass on the source stream (-0.39 ms per step against issuing it
# before the backward pass): see train_step
EARLY_PREFETCH = os.environ.get("ABR_EARLY_PREFETCH", "1") != "0"
# Opt-in (off by default, and off in bench.py's headline): when the source and the target model hold IDENTICAL frozen stem / layer1 weights
# (the reference's setup: both are loaded from the same checkpoint and FREEZE_CONV_BODY_AT = 2 never lets them move), that prefix is the same
# function of the same batch in both models -- compute it once per batch and feed both.  Verified by comparing the tensors, never assumed.
SHARE_FROZEN_PREFIX = [os.environ.get("ABR_SHARE_FROZEN_PREFIX", "0") != "0"]
JOINT_ROI_PASS = os.environ.get("ABR_JOINT_ROI", "1") != "0"

class TrainerState(object):
    """What one (source, target) training pair carries from step to step: the work prefetched for the next batch and the bf16x6 range
    watch.  It lives on the TARGET model object (`trainer_state(model_target)`), not at module level: two trainers in one process do not
    see each other's prefetch, and a model that is dropped takes its state with it."""

    def __init__(self):
        self.prefetched = {}     # {"key": ..., "state": soften_begin(next batch), "target_prefix": (event, frozen_prefix(next batch))}
        self.x6_watch = None
        self.x6_tiny_logged = False
        self.share_key = None    # (weight versions) the frozen-prefix comparison below was made for
        self.share_ok = False

    def drop_prefetch(self):
        self.prefetched = {}


def trainer_state(model_target):
    st = model_target.__dict__.get("_abr_trainer_state")
    if st is None:
        st = TrainerState()
        model_target.__dict__["_abr_trainer_state"] = st
    return st


def frozen_prefix_shareable(model_source, model_target):
    """True iff the two backbones' frozen prefix (stem + leading stages without trainable parameters) is the same function: the same stages are
    frozen in both, same arithmetic, and every parameter and buffer of them compares equal.  The comparison runs once per weight version
    (resnet._STATIC_VERSION moves on checkpoint loads / in-place surgery; training never touches frozen tensors)."""
    from ..modeling.backbone import resnet
    st = trainer_state(model_target)
    key = (id(model_source), resnet._STATIC_VERSION[0], getattr(model_source, "conv_math", None), getattr(model_target, "conv_math", None))
    if st.share_key == key:
        return st.share_ok
    ok = False
    bs, bt = getattr(getattr(model_source, "backbone", None), "body", None), getattr(getattr(model_target, "backbone", None), "body", None)
    if bs is not None and bt is not None and hasattr(bs, "frozen_stage_names") and getattr(model_source, "conv_math", 0) == getattr(model_target, "conv_math", 1):
        ns, nt = bs.frozen_stage_names(), bt.frozen_stage_names()
        if ns is not None and ns == nt:
            ok = True
            for name in ["stem"] + list(nt):
                ms_, mt_ = getattr(bs, name), getattr(bt, name)
                ts_ = list(ms_.named_parameters()) + list(ms_.named_buffers())
                tt_ = dict(list(mt_.named_parameters()) + list(mt_.named_buffers()))
                if len(ts_) != len(tt_) or any(n not in tt_ or v.shape != tt_[n].shape or v.dtype != tt_[n].dtype or not torch.equal(v, tt_[n]) for n, v in ts_):
                    ok = False
                    break
    st.share_key, st.share_ok = key, ok
    return ok


def _prefetch_key(images, model_source, model_target):
    """A prefetch is only valid for the batch it was computed from AND the weights / arithmetic it was computed with: the batch object
    (identity) plus its storage address and in-place version counter (a buffer refilled in place is a different batch), the weight
    versions (a checkpoint load or in-place surgery between two steps moves _STATIC_VERSION; optimiser steps only move the trained
    tensors, which no prefetched quantity reads) and both models' contraction arithmetic."""
    from ..modeling.backbone import resnet
    t = images.tensors if hasattr(images, "tensors") else images
    return (id(images), t.data_ptr(), t._version, tuple(t.shape), id(model_source), id(model_target), resnet._STATIC_VERSION[0],
            getattr(model_source, "conv_math", None), getattr(model_target, "conv_math", None))


# f16x3: the share of operand elements more than 18 binades below their tensor's amax above which the trainer moves to bf16x6, and how often
# (in steps) the device counters are read (0 = never)
H3_MAX_SMALL_FRACTION = float(os.environ.get("ABR_H3_MAX_SMALL_FRACTION", "0.05"))
H3_STATS_EVERY = int(os.environ.get("ABR_H3_STATS_EVERY", "8"))

# ABR_X6_STRICT=1: leave the bf16x6 arithmetic as soon as ANY operand element falls below 2^-110 (round 2's policy), not only on inf / nan
X6_STRICT = os.environ.get("ABR_X6_STRICT", "0") != "0"


def _x6_guard(model_source, model_target, log=None):
    """bf16x6 arithmetic only: poll the kernels' range guard (ops.X6RangeWatch -- asynchronous, no host stall).

    NONFINITE (an inf / nan operand): the split yields NaN where an fp32 multiply-add chain yields inf, so both models switch to the fp32
    MFMA kernels for the rest of the run.  EXPOSURE: the flag word is read one step behind and after optimizer.step(), so the update of the
    step that tripped the guard -- and of the one after it -- is NOT redone (a non-finite operand makes the losses non-finite in either
    arithmetic); the warning says so.

    TINY (a non-zero operand element below 2^-110): informational by default.  Such an element keeps at least its leading bf16 term, so
    its products carry an ABSOLUTE error of at most 2^-9 |x| |w| < 2^-119 |w| -- invisible in any sum whose other terms are not themselves
    that small; only a reduction made ENTIRELY of sub-2^-110 operands comes out with bf16-like relative accuracy, on a result around
    1e-33 of the operand scale (tests/test_gpu_x6_admission.py quantifies both).  Real training produces such elements routinely -- the
    input gradients of a zero-padded image of a ragged batch reach 1e-36 next to 1e-5 (tools/x6_flag_hunt.py) -- so leaving the fast
    arithmetic for them would silently cost 25 % of the throughput for nothing.  It is logged once; ABR_X6_STRICT=1 switches on it too.

    Under data parallelism the flag is MAX-reduced over the ranks first, so that every rank switches at the same step."""
    if getattr(model_target, "conv_math", "f32") not in ("bf16x6", "f16x3"):
        return
    from .. import ops
    st = trainer_state(model_target)
    if st.x6_watch is None:
        st.x6_watch = ops.X6RangeWatch()
    # every rank polls at the same point of every step, so the reduced word is read at the same step everywhere
    multi = get_world_size() > 1 and dist.is_available() and dist.is_initialized()
    flags = st.x6_watch.poll(reduce_over_ranks=multi)
    logger = log or logging.getLogger("abr_iod_amd.trainer")
    if model_target.conv_math == "f16x3" and H3_STATS_EVERY > 0:
        # f16x3's domain: operand elements more than 18 binades below their tensor's amax keep an absolute accuracy of 2^-40 amax only.  They are
        # ordinary in small numbers (a fraction of 1e-3 in this step's tensors); a batch in which they are a large share of the operands is
        # outside what tests/test_gpu_f16x3_admission.py admits at the fp32 bound, and the models move to bf16x6 (exact split, any range)
        st.h3_polls = getattr(st, "h3_polls", 0) + 1
        if st.h3_polls % H3_STATS_EVERY == 0:
            got = st.x6_watch.poll_h3_stats(reduce_over_ranks=multi)
            if got is not None and got[1] > 0:
                st.h3_small_fraction = got[0] / got[1]
                if st.h3_small_fraction > H3_MAX_SMALL_FRACTION:
                    logger.warning("f16x3: {:.1%} of the operand elements of the last steps sit more than 18 binades below their tensor's largest "
                                   "magnitude (limit {:.1%}, ABR_H3_MAX_SMALL_FRACTION): switching both models to the bf16x6 arithmetic".format(
                                       st.h3_small_fraction, H3_MAX_SMALL_FRACTION))
                    for m in (model_source, model_target):
                        if m is not None and hasattr(m, "set_conv_math"):
                            m.set_conv_math("bf16x6")
                    st.drop_prefetch()
    if not flags:
        return
    if flags & ops.H3_FLAG_STALE:
        raise RuntimeError("f16x3: a kernel was handed an amax word that did not carry the epoch it was told (abr_iod_amd.ops amax tags): "
                           "the results of that launch are wrong -- a bug in the host plumbing, not in the data")
    if (flags & ops.X6_FLAG_TINY) and not (flags & ops.X6_FLAG_NONFINITE) and not X6_STRICT:
        if not st.x6_tiny_logged:
            st.x6_tiny_logged = True
            logger.info("bf16x6: operand elements below 2^-110 seen (e.g. input gradients of padded image regions); their products carry an "
                        "absolute error below 2^-119 x the other operand -- staying on the bf16 matrix cores (ABR_X6_STRICT=1 would switch)")
        return
    what = " + ".join(n for b, n in ((ops.X6_FLAG_TINY, "non-zero operand below 2^-110"), (ops.X6_FLAG_NONFINITE, "inf/nan operand")) if flags & b)
    logger.warning("bf16x6 / f16x3 range guard tripped ({}): switching both models to the fp32 MFMA kernels from the next step on; the last two "
                   "updates were computed with operands outside the exact-split domain and are not redone".format(what))
    for m in (model_source, model_target):
        if m is not None and hasattr(m, "set_conv_math"):
            m.set_conv_math("f32")
    st.x6_watch.reset()
    st.drop_prefetch()     # computed in the old arithmetic


def reduce_loss_dict(loss_dict):
    """engine/trainer.py:15-37: sum the loss scalars to rank 0 and average there (logging only)."""
    world_size = get_world_size()
    if world_size < 2:
        return loss_dict
    with torch.no_grad():
        names = sorted(loss_dict.keys())
        all_losses = torch.stack([loss_dict[k] for k in names], dim=0)
        dist.reduce(all_losses, dst=0)
        if dist.get_rank() == 0:
            all_losses /= world_size
        return {k: v for k, v in zip(names, all_losses)}


def _arm_overlap(optimizer, head_inputs, features):
    """Gradient hooks that hand finished buckets of the flat gradient to the optimiser's GradReducer while backward is still running:
    layer4 + predictor once every RoI pass's pooled input has its gradient, the RPN once the C4 feature map has its own."""
    reducer = getattr(optimizer, "reducer", None)
    final = getattr(optimizer, "bucket_final", None)    # FusedSGD: all-reduce of the bucket + (round 5) its early update
    if reducer is None or final is None:
        return
    from ..solver.build import EARLY_SGD
    if not (reducer.active or EARLY_SGD):
        return
    heads = [t for t in head_inputs if torch.is_tensor(t) and t.requires_grad]
    left = [len(heads)]

    def head_done(g):
        left[0] -= 1
        if left[0] == 0:
            final("roi_heads")
        return None

    for t in heads:
        t.register_hook(head_done)
    f = features[0] if isinstance(features, (list, tuple)) else features
    if heads and torch.is_tensor(f) and f.requires_grad:
        def features_done(g):
            final("roi_heads")   # (no-op when already sent)
            final("rpn")
            return None
        f.register_hook(features_done)


def _join_source_stream(deferred):
    """make the current stream (and the caching allocator) see what the source model's stream produced"""
    src = deferred.pop("_stream", None)
    if src is None:
        return
    cur = torch.cuda.current_stream()
    cur.wait_stream(src)
    for t in list(deferred["features"]) + [x for pair in deferred["rpn_output"] for x in pair]:
        if torch.is_tensor(t):
            t.record_stream(cur)


def train_step(model_source, model_target, images, targets, optimizer, scheduler, cfg, faithful_rng=False, log=None, next_images=None):
    """One iteration of tools/train_incremental.py:77-147.  Returns (loss_dict_target incl. 'distillation_loss', total loss)."""
    dist_type = cfg.DIST.TYPE
    use_id = cfg.DIST.ALPHA > 0
    use_ard = cfg.DIST.FEAT == "ard"
    need_source = use_id or use_ard or cfg.DIST.RPN or cfg.DIST.FEAT == "std"

    soften_result = soften_proposal = roi_align_features_source = rpn_output_source = roi_align_features_target = None
    deferred = None
    second_done = False
    target_prefix = None      # (event, frozen_prefix result) of the target's frozen stem / layer1 for THIS batch, computed during the previous step
    if need_source:
        with torch.no_grad():                                                                              # :82-86
            on_gpu = (images.tensors if hasattr(images, "tensors") else images).is_cuda
            if SOURCE_OVERLAP and not faithful_rng and on_gpu and hasattr(model_source, "soften_begin") and not model_source.training:
                # source backbone + RPN head now, its proposal selection on a side stream; finished after the target's forward
                pre = None
                tstate = trainer_state(model_target)
                pf = tstate.prefetched
                if pf and pf.get("images") is images and pf.get("key") == _prefetch_key(images, model_source, model_target):
                    pre = pf["state"]                 # enqueued during the previous step's backward pass
                    target_prefix = pf.get("target_prefix")
                tstate.drop_prefetch()
                if pre is not None:
                    deferred = pre
                elif SOURCE_STREAM:
                    from .. import ops
                    cur = torch.cuda.current_stream()
                    src = ops.side_stream((cur.device.index, "source-model"))
                    src.wait_stream(cur)          # the images, and everything of the previous step that read this stream's buffers
                    with torch.cuda.stream(src):
                        deferred = model_source.soften_begin(images)
                    deferred["_stream"] = src
                else:
                    deferred = model_source.soften_begin(images)
                rpn_output_source = deferred["rpn_output"]
            else:
                soften_result, _, soften_proposal, feature_source, _, _, rpn_output_source, roi_align_features_source = \
                    model_source.generate_soften_proposal(images)
                if faithful_rng:
                    model_source.roi_heads.box.loss_evaluator.subsample(soften_proposal, targets)

    prefetched_now = [False]

    def enqueue_prefetch():
        if prefetched_now[0]:
            return
        prefetched_now[0] = True
        if (PIPELINE_SOURCE and need_source and next_images is not None and SOURCE_STREAM and SOURCE_OVERLAP and not faithful_rng
                and hasattr(model_source, "soften_begin") and not model_source.training
                and (next_images.tensors if hasattr(next_images, "tensors") else next_images).is_cuda):
            # software pipelining: the frozen source model's backbone + RPN head for the NEXT batch go onto the source stream now (their result
            # does not depend on this step's update): with EARLY_PREFETCH right behind the current batch's source head pass, where they fill the
            # proposal selection's wait and run next to the RoI heads; otherwise just before the backward pass
            from .. import ops
            cur = torch.cuda.current_stream()
            src = ops.side_stream((cur.device.index, "source-model"))
            src.wait_stream(cur)       # next_images may have been produced on the current stream (async upload, device-side padding / augmentation)
            tstate = trainer_state(model_target)
            shared = None
            if (SHARE_FROZEN_PREFIX[0] and PIPELINE_TARGET_FROZEN and hasattr(model_target, "prefetch_frozen")
                    and frozen_prefix_shareable(model_source, model_target)):
                with torch.no_grad(), torch.cuda.stream(src):
                    shared = model_target.prefetch_frozen(next_images)     # ONE stem + layer1 pass for both models
                    if shared is not None:
                        ev_shared = torch.cuda.Event()
                        ev_shared.record()
            with torch.no_grad(), torch.cuda.stream(src):
                nxt = model_source.soften_begin(next_images, prefix=shared) if shared is not None else model_source.soften_begin(next_images)
            nxt["_stream"] = src
            # (holds the batch object, so its id() cannot be recycled while the entry lives)
            tstate.prefetched = dict(images=next_images, key=_prefetch_key(next_images, model_source, model_target), state=nxt)
            if shared is not None:
                tstate.prefetched["target_prefix"] = (ev_shared, shared)
            elif PIPELINE_TARGET_FROZEN and hasattr(model_target, "prefetch_frozen"):
                # the TARGET's frozen stem + layer1 (FREEZE_CONV_BODY_AT = 2) for the next batch too: their output does not depend on this step's
                # update either, and these bandwidth-bound convolutions overlap better with the backward pass's GEMMs than with the target's own
                # layer2 / layer3 in the next forward.  Same stream as the source model's prefetch: one bandwidth-bound chain at a time.
                with torch.no_grad(), torch.cuda.stream(src):
                    pf = model_target.prefetch_frozen(next_images)
                    if pf is not None:
                        ev = torch.cuda.Event()
                        ev.record()
                        tstate.prefetched["target_prefix"] = (ev, pf)

    joint = need_source and JOINT_ROI_PASS and deferred is None and hasattr(model_target, "forward_joint")
    if joint:   # :89-95 as one pass: the distillation RoIs share the detection pass's trip through layer4
        (loss_dict_target, feature_target, _, _, rpn_output_target, target_proposals, det_pooled, target_soften_results), \
            (target_result, _, roi_align_features_target) = model_target.forward_joint(images, targets, soften_proposal,
                                                                                       rpn_output_source=rpn_output_source)
    elif deferred is not None and hasattr(model_target, "forward_begin"):
        # the target's backbone / RPN head / RPN loss are queued and its proposal selection is in flight on a side stream; the source
        # model's selection finished long ago, so its head pass (a few ms of small GEMMs) goes in NOW: the device has work while
        # the target's top-k / NMS run and while the host waits for their counts
        prefix = None
        if target_prefix is not None:
            ev, prefix = target_prefix
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)
            for t in [prefix[0]] + list(prefix[1]):
                t.record_stream(cur)
        _ops.mark("step: target forward_begin issued from here")
        begun = model_target.forward_begin(images, targets, rpn_output_source=rpn_output_source, prefix=prefix)   # :89-90 (first half)
        _ops.mark("target backbone + RPN head + RPN loss done")
        src = deferred.get("_stream") if SOURCE_HEAD_STREAM else None
        if src is not None:       # the source's head pass stays on its stream, next to the target's big RoI pass below
            deferred.pop("_stream")
            with torch.no_grad(), torch.cuda.stream(src):
                soften_result, _, soften_proposal, feature_source, _, _, rpn_output_source, roi_align_features_source = \
                    model_source.soften_finish(deferred)
        else:
            _join_source_stream(deferred)
            with torch.no_grad():
                soften_result, _, soften_proposal, feature_source, _, _, rpn_output_source, roi_align_features_source = \
                    model_source.soften_finish(deferred)
        deferred = None
        src_done = None
        if src is not None:
            src_done = torch.cuda.Event()
            src_done.record(src)      # the source's results for THIS batch are complete here, whatever is queued on its stream afterwards
        if EARLY_PREFETCH and src is not None:
            # the next batch's source-model prefetch goes onto the source stream behind the head pass just queued there: it has nothing to
            # wait for and the device has the proposal selection's wait to fill
            enqueue_prefetch()
        ready = getattr(soften_proposal[0], "_roi_ready", None) if (EARLY_SECOND_PASS and src is not None and soften_proposal) else None
        joint_ov = JOINT_ROI_PASS and need_source and bool(soften_proposal) and hasattr(model_target.roi_heads, "forward_joint")
        if joint_ov:
            # ABR_JOINT_ROI in the overlapped step: the 64 distillation RoIs per image ride along with the 512 detection RoIs through ONE
            # layer4 / predictor pass (rows of the same GEMMs) instead of a second pass of M = 4096-row launches at ~half the big pass's rate
            cur = torch.cuda.current_stream()
            if src_done is not None:
                cur.wait_event(src_done)
            tab = getattr(soften_proposal[0], "_roi_table", None)
            if tab is not None:
                tab[0].record_stream(cur)
            (loss_dict_target, feature_target, _, _, rpn_output_target, target_proposals, det_pooled, target_soften_results), \
                (target_result, _, roi_align_features_target) = model_target.forward_finish(begun, soften_proposals=soften_proposal)
            second_done = True
            ready = None
        if ready is not None:
            # :93-95 ahead of the second half of :89-90.  The target's pass over the distillation RoIs needs its backbone features and the
            # SOURCE's proposals -- not the target's own proposals, whose selection (top-k, NMS) the main stream would otherwise sit and
            # wait for: ~1 ms of layer4 work over 256 RoIs fills that wait.  Same values; the two passes only swap places in the queue.
            cur = torch.cuda.current_stream()
            cur.wait_event(ready)
            tab = getattr(soften_proposal[0], "_roi_table", None)
            if tab is not None:
                tab[0].record_stream(cur)
            target_result, _, roi_align_features_target = model_target.forward(images, targets, features=begun["features"],
                                                                               proposals=soften_proposal)  # :93-95
            second_done = True
        if not joint_ov:
            loss_dict_target, feature_target, _, _, rpn_output_target, target_proposals, det_pooled, target_soften_results = \
                model_target.forward_finish(begun)                                                         # :89-90 (second half)
        if src is not None:       # everything the source stream produced becomes visible to the main stream here
            cur = torch.cuda.current_stream()
            cur.wait_event(src_done)
            outs = [soften_result[0], soften_result[1], roi_align_features_source] + list(feature_source) + [p.bbox for p in soften_proposal] + \
                   [x for pair in rpn_output_source for x in pair]
            tab = getattr(soften_proposal[0], "_roi_table", None) if soften_proposal else None
            if tab is not None:
                outs.append(tab[0])
            for t in outs:
                if torch.is_tensor(t):
                    t.record_stream(cur)
    else:
        loss_dict_target, feature_target, _, _, rpn_output_target, target_proposals, det_pooled, target_soften_results = \
            model_target(images, targets, rpn_output_source=rpn_output_source)                             # :89-90
    if deferred is not None:
        _join_source_stream(deferred)
        with torch.no_grad():
            soften_result, _, soften_proposal, feature_source, _, _, rpn_output_source, roi_align_features_source = \
                model_source.soften_finish(deferred)

    # the loss arithmetic of train_incremental.py:91,101-128 -- sum of the detector losses, alpha * ID (+ std) + beta * ARD (+ RPN), their
    # sum -- gathered as (term, weight, group) and evaluated by ONE kernel (ops.loss_sum) instead of a chain of scalar adds / muls
    terms = [(v, 1.0, 0) for v in loss_dict_target.values()]                                               # :91
    if need_source:
        if not joint and not second_done:
            target_result, _, roi_align_features_target = model_target.forward(images, targets, features=feature_target,
                                                                               proposals=soften_proposal)  # :93-95
        if use_id:                                                                                         # :101-103
            terms.append((calculate_roi_distillation_losses(soften_result, target_result, dist=dist_type, soften_proposal=None),
                          cfg.DIST.ALPHA, 1))
        if cfg.DIST.FEAT == "std":                                                                         # :108-112 (ablation)
            terms.append((calculate_feature_distillation_loss(feature_source, feature_target, loss="normalized_filtered_l1"), 1.0, 1))
        elif use_ard:                                                                                      # :113-116
            terms.append((calculate_attentive_roi_feature_distillation(roi_align_features_source, roi_align_features_target,
                                                                       gamma=cfg.DIST.GAMMA), cfg.DIST.BETA, 1))
        if cfg.DIST.RPN:                                                                                   # :120-122 (ablation)
            terms.append((calculate_rpn_distillation_loss(rpn_output_source, rpn_output_target, cls_loss="filtered_l2", bbox_loss="l2",
                                                          bbox_threshold=0.1), 1.0, 1))
    loss_dict_target = dict(loss_dict_target)
    if terms[0][0].is_cuda and len(terms) <= 8:
        from .. import ops
        losses, parts = ops.loss_sum([t for t, _, _ in terms], [w for _, w, _ in terms], [g for _, _, g in terms])
        loss_dict_target["distillation_loss"] = parts[2]                                                   # :124-126
    else:
        faster_rcnn_losses = sum(t for t, _, g in terms if g == 0)
        distillation_losses = sum((w * t for t, w, g in terms if g == 1), torch.zeros((), device=faster_rcnn_losses.device))
        loss_dict_target["distillation_loss"] = distillation_losses.clone().detach()
        losses = faster_rcnn_losses + distillation_losses                                                  # :128

    enqueue_prefetch()
    if terms[0][0].is_cuda:
        _ops.mark("losses done (backward starts)")
    optimizer.zero_grad()                                                                                  # :142
    # tensors whose gradients together say "every kernel of layer4's + the predictor's backward is queued": the pooled inputs of the RoI passes
    # (the joint pass has ONE pooled input for both RoI sets; its distillation-RoI output gets ARD's gradient much earlier and must not count)
    head_inputs = [det_pooled] if getattr(det_pooled, "_abr_joint_pool", False) else [det_pooled, roi_align_features_target if need_source else None]
    _arm_overlap(optimizer, head_inputs, feature_target)
    losses.backward()                                                                                      # :144-145 (amp O0 = identity)
    if terms[0][0].is_cuda:
        _ops.mark("backward done on the main stream")
    optimizer.step()                                                                                       # :146 (+ RCCL all-reduce)
    if terms[0][0].is_cuda:
        _ops.mark("optimizer.step done (SGD kernel + joins)")
    scheduler.step()                                                                                       # :147
    _x6_guard(model_source, model_target, log)
    return loss_dict_target, losses


def do_train(model_source, model_target, data_loader, optimizer, scheduler, checkpointer_target, device, checkpoint_period,
             arguments_target, summary_writer, cfg, faithful_rng=False):
    """Same signature as tools/train_incremental.py:55-56.  data_loader yields (images, targets, _, idx).
    `faithful_rng=True` also runs the reference's dead `subsample` call on the source model (:86), which consumes device RNG."""
    logger = logging.getLogger("abr_iod_amd.trainer")
    logger.info("Start training")
    max_iter = len(data_loader)
    start_iter = arguments_target["iteration"]
    model_target.train()
    model_source.eval()
    start_training_time = time.time()
    end = time.time()
    def on_device(batch):
        images, targets, _, idx = batch
        return images.to(device), [t.to(device) for t in targets], idx

    it = iter(data_loader)
    nxt = next(it, None)
    nxt = on_device(nxt) if nxt is not None else None
    iteration = start_iter
    while nxt is not None:
        data_time = time.time() - end
        images, targets, idx = nxt
        nxt = next(it, None)                      # one batch of lookahead: the source model's forward for it is pipelined into this step
        nxt = on_device(nxt) if nxt is not None else None
        iteration = iteration + 1
        arguments_target["iteration"] = iteration
        loss_dict_target, losses = train_step(model_source, model_target, images, targets, optimizer, scheduler, cfg,
                                              faithful_rng=faithful_rng, next_images=nxt[0] if nxt is not None else None)
        loss_dict_reduced = reduce_loss_dict(loss_dict_target)
        batch_time = time.time() - end
        end = time.time()
        if iteration % 100 == 0 or iteration == max_iter:
            losses_reduced = sum(loss for loss in loss_dict_reduced.values())
            eta = str(datetime.timedelta(seconds=int(batch_time * (max_iter - iteration))))
            logger.info("eta: {}  iter: {}  loss: {:.4f}  {}  time: {:.4f}  data: {:.4f}  lr: {:.6f}".format(
                eta, iteration, float(losses_reduced), "  ".join("{}: {:.4f}".format(k, float(v)) for k, v in loss_dict_reduced.items()),
                batch_time, data_time, optimizer.param_groups[0]["lr"]))
            if summary_writer is not None:
                summary_writer.add_scalar("train_loss_raw", float(losses_reduced), iteration)
        if checkpointer_target is not None:
            if iteration % checkpoint_period == 0:
                checkpointer_target.save("model_last", **arguments_target)
            if iteration == max_iter:
                checkpointer_target.save("model_final", **arguments_target)
    total = time.time() - start_training_time
    logger.info("Total training time: {} ({:.4f} s / it)".format(str(datetime.timedelta(seconds=total)), total / max(max_iter, 1)))
