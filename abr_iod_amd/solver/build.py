"""make_optimizer / make_lr_scheduler (mirror of maskrcnn_benchmark/solver/build.py:7-30).

The reference builds one torch.optim.SGD param group PER TENSOR (weights: lr=BASE_LR, wd=WEIGHT_DECAY; any name containing
"bias": lr*BIAS_LR_FACTOR, wd=WEIGHT_DECAY_BIAS) -> 52 x 3 tiny launches per step.  FusedSGD keeps exactly those per-tensor
hyper-parameters but applies the whole update as ONE kernel over the model's flat parameter buffer, and owns the
data-parallel gradient exchange (a few large RCCL all-reduces of the flat gradient buffer over xGMI, the first ones issued during
the backward pass: solver/grad_reducer.py)."""
import os

import torch
import torch.distributed as dist

from .. import ops
from ..modeling.backbone.resnet import Conv2d, bump_trained_version
from .grad_reducer import BUCKET_ORDER, GradReducer, bucket_of
from .lr_scheduler import WarmupMultiStepLR

# Round 5: the update of a gradient bucket (solver/grad_reducer.py: roi_heads, rpn, backbone) is applied AS SOON AS the bucket is final --
# from the gradient hooks of engine/trainer.py::_arm_overlap, on a stream of its own behind the weight-gradient streams (and behind the bucket's
# all-reduce under data parallelism) -- together with the preparation of the data derived from those weights for the NEXT step.  layer4 + predictor
# and the RPN (75 % of the parameters, the largest derived data) are then updated and re-packed underneath the backbone's backward pass;
# optimizer.step() is left with the backbone bucket.  Same kernels on the same values: the results do not change.
# OPT-IN (ABR_EARLY_SGD=1): measured SLOWER in the step -- B = 4: 18.91 vs 18.24 ms, B = 2: 11.53 vs 10.89 ms (same session, two rounds each): the
# hook issues its ~10 launches from the autograd thread in the middle of the backward pass (the dgrad chain's issue stalls behind them), and the
# re-packing kernels then share the CUs with the backbone's backward instead of with the next step's prefetched forward.
EARLY_SGD = os.environ.get("ABR_EARLY_SGD", "0") != "0"


class FusedSGD(object):
    """torch.optim.SGD(momentum) semantics (dampening 0, no nesterov) on FlatParams.  `param_groups` mirrors the reference's
    per-tensor groups so schedulers / loggers that read or scale group['lr'] keep working."""

    def __init__(self, model, base_lr, momentum, weight_decay, bias_lr_factor, weight_decay_bias):
        if getattr(model, "flat", None) is None:
            model.flatten_parameters()
        self.model = model
        self.flat = model.flat
        for m in model.modules():   # these convs' weights change with every step() (modeling/backbone/resnet.py: weight versions)
            if isinstance(m, Conv2d) and m.weight.requires_grad:
                m._optimised = True
        # modules that keep data derived from trainable weights (flipped dgrad copies, Winograd-domain weights): rebuilt right after
        # the update on a stream of their own, off the next step's critical path (ABR_WEIGHT_PREP_STREAM=0: lazily, at first use)
        self._derived = [m for m in model.modules() if hasattr(m, "prepare_derived") and any(p.requires_grad for p in m.parameters())]
        self._prep_stream = os.environ.get("ABR_WEIGHT_PREP_STREAM", "1") != "0"
        self._batch_prep = os.environ.get("ABR_BATCH_WEIGHT_PREP", "1") != "0"
        # stage of every module with derived data (backbone.body.layer2 -> "layer2", rpn.head -> "rpn", ...): one preparation batch per stage
        self._derived_group = {}
        for name, m in model.named_modules():
            parts = name.split(".")
            self._derived_group[id(m)] = next((p_ for p_ in parts if p_.startswith("layer")), parts[0] if parts else "")
        self.momentum = momentum
        self.param_groups = []
        for name, a, b, is_bias in self.flat.segments:
            lr = base_lr * bias_lr_factor if is_bias else base_lr
            wd = weight_decay_bias if is_bias else weight_decay
            self.param_groups.append({"name": name, "lr": lr, "initial_lr": lr, "weight_decay": wd, "range": (a, b)})
        dev = self.flat.params.device
        self.momentum_buffer = torch.zeros_like(self.flat.grads)
        self._seg_end = torch.tensor([g["range"][1] for g in self.param_groups], dtype=torch.int64, device=dev)
        self._wd = torch.tensor([g["weight_decay"] for g in self.param_groups], dtype=torch.float32, device=dev)
        self._lr = torch.empty(len(self.param_groups), dtype=torch.float32, device=dev)
        self._lr_host = None
        self._steps = 0
        self._prep_plan = None
        self.reducer = GradReducer(self.flat.grads, self.flat.segments, side_streams=self._grad_writer_streams)
        # per bucket: [(a, b, first segment, one past the last segment, segment ends relative to a)] -- the SGD kernel runs on sub-ranges
        self._bucket_tables = {}
        ends = [g["range"][1] for g in self.param_groups]
        for name, ranges in self.reducer.buckets.items():
            tabs = []
            for a, b in ranges:
                idx = [i for i, g in enumerate(self.param_groups) if a <= g["range"][0] and g["range"][1] <= b]
                if idx:
                    tabs.append((a, b, idx[0], idx[-1] + 1, torch.tensor([ends[i] - a for i in idx], dtype=torch.int64, device=dev)))
            self._bucket_tables[name] = tabs
        self._early_done = set()
        self._early_stream_used = False

    @property
    def world_size(self):
        """read when used (not snapshotted at construction): an optimizer built before init_process_group still averages over the ranks"""
        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    @property
    def force_all_reduce(self):  # single-rank RCCL run in tests/test_gpu_dist.py
        return self.reducer.force

    @force_all_reduce.setter
    def force_all_reduce(self, on):
        self.reducer.force = bool(on)

    def _grad_writer_streams(self):
        if not self.flat.grads.is_cuda:
            return []
        # the streams that hold kernels writing the gradient buffer: the weight-gradient streams (ops._wgrad_stream: key = device index, or
        # (device, "wgradN")) and the stream the step was started on.  NOT the frozen source model's stream, the proposal-selection streams
        # or the weight-preparation stream: they never touch gradients, and the source stream carries the next batch's whole prefetch --
        # a bucket's all-reduce waiting for that would start milliseconds late.
        dev = self.flat.grads.device.index
        out = [s for k, s in ops._side_streams.items()
               if (k == dev) or (isinstance(k, tuple) and k[0] == dev and str(k[1]).startswith("wgrad"))]
        if getattr(self, "_main_stream", None) is not None:
            out.append(self._main_stream)
        return out

    def zero_grad(self, set_to_none=False):
        self._main_stream = torch.cuda.current_stream() if self.flat.grads.is_cuda else None
        self.flat.zero_grad()
        self.reducer.begin()
        self._early_done = set()
        self._upload_lr()    # (the early bucket updates read it during backward; scheduler.step() runs after optimizer.step())

    def _upload_lr(self):
        lrs = [g["lr"] for g in self.param_groups]
        if lrs != self._lr_host:  # only re-upload when the scheduler changed something
            src = torch.tensor(lrs, dtype=torch.float32)
            self._lr.copy_(src.pin_memory() if self._lr.is_cuda else src, non_blocking=True)  # pageable source = host stall
            self._lr_host = lrs

    def _sgd_ranges(self, buckets):
        for name in buckets:
            for a, b, i0, i1, seg_end in self._bucket_tables[name]:
                ops.sgd_momentum_(self.flat.params[a:b], self.flat.grads[a:b], self.momentum_buffer[a:b], seg_end, self._lr[i0:i1], self._wd[i0:i1],
                                  self.momentum, gscale=1.0 / self.world_size, first_step=(self._steps == 0))

    def bucket_final(self, name):
        """Every kernel that writes gradient bucket `name` (and every earlier bucket of BUCKET_ORDER) has been ENQUEUED: send its all-reduce
        (data parallelism) and, with EARLY_SGD, apply its update and prepare its derived data for the next step right now, on the early stream."""
        self.reducer.reduce_bucket_async(name)
        if not (EARLY_SGD and self.flat.grads.is_cuda and self._prep_stream and self._batch_prep and name != BUCKET_ORDER[-1]):
            return
        todo = [b for b in BUCKET_ORDER[: BUCKET_ORDER.index(name) + 1] if b not in self._early_done]
        if not todo:
            return
        if self.reducer.active and not all(b in self.reducer._done for b in todo):
            return   # (overlap of the exchange is off: the buckets go out from step(), and so do the updates)
        from ..modeling.backbone.resnet import _PARAM_VERSION
        plan = self._prep_plan
        if plan is None or plan["signature"] != self._prep_signature():
            plan = self._prep_plan = self._build_prep_plan()
        cur = torch.cuda.current_stream()
        early = ops.side_stream((self.flat.params.device.index, "early-sgd"))
        early.wait_stream(cur)
        for s in self._grad_writer_streams():
            early.wait_stream(s)
        with torch.cuda.stream(early), torch.no_grad():
            if self.reducer.active:
                for w in list(self.reducer._works):
                    if w is not None:      # (None = a bucket-end marker)
                        w.wait()           # the early stream waits for the buckets' all-reduces
            self._sgd_ranges(todo)
            for b in todo:
                sub = plan["buckets"].get(b)
                if sub is None:
                    continue
                for pb in sub["forward"]:
                    pb.run()
                sub["backward"].run()
                for conv in sub["convs"]:
                    conv._wt_version = _PARAM_VERSION[0] + 1   # (the version optimizer.step() is about to give the weights)
        self._early_done.update(todo)
        self._early_stream_used = True

    def all_reduce_grads(self):
        """DistributedDataParallel's job in the reference (train_incremental.py:231): sum the flat gradient over ranks -- the
        buckets the trainer's gradient hooks did not already send during backward, then wait for all of them."""
        self.reducer.finish()

    def step(self):
        if self.flat is not self.model.flat:
            raise RuntimeError("the model's flat parameter storage was rebuilt (model.to(device) after the optimizer was made): "
                               "build the optimizer after moving the model")
        ops.join_side_stream()  # weight gradients queued on the side stream (no-op when already joined after backward)
        self.all_reduce_grads()
        self._upload_lr()
        early_done = set(self._early_done)
        if early_done:
            # the buckets updated from the gradient hooks: only the rest is left (the early stream is joined below, behind this kernel)
            self._sgd_ranges([b for b in BUCKET_ORDER if b not in early_done])
        else:
            n = self.flat.n_trainable
            ops.sgd_momentum_(self.flat.params[:n], self.flat.grads, self.momentum_buffer, self._seg_end, self._lr, self._wd,
                              self.momentum, gscale=1.0 / self.world_size, first_step=(self._steps == 0))
        if self._early_stream_used and self.flat.params.is_cuda:
            torch.cuda.current_stream().wait_stream(ops.side_stream((self.flat.params.device.index, "early-sgd")))
        self._early_done = set()
        self._steps += 1
        bump_trained_version()  # data derived from the optimised weights (dgrad copies, Winograd-domain weights) is stale now
        if self._prep_stream and self.flat.params.is_cuda and self._derived:
            cur = torch.cuda.current_stream()
            prep = ops.side_stream((self.flat.params.device.index, "weight-prep"))
            prep.wait_stream(cur)            # behind the SGD kernel and every reader of the previous copies
            with torch.cuda.stream(prep), torch.no_grad():
                if self._batch_prep:
                    # every trainable conv's derived data in a dozen launches (ops.conv_prepare_batch) instead of ~190 (3.2 -> 1.0 ms of host time per step, and
                    # a host-paced train of tiny kernels on a hardware queue the source model's head pass shares).  In the order of first use,
                    # one batch per stage: the next forward pass waits for a stage's data only (the library orders consumers behind the fill of
                    # THEIR entry); the dgrad copies and what derives from them, first needed ~10 ms later, come last.  The tables are built
                    # once (ops.PreparedBatch: every pointer is fixed from step to step) and rebuilt when a module's inputs to them change.
                    from ..modeling.backbone.resnet import _PARAM_VERSION
                    plan = self._prep_plan
                    if plan is None or plan["signature"] != self._prep_signature():
                        plan = self._prep_plan = self._build_prep_plan()
                    if early_done:
                        for bname in BUCKET_ORDER:
                            if bname in early_done or bname not in plan["buckets"]:
                                continue
                            for b in plan["buckets"][bname]["forward"]:
                                b.run()
                        for fn in plan["rest"]:
                            fn()
                        for bname in BUCKET_ORDER:
                            if bname in early_done or bname not in plan["buckets"]:
                                continue
                            plan["buckets"][bname]["backward"].run()
                            for conv in plan["buckets"][bname]["convs"]:
                                conv._wt_version = _PARAM_VERSION[0]
                    else:
                        for b in plan["forward"]:    # forward halves, stage by stage
                            b.run()
                        for fn in plan["rest"]:
                            fn()
                        plan["backward"].run()
                        for conv in plan["convs"]:
                            conv._wt_version = _PARAM_VERSION[0]
                else:
                    for m in self._derived:
                        m.prepare_derived()
            ops.prep_done(prep)

    def _prep_signature(self):
        """what the cached preparation tables were built from: the modules' entries (conv, FrozenBN scale tensor, geometry, math mode), the
        weights' and dgrad buffers' addresses"""
        sig = []
        for m in self._derived:
            if hasattr(m, "prep_entries"):
                for conv, scale, stride, pad, math in m.prep_entries():
                    sig.append((id(conv), conv.weight.data_ptr(), id(scale), conv.dgrad_buffer().data_ptr(), stride, pad, math))
        return tuple(sig)

    def _build_prep_plan(self):
        groups, rest = [], []
        names = {id(m): n for n, m in self.model.named_modules()}
        per_bucket = {}
        for m in self._derived:
            if hasattr(m, "prep_entries"):
                ent = list(m.prep_entries())
                per_bucket.setdefault(bucket_of(names.get(id(m), "") + "."), []).extend(ent)
                key = self._derived_group.get(id(m), "")
                if not groups or groups[-1][0] != key:
                    groups.append((key, []))
                groups[-1][1].extend(ent)
                if hasattr(m, "prepare_rest"):
                    rest.append(m.prepare_rest)
            else:
                rest.append(m.prepare_derived)
        allent = [e for _, ent in groups for e in ent]
        forward = [ops.PreparedBatch([(conv.weight.detach(), None, None, stride, pad, math, conv.version) for conv, _, stride, pad, math in ent])
                   for _, ent in groups]
        backward = ops.PreparedBatch([(conv.weight.detach(), scale, conv.dgrad_buffer(), stride, pad, math, conv.version)
                                      for conv, scale, stride, pad, math in allent])
        # the same tables cut by gradient bucket, for the early updates: `nxt` = the version the weights get at the END of the current step
        buckets = {}
        for bname, ent in per_bucket.items():
            early = bname != BUCKET_ORDER[-1]
            ver = (lambda c: (lambda: c.version() + 2)) if early else (lambda c: c.version)
            buckets[bname] = dict(
                forward=[ops.PreparedBatch([(conv.weight.detach(), None, None, stride, pad, math, ver(conv)) for conv, _, stride, pad, math in ent])],
                backward=ops.PreparedBatch([(conv.weight.detach(), scale, conv.dgrad_buffer(), stride, pad, math, ver(conv)) for conv, scale, stride, pad, math in ent]),
                convs=[e[0] for e in ent])
        return dict(signature=self._prep_signature(), forward=forward, rest=rest, backward=backward, convs=[e[0] for e in allent], buckets=buckets)

    def _reference_params(self):
        """(name, parameter, offset into the flat buffer, Conv2d module or None) for every trainable tensor, in the
        reference optimiser's order = named_parameters() order of the requires_grad tensors (solver/build.py:9-18)."""
        from ..modeling.backbone.resnet import Conv2d
        convs = {id(m.weight): m for m in self.model.modules() if isinstance(m, Conv2d)}
        base = self.flat.params.data_ptr()
        for name, p in self.model.named_parameters():
            if p.requires_grad:
                yield name, p, (p.data_ptr() - base) // 4, convs.get(id(p))

    def _group_of(self, off):
        for g in self.param_groups:
            if g["range"][0] <= off < g["range"][1]:
                return g
        raise KeyError(off)

    def state_dict(self):
        """torch.optim.SGD.state_dict() layout -- what the reference's Checkpointer stores under "optimizer"
        (utils/checkpoint.py:41-43): one param group per tensor, momentum buffers in the reference's OIHW layout."""
        groups, state = [], {}
        for i, (name, p, off, conv) in enumerate(self._reference_params()):
            g = self._group_of(off)
            groups.append({"lr": g["lr"], "momentum": self.momentum, "dampening": 0, "weight_decay": g["weight_decay"],
                           "nesterov": False, "maximize": False, "foreach": None, "differentiable": False, "fused": None,
                           "initial_lr": g["initial_lr"], "params": [i]})
            if self._steps > 0:
                m = self.momentum_buffer[off:off + p.numel()].view(p.shape)
                if conv is not None:
                    m = m[..., : conv.in_channels].permute(0, 3, 1, 2)
                state[i] = {"momentum_buffer": m.contiguous().clone()}
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        ref = list(self._reference_params())
        if len(sd["param_groups"]) != len(ref):
            raise ValueError("loaded state dict has {} parameter groups, the optimizer has {}".format(len(sd["param_groups"]), len(ref)))
        self.momentum_buffer.zero_()
        any_state = False
        for i, ((name, p, off, conv), g_in) in enumerate(zip(ref, sd["param_groups"])):
            g = self._group_of(off)
            for k in ("lr", "weight_decay", "initial_lr"):
                if k in g_in:
                    g[k] = g_in[k]
            st = sd["state"].get(i, sd["state"].get(str(i)))
            if st is None or st.get("momentum_buffer") is None:
                continue
            v = st["momentum_buffer"].to(self.momentum_buffer.device)
            m = self.momentum_buffer[off:off + p.numel()].view(p.shape)
            if conv is not None:
                v = v.permute(0, 2, 3, 1)
                m = m[..., : conv.in_channels]
            if v.shape != m.shape:
                raise ValueError("momentum buffer of {} has shape {}, expected {}".format(name, tuple(v.shape), tuple(m.shape)))
            m.copy_(v)
            any_state = True
        self._steps = 1 if any_state else 0
        self._wd.copy_(torch.tensor([g["weight_decay"] for g in self.param_groups], dtype=torch.float32))
        self._lr_host = None


def make_optimizer(cfg, model):
    return FusedSGD(model, cfg.SOLVER.BASE_LR, cfg.SOLVER.MOMENTUM, cfg.SOLVER.WEIGHT_DECAY, cfg.SOLVER.BIAS_LR_FACTOR,
                    cfg.SOLVER.WEIGHT_DECAY_BIAS)


def make_lr_scheduler(cfg, optimizer):
    return WarmupMultiStepLR(optimizer, cfg.SOLVER.STEPS, cfg.SOLVER.GAMMA, warmup_factor=cfg.SOLVER.WARMUP_FACTOR,
                             warmup_iters=cfg.SOLVER.WARMUP_ITERS, warmup_method=cfg.SOLVER.WARMUP_METHOD)
