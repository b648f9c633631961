"""Flat parameter / gradient storage.

The reference keeps 52 separate trainable tensors, gives each its own SGD param group (solver/build.py:7-21) and lets
DistributedDataParallel bucket their gradients.  On MI355X the whole trainable set (32.96 M fp32 = 131.9 MB) is ONE
contiguous buffer, its gradient ONE buffer of the same shape:
  * wgrad kernels accumulate straight into views of the gradient buffer (no per-tensor autograd accumulation),
  * the data-parallel exchange is a single RCCL all-reduce over that buffer (or a few large chunks),
  * the optimiser is one fused multi-tensor kernel with per-segment (lr, weight-decay).
Every nn.Parameter keeps its reference name and shape: `p.data` / `p.grad` are views into the flat buffers.
Modules that fuse tensors (RPN cls+bbox head, predictor cls+bbox FC) declare `flat_groups()` so their pieces stay adjacent.
"""
import torch

ALIGN = 64  # floats (256 B)


class FlatParams(object):
    def __init__(self):
        self.params = None      # all parameters, trainable region first
        self.grads = None       # trainable region only
        self.n_trainable = 0    # elements in the trainable region (incl. alignment padding)
        self.segments = []      # (name, start, end, is_bias) per optimiser tensor, ascending, trainable region

    def zero_grad(self):
        self.grads.zero_()


def _round_up(n):
    return (n + ALIGN - 1) // ALIGN * ALIGN


def flatten_parameters(model):
    names = {id(p): n for n, p in model.named_parameters()}
    units, grouped = [], set()
    for m in model.modules():
        if hasattr(m, "flat_groups"):
            for which, plist, pad_rows in m.flat_groups():
                row = plist[0].numel() // plist[0].shape[0]
                units.append(dict(module=m, which=which, params=plist, numel=sum(p.numel() for p in plist) + pad_rows * row,
                                  trainable=any(p.requires_grad for p in plist), name=names[id(plist[0])]))
                grouped.update(id(p) for p in plist)
    for n, p in model.named_parameters():
        if id(p) not in grouped:
            units.append(dict(module=None, which=None, params=[p], numel=p.numel(), trainable=p.requires_grad, name=n))
    units.sort(key=lambda u: not u["trainable"])  # stable: trainable first, original order otherwise

    dev = next(model.parameters()).device
    off, n_train = 0, 0
    for u in units:
        u["off"] = off
        off = _round_up(off + u["numel"])
        if u["trainable"]:
            n_train = off
    flat = FlatParams()
    flat.params = torch.zeros(off, dtype=torch.float32, device=dev)
    flat.grads = torch.zeros(n_train, dtype=torch.float32, device=dev)
    flat.n_trainable = n_train

    with torch.no_grad():
        for u in units:
            a, b = u["off"], u["off"] + u["numel"]
            view = flat.params[a:b]
            gview = flat.grads[a:b] if u["trainable"] else None
            if u["module"] is None:
                p = u["params"][0]
                view.copy_(p.data.reshape(-1))
                p.data = view.view(p.shape)
                if u["trainable"]:
                    p.grad = gview.view(p.shape)
            else:
                o = 0
                for p in u["params"]:
                    view[o:o + p.numel()].copy_(p.data.reshape(-1))
                    o += p.numel()
                u["module"].rehome(u["which"], view, gview if gview is not None else torch.zeros_like(view))
            if u["trainable"]:
                is_bias = "bias" in u["name"]  # solver/build.py:14 `if "bias" in key`
                flat.segments.append((u["name"], a, _round_up(b), is_bias))
    return flat
