"""Checkpoint boundary: convert between this package's storage (OHWI conv weights, padded stem, fused heads in flat
buffers) and the reference's state_dict layout (OIHW, names of maskrcnn_benchmark/utils/checkpoint.py:32-74).
Keys and shapes of `reference_state_dict(model)` equal those of the reference's `model.state_dict()` for the same cfg."""
import torch

from ..modeling.backbone.resnet import Conv2d, bump_param_version


def reference_state_dict(model):
    out = {}
    convs = {id(m.weight): m for m in model.modules() if isinstance(m, Conv2d)}
    for name, p in model.named_parameters():
        m = convs.get(id(p))
        out[name] = m.oihw().clone() if m is not None else p.detach().clone()
    for name, b in model.named_buffers():
        if name.endswith("cell_anchors"):
            continue
        out[name] = b.detach().clone()
    return out


def load_reference_state_dict(model, sd, strict=True):
    """Inverse of reference_state_dict.  Grown heads (more classes in the model than in `sd`) get the stored rows copied
    into their first rows (utils/model_serialization.py:47-55)."""
    convs = {id(m.weight): m for m in model.modules() if isinstance(m, Conv2d)}
    missing = []
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name not in sd:
                missing.append(name)
                continue
            v = sd[name].to(p.device)
            m = convs.get(id(p))
            if m is not None:
                if v.shape[0] == p.shape[0]:
                    m.load_oihw(v)
                else:
                    p[: v.shape[0]].copy_(v.permute(0, 2, 3, 1))
            elif v.shape == p.shape:
                p.copy_(v)
            else:
                p[: v.shape[0]].copy_(v)
        for name, b in model.named_buffers():
            if name in sd and sd[name].shape == b.shape:
                b.copy_(sd[name].to(b.device))
    from ..layers import FrozenBatchNorm2d
    for m in model.modules():
        if isinstance(m, FrozenBatchNorm2d):
            m.invalidate()
    bump_param_version()
    if strict and missing:
        raise KeyError("missing keys: {}".format(missing))
    return missing
