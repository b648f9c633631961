"""Checkpoint boundary: convert between this package's storage (OHWI conv weights, padded stem, fused heads in flat
buffers) and the reference's state_dict layout (OIHW, names of maskrcnn_benchmark/utils/checkpoint.py:32-74).
Keys and shapes of `reference_state_dict(model)` equal those of the reference's `model.state_dict()` for the same cfg
(the anchor generator's `cell_anchors.0` buffer included)."""
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
            # the reference registers its cell anchors through a BufferList (rpn/anchor_generator.py:13-31, 61): one buffer per level
            out[name.replace("cell_anchors", "cell_anchors.0")] = b.detach().clone()
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


# ------------------------------------------------------------------------------------------------------------------
# F3: checkpoint FILES in the reference's format (utils/checkpoint.py:13-142, utils/model_serialization.py:10-91)
# ------------------------------------------------------------------------------------------------------------------
import logging
import os
import pickle
import re


def strip_prefix_if_present(state_dict, prefix):
    """A state_dict saved from a DistributedDataParallel wrapper carries "module." on every key
    (utils/model_serialization.py:62-69)."""
    if not state_dict or not all(k.startswith(prefix) for k in state_dict):
        return state_dict
    return type(state_dict)((k.replace(prefix, ""), v) for k, v in state_dict.items())


def align_keys(model_keys, loaded_keys):
    """Suffix matching of utils/model_serialization.py:10-35: a model key takes the LONGEST loaded key that is a suffix of
    it (so `backbone.body.layer1.0.conv1.weight` takes `layer1.0.conv1.weight`, not `conv1.weight`).  Candidates are
    scanned in sorted order and ties keep the first, as the reference's argmax over the sorted list does.
    Returns {model_key: loaded_key}."""
    loaded_sorted = sorted(loaded_keys)
    out = {}
    for mk in sorted(model_keys):
        best, best_len = None, 0
        for lk in loaded_sorted:
            if len(lk) > best_len and mk.endswith(lk):
                best, best_len = lk, len(lk)
        if best is not None:
            out[mk] = best
    return out


def load_state_dict(model, loaded_state_dict):
    """Mirror of utils/model_serialization.py:72-91: strip "module.", suffix-align, copy (partial rows for grown heads);
    model keys without a match keep their values, unmatched loaded keys are ignored."""
    loaded = strip_prefix_if_present(loaded_state_dict, "module.")
    model_keys = [n for n, _ in model.named_parameters()] + [n for n, _ in model.named_buffers()]
    amap = align_keys(model_keys, loaded.keys())
    return load_reference_state_dict(model, {mk: loaded[lk] for mk, lk in amap.items()}, strict=False)


_C2_BLOCK = re.compile(r"^res(\d)_(\d+)_branch(1|2a|2b|2c)(_bn)?_(w|b|s)$")
_C2_HEADS = {"conv_rpn": "rpn.head.conv", "rpn_cls_logits": "rpn.head.cls_logits", "rpn_bbox_pred": "rpn.head.bbox_pred",
             "cls_score": "cls_score", "bbox_pred": "bbox_pred", "fc1000": "fc1000", "pred": "fc1000"}


def c2_blob_to_key(name):
    """Detectron/Caffe2 blob name -> state_dict key for the ResNet-C4 family (the only one this path uses; what
    utils/c2_model_loading.py:12-66 does with chained string replaces).  Returns None for blobs that are not weights."""
    if name.endswith("_momentum"):
        return None
    m = _C2_BLOCK.match(name)
    if m:
        stage, blk, branch, bn, kind = m.groups()
        layer = "layer{}.{}".format(int(stage) - 1, blk)
        if branch == "1":
            mod = "downsample.1" if bn else "downsample.0"
        else:
            mod = ("bn" if bn else "conv") + str("abc".index(branch[1]) + 1)
        return "{}.{}.{}".format(layer, mod, "bias" if kind == "b" else "weight")
    if name in ("conv1_w", "conv1_b"):
        return "conv1." + ("weight" if name[-1] == "w" else "bias")
    if name in ("res_conv1_bn_s", "res_conv1_bn_b"):
        return "bn1." + ("weight" if name[-1] == "s" else "bias")
    base, _, kind = name.rpartition("_")
    if base in _C2_HEADS and kind in ("w", "b"):
        return "{}.{}".format(_C2_HEADS[base], "weight" if kind == "w" else "bias")
    return None


def load_c2_format(cfg, f):
    """Caffe2 `.pkl` (e.g. the MSRA R-50 ImageNet weights the reference's first task starts from) -> {"model": state_dict}."""
    body = cfg.MODEL.BACKBONE.CONV_BODY
    if not (body.startswith("R-") and body.endswith("-C4")):
        raise KeyError("load_c2_format: only the ResNet-C4 bodies are supported here, got {}".format(body))
    with open(f, "rb") as fh:
        data = pickle.load(fh, encoding="latin1")
    blobs = data["blobs"] if "blobs" in data else data
    sd = {}
    for k in sorted(blobs):
        key = c2_blob_to_key(k)
        if key is not None:
            sd[key] = torch.from_numpy(blobs[k])
    return {"model": sd}


class Checkpointer(object):
    """Same files as the reference's Checkpointer (utils/checkpoint.py:13-103): `<save_dir>/<name>.pth` =
    {"model", "optimizer", "scheduler", **kwargs} (trim=True: model only, not tagged), `<save_dir>/last_checkpoint` holds the
    path of the newest full checkpoint and wins over the `f` argument on load.  "model" is written in the reference's
    layout (OIHW, unfused heads) whatever the in-memory layout is, so either implementation can resume the other's run."""

    def __init__(self, model, optimizer=None, scheduler=None, save_dir="", save_to_disk=None, logger=None):
        self.model, self.optimizer, self.scheduler = model, optimizer, scheduler
        self.save_dir, self.save_to_disk = save_dir, save_to_disk
        self.logger = logger if logger is not None else logging.getLogger(__name__)

    def save(self, name, trim=False, **kwargs):
        if not self.save_dir or not self.save_to_disk:
            return
        data = {"model": {k: v.cpu() for k, v in reference_state_dict(self.model).items()}}
        if not trim:
            if self.optimizer is not None:
                data["optimizer"] = self.optimizer.state_dict()
            if self.scheduler is not None:
                data["scheduler"] = self.scheduler.state_dict()
            data.update(kwargs)
        save_file = os.path.join(self.save_dir, "{}.pth".format(name))
        self.logger.info("Saving checkpoint to {}".format(save_file))
        torch.save(data, save_file)
        if not trim:
            self.tag_last_checkpoint(save_file)

    def load(self, f=None):
        if self.has_checkpoint():
            self.logger.info("Overriding ckpt config with last_checkpoint")
            f = self.get_checkpoint_file()
        if not f:
            self.logger.info("No checkpoint found. Initializing model from scratch")
            return {}
        self.logger.info("Loading checkpoint from {}".format(f))
        checkpoint = self._load_file(f)
        self._load_model(checkpoint)
        if "optimizer" in checkpoint and self.optimizer:
            self.optimizer.load_state_dict(checkpoint.pop("optimizer"))
        if "scheduler" in checkpoint and self.scheduler:
            self.scheduler.load_state_dict(checkpoint.pop("scheduler"))
        return checkpoint  # whatever else was saved (iteration, ...)

    def has_checkpoint(self):
        return os.path.exists(os.path.join(self.save_dir, "last_checkpoint"))

    def get_checkpoint_file(self):
        try:
            with open(os.path.join(self.save_dir, "last_checkpoint"), "r") as f:
                return f.read().strip()
        except IOError:  # removed by another process in between
            return ""

    def tag_last_checkpoint(self, last_filename):
        with open(os.path.join(self.save_dir, "last_checkpoint"), "w") as f:
            f.write(last_filename)

    def _load_file(self, f):
        return torch.load(f, map_location=torch.device("cpu"), weights_only=False)

    def _load_model(self, checkpoint):
        load_state_dict(self.model, checkpoint.pop("model"))


class DetectronCheckpointer(Checkpointer):
    """utils/checkpoint.py:106-142.  `catalog://` names and http(s) URLs resolve to downloads in the reference; this build
    has no network path, so they fail loudly with the file the caller must provide instead."""

    def __init__(self, cfg, model, optimizer=None, scheduler=None, save_dir="", save_to_disk=None, logger=None):
        super(DetectronCheckpointer, self).__init__(model, optimizer, scheduler, save_dir, save_to_disk, logger)
        self.cfg = cfg.clone() if hasattr(cfg, "clone") else cfg

    def _load_file(self, f):
        if f.startswith("catalog://") or f.startswith("http"):
            raise FileNotFoundError("{}: remote weights are not fetched here; download the file and pass its local path "
                                    "(MODEL.WEIGHT)".format(f))
        if f.endswith(".pkl"):
            return load_c2_format(self.cfg, f)
        loaded = super(DetectronCheckpointer, self)._load_file(f)
        if "model" not in loaded:
            loaded = dict(model=loaded)
        return loaded
