"""In-container ONLY: import the reference (YuyangSunshine/ABR_IOD @ /root/reference) so that golden
vectors can be generated from its own code.  Nothing under tests/ imports this at test time; only
tests/golden/make_golden*.py do, and those are run by hand in the build container (the GPU box has
no /root/reference).

What is stubbed, and why it does not touch the arithmetic being pinned:
  * apex.amp            -> identity decorators / context managers (reference runs amp O0 = no-op)
  * cv2, pycocotools    -> empty modules (only imported by mask/keypoint heads, never executed)
  * yacs.config.CfgNode -> a small attribute-dict with clone/freeze/merge (config plumbing only)
  * maskrcnn_benchmark._C -> the reference's OWN csrc compiled by oracle/Makefile (`make ref`)
  * numpy.float         -> float (alias removed in numpy>=1.24; anchor_generator.py:224 uses it)
"""
import ast
import contextlib
import copy
import importlib.util
import os
import sys
import types

import numpy as np
import yaml

REF = os.environ.get("ABR_REFERENCE", "/root/reference")
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class CfgNode(dict):
    """Minimal stand-in for yacs.config.CfgNode (attribute access + the 6 methods the reference calls)."""

    def __init__(self, init=None, **_):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return copy.deepcopy(self)

    def freeze(self):
        pass

    def defrost(self):
        pass

    @staticmethod
    def _coerce(v):
        if isinstance(v, str):
            try:
                return ast.literal_eval(v)
            except Exception:
                return v
        if isinstance(v, list):
            return tuple(v) if not any(isinstance(x, (dict, list)) for x in v) else v
        return v

    def _merge(self, other):
        for k, v in other.items():
            if isinstance(v, dict):
                if k not in self or not isinstance(self[k], CfgNode):
                    self[k] = CfgNode()
                self[k]._merge(v)
            else:
                self[k] = self._coerce(v)

    def merge_from_file(self, path):
        with open(path) as f:
            self._merge(yaml.safe_load(f) or {})

    def merge_from_list(self, lst):
        for k, v in zip(lst[0::2], lst[1::2]):
            node = self
            parts = k.split(".")
            for p in parts[:-1]:
                node = node[p]
            node[parts[-1]] = self._coerce(v)


def _install_stubs():
    if not hasattr(np, "float"):
        np.float = float  # noqa: NPY001 (reference uses the removed alias)

    apex = types.ModuleType("apex")
    amp = types.ModuleType("apex.amp")
    amp.float_function = lambda f: f
    amp.half_function = lambda f: f
    amp.initialize = lambda m, o, opt_level=None, **kw: (m, o)

    @contextlib.contextmanager
    def scale_loss(loss, optimizer):
        yield loss

    amp.scale_loss = scale_loss
    apex.amp = amp
    sys.modules.setdefault("apex", apex)
    sys.modules.setdefault("apex.amp", amp)

    for name in ("cv2", "pycocotools", "pycocotools.mask"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["pycocotools"].mask = sys.modules["pycocotools.mask"]

    yacs = types.ModuleType("yacs")
    yacs_config = types.ModuleType("yacs.config")
    yacs_config.CfgNode = CfgNode
    yacs.config = yacs_config
    sys.modules.setdefault("yacs", yacs)
    sys.modules.setdefault("yacs.config", yacs_config)


def load_ref_C():
    so = os.path.join(REPO, "oracle", "_ref", "_C.so")
    if not os.path.exists(so):
        raise RuntimeError("run `make -C oracle ref` first (builds the reference's own csrc)")
    spec = importlib.util.spec_from_file_location("_C", so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


_READY = False


def setup():
    """Make `import maskrcnn_benchmark...` work against /root/reference.  Idempotent."""
    global _READY
    if _READY:
        return
    if not os.path.isdir(REF):
        raise RuntimeError(f"reference tree not found at {REF}; goldens can only be regenerated in the build container")
    _install_stubs()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import maskrcnn_benchmark  # noqa: F401  (namespace package root)

    C = load_ref_C()
    sys.modules["maskrcnn_benchmark._C"] = C
    maskrcnn_benchmark._C = C
    _READY = True


def default_cfg(yaml_rel=None, overrides=()):
    setup()
    from maskrcnn_benchmark.config import cfg as _cfg

    cfg = _cfg.clone()
    if yaml_rel:
        cfg.merge_from_file(os.path.join(REF, yaml_rel))
    if overrides:
        cfg.merge_from_list(list(overrides))
    return cfg
