"""CPU: the small-table caches of abr_iod_amd/ops.py never drop an entry a caller may still be holding for the launch it is assembling
(ops._evict_oldest: oldest half out, newest kept -- not cache.clear()), and the optimiser's gradient-writer stream set is limited to the
weight-gradient streams (host logic only, no device)."""
import collections


def test_evict_oldest_keeps_the_newest_entries():
    from abr_iod_amd import ops
    cache = collections.OrderedDict((i, object()) for i in range(10))
    ops._evict_oldest(cache, 64)                      # below the limit: untouched
    assert list(cache) == list(range(10))
    cache = {i: object() for i in range(64)}
    newest = cache[63]
    ops._evict_oldest(cache, 64)                      # at the limit: the oldest half goes, the most recent entries stay
    assert list(cache) == list(range(32, 64)) and cache[63] is newest
    cache[64] = object()
    ops._evict_oldest(cache, 64)
    assert 64 in cache and 63 in cache


def test_grad_writer_stream_filter():
    """solver/build.py::FusedSGD._grad_writer_streams picks (device) and (device, 'wgradN') keys only"""
    keys = [0, (0, "wgrad1"), (0, "source-model"), (0, "proposals"), (0, "weight-prep"), (1, "wgrad1"), 1]
    dev = 0
    picked = [k for k in keys if (k == dev) or (isinstance(k, tuple) and k[0] == dev and str(k[1]).startswith("wgrad"))]
    assert picked == [0, (0, "wgrad1")]


def test_prefetch_key_follows_the_batch_the_weights_and_the_arithmetic():
    """engine/trainer.py::_prefetch_key: work prefetched for the next batch (the frozen source model's forward, the target's frozen prefix) is
    only reused for THE batch it was computed from, with the weights and the arithmetic it was computed with (object identity
    alone reused stale features after an in-place refill of the batch buffer, a checkpoint load or a change of the contraction arithmetic)."""
    import types
    import torch
    from abr_iod_amd.engine import trainer
    from abr_iod_amd.modeling.backbone import resnet
    ms, mt = types.SimpleNamespace(conv_math="bf16x6"), types.SimpleNamespace(conv_math="bf16x6")
    x = torch.zeros(2, 3, 8, 8)
    k0 = trainer._prefetch_key(x, ms, mt)
    assert trainer._prefetch_key(x, ms, mt) == k0
    x.add_(1.0)                                   # the same buffer refilled in place: a different batch
    k1 = trainer._prefetch_key(x, ms, mt)
    assert k1 != k0
    assert trainer._prefetch_key(x.clone(), ms, mt) != k1     # another object / storage
    resnet.bump_param_version()                   # checkpoint load / in-place weight surgery
    k2 = trainer._prefetch_key(x, ms, mt)
    assert k2 != k1
    resnet.bump_trained_version()                 # an optimiser step moves only trained tensors: frozen-model prefetches stay valid
    assert trainer._prefetch_key(x, ms, mt) == k2
    mt.conv_math = "f32"                          # the range guard switched the arithmetic
    assert trainer._prefetch_key(x, ms, mt) != k2
    # the state lives on the target model object, not in the module
    a, b = types.SimpleNamespace(), types.SimpleNamespace()
    sa, sb = trainer.trainer_state(a), trainer.trainer_state(b)
    assert sa is trainer.trainer_state(a) and sa is not sb
    sa.prefetched = {"images": x}
    assert sb.prefetched == {}


def test_frozen_prefix_is_shared_only_between_identical_frozen_weights():
    """engine/trainer.py::frozen_prefix_shareable (the opt-in ABR_SHARE_FROZEN_PREFIX): tensors are compared, never assumed equal; the verdict is
    cached per weight version and re-made when resnet._STATIC_VERSION moves (checkpoint load / in-place surgery)."""
    import torch
    from abr_iod_amd.engine import trainer
    from abr_iod_amd.modeling.backbone import resnet

    class Body(torch.nn.Module):
        def __init__(self, frozen=("layer1",)):
            super().__init__()
            self.stem = torch.nn.Conv2d(3, 4, 3)
            self.layer1 = torch.nn.Sequential(torch.nn.Conv2d(4, 4, 1), torch.nn.BatchNorm2d(4))
            self.layer2 = torch.nn.Conv2d(4, 4, 1)
            self._frozen = list(frozen)
            for m in [self.stem] + [getattr(self, n) for n in frozen]:
                for p in m.parameters():
                    p.requires_grad = False

        def frozen_stage_names(self):
            return list(self._frozen)

    class Model(object):
        def __init__(self, body, math="bf16x6"):
            self.backbone = type("B", (), {})()
            self.backbone.body = body
            self.conv_math = math

    torch.manual_seed(0)
    bs = Body()
    bt = Body()
    bt.load_state_dict(bs.state_dict())
    with torch.no_grad():
        bt.layer2.weight.add_(1.0)                       # trainable stages may differ
    ms, mt = Model(bs), Model(bt)
    assert trainer.frozen_prefix_shareable(ms, mt)
    with torch.no_grad():
        bs.layer1[1].running_mean[2] += 1e-3             # one frozen BUFFER moves ...
    assert trainer.frozen_prefix_shareable(ms, mt)       # ... the cached verdict stands until the weight version moves
    resnet._STATIC_VERSION[0] += 1
    assert not trainer.frozen_prefix_shareable(ms, mt)
    bt.load_state_dict(bs.state_dict())
    resnet._STATIC_VERSION[0] += 1
    assert trainer.frozen_prefix_shareable(ms, mt)
    assert not trainer.frozen_prefix_shareable(ms, Model(Body(frozen=()), "bf16x6"))       # different frozen sets
    assert not trainer.frozen_prefix_shareable(ms, Model(bt, "f32"))                        # different arithmetic
