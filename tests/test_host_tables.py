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
