"""Data-parallel gradient exchange over the flat gradient buffer (DistributedDataParallel's job in the reference,
tools/train_incremental.py:231-235), laid out for xGMI: a handful of LARGE sum all-reduces instead of DDP's 25 MB buckets.

The flat buffer is cut into three buckets by the order their gradients become final during the backward pass:
    roi_heads  (layer4 + predictor, 15.1 M floats)  -> final once both RoI passes' pooled inputs have their gradient
    rpn        (3x3 conv + fused heads, 9.5 M)      -> final once the C4 feature map has its gradient
    backbone   (layer2 + layer3, 8.3 M)             -> final at the end of backward
`reduce_bucket_async(name)` is called from gradient hooks the trainer arms on those tensors (engine/trainer.py::_arm_overlap), so
the first two exchanges (75 % of the bytes) run on RCCL's stream underneath the rest of the backward pass; `finish()` issues whatever
is left and makes the current stream wait for all of them.

Every rank must issue the SAME sequence of collectives whatever its local state (which hooks fired, whether overlap is enabled,
whether any tensor of this rank's graph required a gradient): the sequence is therefore fixed -- always the per-range all-reduces
of BUCKET_ORDER, in that order; sending a bucket first sends every earlier one that is still outstanding (their gradients are final
by then: the RPN's become final after the RoI heads').  Hooks only move the issue POINT earlier, never the order or the sizes.
The 1/world factor is folded into the SGD kernel."""
import os

import torch
import torch.distributed as dist

OVERLAP = os.environ.get("ABR_ALLREDUCE_OVERLAP", "1") != "0"

BUCKET_ORDER = ("roi_heads", "rpn", "backbone")


def bucket_of(name):
    for b in BUCKET_ORDER[:-1]:
        if name.startswith(b + ".") or (".%s." % b) in name:
            return b
    return BUCKET_ORDER[-1]


def make_buckets(segments):
    """segments: FlatParams.segments [(name, start, end, is_bias)] ascending -> {bucket: [(a, b), ...]} with adjacent ranges merged."""
    out = {b: [] for b in BUCKET_ORDER}
    for name, a, b, _ in segments:
        r = out[bucket_of(name)]
        if r and r[-1][1] == a:
            r[-1] = (r[-1][0], b)
        else:
            r.append((a, b))
    return out


def modelled_ring_allreduce_ms(nbytes, world, link_gb_s=153.0, links=7, hop_us=8.0):
    """Ring all-reduce of `nbytes` over point-to-point xGMI (MI355X_MICROARCH / the task's hardware notes: 7 links x ~153 GB/s per GPU, each
    ring step bound by ONE link): 2 (w - 1) / w of the buffer crosses each link, plus a per-hop latency for the 2 (w - 1) steps.  RCCL spreads
    a large message over several rings (one per link direction), so the bandwidth term is divided by min(links, w - 1).  A MODEL for planning,
    printed next to the measured step time -- never a measurement."""
    if world <= 1:
        return 0.0
    rings = max(1, min(links, world - 1))
    return 2.0 * (world - 1) / world * nbytes / (link_gb_s * 1e9 * rings) * 1e3 + 2 * (world - 1) * hop_us * 1e-3


class GradReducer(object):
    def __init__(self, grads, segments, side_streams=()):
        """grads: the flat gradient tensor; side_streams: callable returning the streams (besides the current one) that write it."""
        self.grads = grads
        self.buckets = make_buckets(segments)
        self.side_streams = side_streams
        self.force = False          # single-rank RCCL runs in tests
        self._works, self._done = [], set()
        self._comm = None
        self._warm = False
        self.last_issue = []        # [(bucket, bytes, "backward-hook" | "optimizer.step")] of the most recent step, in issue order
        self._issue_log = []

    @property
    def active(self):
        return self.force or (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)

    def begin(self):
        """new backward pass: nothing reduced yet"""
        assert not self._works, "finish() was not called for the previous step"
        self._done.clear()
        self._issue_log = []
        if not self._warm and self.active:
            # the first collective creates the RCCL communicator; do that HERE, on the caller's thread, not inside a gradient hook
            # running on the autograd engine's thread in the middle of backward
            dist.all_reduce(torch.zeros(1, dtype=self.grads.dtype, device=self.grads.device))
            self._warm = True

    def _issue(self, name, where="optimizer.step"):
        """all-reduce bucket `name` and, before it, every earlier bucket of BUCKET_ORDER that has not gone out yet"""
        for b_name in BUCKET_ORDER[: BUCKET_ORDER.index(name) + 1]:
            if b_name in self._done:
                continue
            self._done.add(b_name)
            for a, b in self.buckets[b_name]:
                if b > a:
                    self._works.append(dist.all_reduce(self.grads[a:b], op=dist.ReduceOp.SUM, async_op=True))
            self._issue_log.append((b_name, sum(b - a for a, b in self.buckets[b_name]) * self.grads.element_size(), where))

    def describe(self):
        """the exchange as the last step issued it: one entry per bucket with its bytes, collective count and issue point"""
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        return [{"bucket": n, "bytes": by, "all_reduces": sum(1 for a, b in self.buckets[n] if b > a), "issued_from": w,
                 "modelled_ring_ms": round(modelled_ring_allreduce_ms(by, world), 4),
                 "exposed": w != "backward-hook"}       # a bucket sent from step() has no backward work left to hide under
                for n, by, w in self.last_issue]

    def reduce_bucket_async(self, name):
        """Called when every kernel that writes bucket `name` has been ENQUEUED (on the current stream or a side stream)."""
        if not (self.active and OVERLAP) or name in self._done:
            return
        if self.grads.is_cuda:
            cur = torch.cuda.current_stream()
            if self._comm is None:
                self._comm = torch.cuda.Stream(device=self.grads.device)
            self._comm.wait_stream(cur)
            for s in self.side_streams():
                self._comm.wait_stream(s)
            with torch.cuda.stream(self._comm):   # ProcessGroupNCCL orders its own stream after the CURRENT one
                self._issue(name, "backward-hook")
        else:
            self._issue(name, "backward-hook")

    def finish(self):
        """Issue the buckets still outstanding (the caller has joined its side streams) and wait for all of them."""
        if self.active:
            self._issue(BUCKET_ORDER[-1])    # everything still outstanding, in BUCKET_ORDER (the same collectives on every rank)
        for w in self._works:
            w.wait()                                  # GPU: the current stream waits on RCCL's; CPU (gloo): blocks
        self._works = []
        self._done.clear()
        self.last_issue = self._issue_log
