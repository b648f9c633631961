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
# Who issues the collectives.  "torch" (default): torch.distributed.all_reduce under backend "nccl" (= RCCL on ROCm; gloo in the CPU tests) -- its
# ProcessGroup owns the communicator and the communication stream.  "abr": the library's own RCCL communicator (csrc/comm.hip, abr_allreduce_flat):
# one RCCL group per bucket enqueued on the reducer's stream, the communicator created from a unique id that rank 0 hands out through the existing
# torch.distributed group.  Same buckets, same order, same sums either way (tests/test_gpu_dist.py runs both on one rank); the default stays "torch"
# until an N > 1 run on hardware has compared them (the pool's boxes have one GPU).
BACKEND = os.environ.get("ABR_ALLREDUCE_BACKEND", "torch")

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


class _EventWork(object):
    """what torch's async Work is to the torch backend: `wait()` orders the CURRENT stream behind the collective"""

    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)


def rccl_debug_summary(path, max_lines=24):
    """the few lines of an NCCL_DEBUG=INFO log (NCCL_DEBUG_FILE) that say what RCCL chose: channel counts, rings / trees, and -- with
    NCCL_DEBUG_SUBSYS=INIT,TUNING -- algorithm and protocol per collective size.  De-duplicated, capped; [] when there is no log."""
    import glob
    import re
    out, seen = [], set()
    for f in sorted(glob.glob(path.replace("%p", "*").replace("%h", "*"))):
        try:
            lines = open(f, errors="replace").read().splitlines()
        except OSError:
            continue
        for ln in lines:
            if not re.search(r"channels|Algo|algorithm|protocol|Connected all|nranks|Using network|comm 0x", ln, flags=re.I):
                continue
            key = re.sub(r"^\S+:\d+:\d+ \[\d+\] ", "", ln)          # drop host:pid:tid [dev]
            key = re.sub(r"0x[0-9a-f]+", "0x..", key)
            if key in seen:
                continue
            seen.add(key)
            out.append(key.strip()[:200])
            if len(out) >= max_lines:
                return out
    return out


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
        self.backend = BACKEND if grads.is_cuda else "torch"
        self._abr_comm = None       # the library's communicator handle (backend "abr")
        # bench.py --gpus N: time, per bucket, how long the main stream had to WAIT for the exchange after backward had finished (events)
        self.measure = False
        self._wait_events = None
        self.last_exposed_ms = None

    @property
    def active(self):
        return self.force or (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)

    def _ensure_abr_comm(self):
        """backend "abr": the library's RCCL communicator over the ranks of the torch.distributed group (1 rank when forced in a test)"""
        if self._abr_comm is not None:
            return self._abr_comm
        import ctypes as C
        from .. import _lib as L
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        world, rank = (dist.get_world_size(), dist.get_rank()) if multi else (1, 0)
        idb = C.create_string_buffer(128)
        if rank == 0:
            L.check(L.lib().abr_comm_unique_id(C.cast(idb, C.c_void_p)), "comm_unique_id")
        if multi:
            box = [bytes(idb.raw)]
            dist.broadcast_object_list(box, src=0)       # the out-of-band channel: the group that already exists
            idb = C.create_string_buffer(box[0], 128)
        comm = C.c_void_p()
        L.check(L.lib().abr_comm_init(world, rank, C.cast(idb, C.c_void_p), C.byref(comm)), "comm_init")
        self._abr_comm = comm
        return comm

    def close(self):
        if self._abr_comm is not None:
            from .. import _lib as L
            torch.cuda.synchronize()
            L.check(L.lib().abr_comm_destroy(self._abr_comm), "comm_destroy")
            self._abr_comm = None

    def begin(self):
        """new backward pass: nothing reduced yet"""
        assert not self._works, "finish() was not called for the previous step"
        self._done.clear()
        self._issue_log = []
        if not self._warm and self.active and self.backend == "abr":
            self._ensure_abr_comm()      # (collective; here, on the caller's thread -- see below)
            self._warm = True
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
            if self.backend == "abr":
                import ctypes as C
                from .. import _lib as L
                rs = [(a, b) for a, b in self.buckets[b_name] if b > a]
                if rs:
                    arr = (C.c_int64 * (2 * len(rs)))(*[v for r in rs for v in r])
                    # ONE RCCL group for the bucket's ranges, on the current stream (the reducer's stream under a hook, the caller's in finish())
                    L.check(L.lib().abr_allreduce_flat(self._ensure_abr_comm(), self.grads.data_ptr(), C.cast(arr, C.c_void_p), len(rs), L.stream()),
                            "allreduce_flat")
                    ev = torch.cuda.Event()
                    ev.record()
                    self._works.append(_EventWork(ev))
            else:
                for a, b in self.buckets[b_name]:
                    if b > a:
                        self._works.append(dist.all_reduce(self.grads[a:b], op=dist.ReduceOp.SUM, async_op=True))
            self._works.append(None)     # end of this bucket's collectives (finish() records its timing event there)
            self._issue_log.append((b_name, sum(b - a for a, b in self.buckets[b_name]) * self.grads.element_size(), where))

    def describe(self):
        """the exchange as the last step issued it: one entry per bucket with its bytes, collective count and issue point"""
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        out = [{"bucket": n, "bytes": by, "all_reduces": sum(1 for a, b in self.buckets[n] if b > a), "issued_from": w,
                "modelled_ring_ms": round(modelled_ring_allreduce_ms(by, world), 4),
                "exposed": w != "backward-hook"}       # a bucket sent from step() has no backward work left to hide under
               for n, by, w in self.last_issue]
        ex = self.exposed_ms()
        if ex is not None:
            # measured on the last step: ms between the end of backward on the main stream (side streams joined) and the moment the main stream
            # had every collective up to and including this bucket's behind it -- 0 = fully hidden under the backward pass
            for row, v in zip(out, ex):
                row["main_stream_wait_ms_cumulative"] = v
        for row in out:
            row["backend"] = self.backend
        return out

    def exposed_ms(self):
        """per bucket (issue order) of the last measured step: see describe().  Synchronises on the events."""
        if not self._wait_events:
            return self.last_exposed_ms
        ev0, evs = self._wait_events
        evs[-1].synchronize()
        self.last_exposed_ms = [round(max(0.0, ev0.elapsed_time(e)), 4) for e in evs]
        self._wait_events = None
        return self.last_exposed_ms

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
        timed = self.measure and self.active and self.grads.is_cuda
        if timed:
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record()                              # backward is over on the main stream (its side streams are joined)
        if self.active:
            self._issue(BUCKET_ORDER[-1])    # everything still outstanding, in BUCKET_ORDER (the same collectives on every rank)
        evs = []
        for w in self._works:
            if w is not None:
                w.wait()                              # GPU: the current stream waits on RCCL's; CPU (gloo): blocks
            elif timed:                               # (a bucket's last collective is behind the main stream here)
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                evs.append(e)
        self._wait_events = (ev0, evs) if (timed and evs) else None
        self._works = []
        self._done.clear()
        self.last_issue = self._issue_log
