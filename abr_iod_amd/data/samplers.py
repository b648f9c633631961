"""Index samplers of the data loader (mirror of maskrcnn_benchmark/data/samplers/{distributed,grouped_batch_sampler,
iteration_based_batch_sampler}.py and the helpers of data/build.py:64-106): which images a rank sees and how they are batched.
Host-side integer bookkeeping; the index sequences equal the reference's (tests/test_samplers.py, tests/golden/samplers.json)."""
import bisect
import math

import torch
import torch.distributed as dist


class DistributedSampler(object):
    """Rank `rank` of `num_replicas` gets a contiguous slice of the (epoch-seeded) permutation, padded by wrapping around so that
    every rank has ceil(len / num_replicas) samples (samplers/distributed.py:9-67)."""

    def __init__(self, dataset, num_replicas=None, rank=None, shuffle=True):
        if num_replicas is None:
            num_replicas = dist.get_world_size()
        if rank is None:
            rank = dist.get_rank()
        self.dataset, self.num_replicas, self.rank, self.shuffle = dataset, num_replicas, rank, shuffle
        self.epoch = 0
        self.num_samples = int(math.ceil(len(dataset) * 1.0 / num_replicas))
        self.total_size = self.num_samples * num_replicas

    def __iter__(self):
        n = len(self.dataset)
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.epoch)
            indices = torch.randperm(n, generator=g).tolist()
        else:
            indices = list(range(n))
        indices += indices[: self.total_size - n]
        lo = self.num_samples * self.rank
        return iter(indices[lo: lo + self.num_samples])

    def __len__(self):
        return self.num_samples

    def set_epoch(self, epoch):
        self.epoch = epoch


class GroupedBatchSampler(object):
    """Batches hold indices of ONE group (aspect-ratio bin) only; inside a group the base sampler's order is kept and cut into
    batch_size pieces; the batches are emitted in the order in which their first element appears in the base sampler
    (samplers/grouped_batch_sampler.py:9-115)."""

    def __init__(self, sampler, group_ids, batch_size, drop_uneven=False):
        self.sampler, self.group_ids = sampler, [int(g) for g in group_ids]
        self.batch_size, self.drop_uneven = batch_size, drop_uneven
        self._batches, self._can_reuse = None, False

    def _prepare_batches(self):
        sampled = list(self.sampler)
        position = {}
        for k, idx in enumerate(sampled):
            position[idx] = k          # a repeated index keeps its LAST position, as the reference's scatter does
        per_group = {}
        for idx in sorted(position, key=position.get):
            per_group.setdefault(self.group_ids[idx], []).append(idx)
        batches = []
        for g in sorted(per_group):
            ids = per_group[g]
            batches += [ids[i: i + self.batch_size] for i in range(0, len(ids), self.batch_size)]
        first_seen = {idx: k for k, idx in enumerate(sampled)}   # ... while the batch order uses the LAST occurrence too (dict build)
        batches.sort(key=lambda b: first_seen[b[0]])
        if self.drop_uneven:
            batches = [b for b in batches if len(b) == self.batch_size]
        return batches

    def __iter__(self):
        if self._can_reuse:
            self._can_reuse = False
        else:
            self._batches = self._prepare_batches()
        return iter(self._batches)

    def __len__(self):
        if self._batches is None:
            self._batches = self._prepare_batches()
            self._can_reuse = True
        return len(self._batches)


class BatchSampler(object):
    """torch.utils.data.BatchSampler(sampler, batch_size, drop_last=False) for samplers that are plain iterables"""

    def __init__(self, sampler, batch_size, drop_last=False):
        self.sampler, self.batch_size, self.drop_last = sampler, batch_size, drop_last

    def __iter__(self):
        batch = []
        for idx in self.sampler:
            batch.append(idx)
            if len(batch) == self.batch_size:
                yield batch
                batch = []
        if batch and not self.drop_last:
            yield batch

    def __len__(self):
        n = len(self.sampler)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size


class IterationBasedBatchSampler(object):
    """Re-iterates the wrapped batch sampler until `num_iterations` batches have been produced; the base sampler's epoch is set to
    the iteration count at every restart (samplers/iteration_based_batch_sampler.py:5-31)."""

    def __init__(self, batch_sampler, num_iterations, start_iter=0):
        self.batch_sampler, self.num_iterations, self.start_iter = batch_sampler, num_iterations, start_iter

    def __iter__(self):
        iteration = self.start_iter
        while iteration <= self.num_iterations:
            if hasattr(self.batch_sampler.sampler, "set_epoch"):
                self.batch_sampler.sampler.set_epoch(iteration)
            for batch in self.batch_sampler:
                iteration += 1
                if iteration > self.num_iterations:
                    break
                yield batch

    def __len__(self):
        return self.num_iterations


def quantize(x, bins):
    """build.py:73-77: index of the aspect-ratio bin of every value"""
    bins = sorted(bins)
    return [bisect.bisect_right(bins, y) for y in x]


def compute_aspect_ratios(dataset):
    """build.py:80-86: height / width from get_img_info"""
    out = []
    for i in range(len(dataset)):
        info = dataset.get_img_info(i)
        out.append(float(info["height"]) / float(info["width"]))
    return out


def make_batch_data_sampler(dataset, sampler, aspect_grouping, images_per_batch, num_iters=None, start_iter=0):
    """build.py:89-106"""
    if aspect_grouping:
        if not isinstance(aspect_grouping, (list, tuple)):
            aspect_grouping = [aspect_grouping]
        batch_sampler = GroupedBatchSampler(sampler, quantize(compute_aspect_ratios(dataset), aspect_grouping), images_per_batch, drop_uneven=False)
    else:
        batch_sampler = BatchSampler(sampler, images_per_batch, drop_last=False)
    if num_iters is not None:
        batch_sampler = IterationBasedBatchSampler(batch_sampler, num_iters, start_iter)
    return batch_sampler
