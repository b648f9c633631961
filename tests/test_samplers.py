"""Index samplers (data/samplers.py) against the sequences the reference's samplers produce (tests/golden/samplers.json, written by
tests/golden/make_golden_samplers.py from maskrcnn_benchmark/data/samplers/*.py)."""
import json
import os

from abr_iod_amd.data.samplers import (BatchSampler, DistributedSampler, GroupedBatchSampler, IterationBasedBatchSampler, make_batch_data_sampler,
                                      quantize)

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "samplers.json")))
DS = list(range(23))


def test_distributed_sampler_sequences():
    for shuffle in (True, False):
        for world in (1, 2, 4):
            seen = []
            for rank in range(world):
                s = DistributedSampler(DS, num_replicas=world, rank=rank, shuffle=shuffle)
                seqs = []
                for epoch in (0, 3):
                    s.set_epoch(epoch)
                    seqs.append(list(s))
                assert seqs == GOLD["dist_{}_{}_{}".format(int(shuffle), world, rank)]
                assert len(s) == len(seqs[0])
                seen += seqs[0]
            assert set(seen) == set(DS)   # every image lands on some rank


def test_grouped_and_iteration_based_samplers():
    gids = GOLD["group_ids"]
    for world, rank in ((1, 0), (2, 1)):
        for drop in (False, True):
            s = DistributedSampler(DS, num_replicas=world, rank=rank, shuffle=True)
            s.set_epoch(5)
            gb = GroupedBatchSampler(s, gids, 4, drop_uneven=drop)
            g = GOLD["grouped_{}_{}_{}".format(world, rank, int(drop))]
            assert len(gb) == g["len"] and [list(b) for b in gb] == g["batches"]
            assert all(len({gids[i] for i in b}) == 1 for b in g["batches"])
    s = DistributedSampler(DS, num_replicas=2, rank=0, shuffle=True)
    it = IterationBasedBatchSampler(GroupedBatchSampler(s, gids, 4), num_iterations=9, start_iter=2)
    assert len(it) == GOLD["iteration_based"]["len"] and [list(b) for b in it] == GOLD["iteration_based"]["batches"]


def test_make_batch_data_sampler_and_quantize():
    class DSInfo(list):
        def get_img_info(self, i):
            return {"height": 300 + 100 * (i % 3), "width": 400}
    ds = DSInfo(range(10))
    assert quantize([0.5, 1.0, 1.5], [1]) == [0, 1, 1]
    bs = make_batch_data_sampler(ds, DistributedSampler(ds, 1, 0, shuffle=False), aspect_grouping=[1], images_per_batch=2)
    batches = list(bs)
    assert sorted(i for b in batches for i in b) == list(range(10))
    assert all(len({(300 + 100 * (i % 3)) / 400 >= 1 for i in b}) == 1 for b in batches)
    plain = make_batch_data_sampler(ds, DistributedSampler(ds, 1, 0, shuffle=False), aspect_grouping=0, images_per_batch=4, num_iters=5)
    assert len(list(plain)) == 5 and isinstance(plain.batch_sampler, BatchSampler)
