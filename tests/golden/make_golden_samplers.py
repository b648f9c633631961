#!/usr/bin/env python3
"""Sampler fixtures FROM THE REFERENCE (in-container only): index sequences of maskrcnn_benchmark/data/samplers/*.py, loaded by file
path (the data package's __init__ chain needs torchvision, which is not installed and not used by the samplers)."""
import importlib.util
import json
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/maskrcnn_benchmark/data/samplers/"


def load(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, REF + name + ".py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    D = load("distributed").DistributedSampler
    G = load("grouped_batch_sampler").GroupedBatchSampler
    I = load("iteration_based_batch_sampler").IterationBasedBatchSampler
    out = {}
    ds = list(range(23))
    for shuffle in (True, False):
        for world in (1, 2, 4):
            for rank in range(world):
                s = D(ds, num_replicas=world, rank=rank, shuffle=shuffle)
                seqs = []
                for epoch in (0, 3):
                    s.set_epoch(epoch)
                    seqs.append(list(s))
                out["dist_{}_{}_{}".format(int(shuffle), world, rank)] = seqs
    g = torch.Generator().manual_seed(1)
    group_ids = torch.randint(0, 3, (23,), generator=g).tolist()
    out["group_ids"] = group_ids
    for world, rank in ((1, 0), (2, 1)):
        for drop in (False, True):
            s = D(ds, num_replicas=world, rank=rank, shuffle=True)
            s.set_epoch(5)
            gb = G(s, group_ids, 4, drop_uneven=drop)
            out["grouped_{}_{}_{}".format(world, rank, int(drop))] = {"len": len(gb), "batches": [list(b) for b in gb]}
    s = D(ds, num_replicas=2, rank=0, shuffle=True)
    it = I(G(s, group_ids, 4, drop_uneven=False), num_iterations=9, start_iter=2)
    out["iteration_based"] = {"len": len(it), "batches": [list(b) for b in it]}
    with open(os.path.join(HERE, "samplers.json"), "w") as f:
        json.dump(out, f)
    print("wrote samplers.json", len(out))


if __name__ == "__main__":
    main()
