"""BalancedPositiveNegativeSampler (mirror of maskrcnn_benchmark/modeling/balanced_positive_negative_sampler.py:5-77).
Random choice uses the device RNG exactly as the reference (torch.randperm per image); the draws are index glue, not
arithmetic, and cannot be reproduced across devices -> parity tests inject the sampled indices instead."""
import torch


class BalancedPositiveNegativeSampler(object):
    def __init__(self, batch_size_per_image, positive_fraction):
        self.batch_size_per_image, self.positive_fraction = batch_size_per_image, positive_fraction

    def sample_indices(self, labels):
        """one image: labels [n] (>=1 positive, 0 negative, -1 ignored) -> (pos_idx, neg_idx) int64 index tensors"""
        positive = torch.nonzero(labels >= 1).squeeze(1)
        negative = torch.nonzero(labels == 0).squeeze(1)
        num_pos = min(positive.numel(), int(self.batch_size_per_image * self.positive_fraction))
        num_neg = min(negative.numel(), self.batch_size_per_image - num_pos)
        pos = positive[torch.randperm(positive.numel(), device=positive.device)[:num_pos]]
        neg = negative[torch.randperm(negative.numel(), device=negative.device)[:num_neg]]
        return pos, neg

    def __call__(self, matched_idxs, objectness=None):
        pos_idx, neg_idx = [], []
        for lab in matched_idxs:
            pos, neg = self.sample_indices(lab)
            pm = torch.zeros_like(lab, dtype=torch.uint8)
            nm = torch.zeros_like(lab, dtype=torch.uint8)
            pm[pos] = 1
            nm[neg] = 1
            pos_idx.append(pm)
            neg_idx.append(nm)
        return pos_idx, neg_idx
