"""BalancedPositiveNegativeSampler (mirror of maskrcnn_benchmark/modeling/balanced_positive_negative_sampler.py:5-77).
Random choice uses the device RNG exactly as the reference (torch.randperm per image); the draws are index glue, not
arithmetic, and cannot be reproduced across devices -> parity tests inject the sampled indices instead."""
import torch

from .. import ops


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

    def sample_padded(self, labels2d, index_offset_per_image=0):
        """Fused device path (one kernel for all rows of labels2d [N,n], no host sync):
        -> pos_idx [N,num_pos_max], neg_idx [N,batch] (ascending, -1 padded, + i*index_offset_per_image), counts [N,2] int32."""
        return ops.sample_pos_neg(labels2d, self.batch_size_per_image, int(self.batch_size_per_image * self.positive_fraction),
                                  index_offset_per_image)

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
