"""Matcher (mirror of maskrcnn_benchmark/modeling/matcher.py:5-112).  The training path never builds the IoU
matrix: IoU + thresholds + low-quality rule + labels + BoxCoder.encode are ONE kernel (ops.match_encode)."""
import torch

from .. import ops


class Matcher(object):
    BELOW_LOW_THRESHOLD = -1
    BETWEEN_THRESHOLDS = -2

    def __init__(self, high_threshold, low_threshold, allow_low_quality_matches=False):
        assert low_threshold <= high_threshold
        self.high_threshold, self.low_threshold = high_threshold, low_threshold
        self.allow_low_quality_matches = allow_low_quality_matches

    def match_boxes(self, gt_boxes, boxes, gt_labels=None, visibility=None, weights=(1.0, 1.0, 1.0, 1.0), rpn_labels=False):
        """fused path -> (matched_idxs int64 [n], labels, regression_targets [n,4])"""
        if gt_boxes.shape[0] == 0:  # matcher.py:53-57
            raise ValueError("No ground-truth boxes available for one of the images during training")
        if boxes.shape[0] == 0:     # :58-62
            raise ValueError("No proposal boxes available for one of the images during training")
        return ops.match_encode(boxes, gt_boxes, gt_labels, visibility, self.high_threshold, self.low_threshold,
                                self.allow_low_quality_matches, weights, rpn_labels)

    def __call__(self, match_quality_matrix):
        """reference signature (IoU matrix [G,n] in, matches [n] out) for callers that already hold the matrix."""
        if match_quality_matrix.numel() == 0:
            raise ValueError("No ground-truth boxes available for one of the images during training"
                             if match_quality_matrix.shape[0] == 0 else "No proposal boxes available for one of the images during training")
        matched_vals, matches = match_quality_matrix.max(dim=0)
        all_matches = matches.clone() if self.allow_low_quality_matches else None
        matches[matched_vals < self.low_threshold] = Matcher.BELOW_LOW_THRESHOLD
        matches[(matched_vals >= self.low_threshold) & (matched_vals < self.high_threshold)] = Matcher.BETWEEN_THRESHOLDS
        if self.allow_low_quality_matches:
            best, _ = match_quality_matrix.max(dim=1)
            upd = torch.nonzero(match_quality_matrix == best[:, None])[:, 1]
            matches[upd] = all_matches[upd]
        return matches
