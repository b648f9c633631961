"""BoxCoder (mirror of maskrcnn_benchmark/modeling/box_coder.py:7-95) on the HIP glue kernels."""
import math

import torch

from .. import ops


class BoxCoder(object):
    def __init__(self, weights, bbox_xform_clip=math.log(1000.0 / 16)):
        self.weights = tuple(float(w) for w in weights)
        self.bbox_xform_clip = bbox_xform_clip  # the kernel hard-codes log(1000/16) (box_coder.py:20)

    def encode(self, reference_boxes, proposals):
        """deltas that take `proposals` to `reference_boxes` (:22-50); row-wise, both [n,4]"""
        n = proposals.shape[0]
        if n == 0:
            return proposals.new_zeros((0, 4))
        return ops.box_encode_rows(reference_boxes, proposals, self.weights)

    def decode(self, rel_codes, boxes):
        """[n,4k] deltas + [n,4] boxes -> [n,4k] boxes (:52-95), no clipping"""
        n = boxes.shape[0]
        k = rel_codes.shape[1] // 4
        if n == 0:
            return rel_codes.new_zeros((0, 4 * k))
        idx = torch.arange(n, device=boxes.device).view(1, n)
        hw = torch.tensor([[1 << 30, 1 << 30]], dtype=torch.int32, device=boxes.device)
        outs = [ops.rpn_decode_clip(rel_codes.contiguous().view(1, n, 4 * k), 4 * c, boxes.contiguous(), idx, hw, self.weights,
                                    clip=False)[0] for c in range(k)]
        return outs[0] if k == 1 else torch.stack(outs, 1).reshape(n, 4 * k)
