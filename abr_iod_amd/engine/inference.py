"""Test loop (mirror of maskrcnn_benchmark/engine/inference.py:43-213): run the detector in eval mode over a data loader,
collect per-image detections on the host, merge the ranks' shares and hand them to the dataset's metric (VOC mAP).

One process per GPU as in training: each rank evaluates its shard of the loader; the merge is an object all-gather
(the reference has that call commented out, inference.py:144, and so only ever evaluates rank 0's shard)."""
import logging
import os
import time

import torch
import torch.distributed as dist

from ..data.datasets.evaluation import evaluate
from ..utils.comm import get_world_size, is_main_process, synchronize


# The fp32-accurate split arithmetics have a DOMAIN: f16x3 keeps only an absolute accuracy of 2^-40 amax for operand elements more
# than 18 binades below their tensor's amax, and both splits turn an inf / nan operand into NaN.  The kernels count / flag such operands; the
# training loop polls those words (engine/trainer.py::_x6_guard) -- and so does this loop: a batch whose operands left the domain is RE-RUN in the
# next arithmetic down (f16x3 -> bf16x6 -> fp32 MFMA), so no detection handed to the metric was computed outside it.  The read is synchronous and
# per batch: the detections are copied to the host per batch anyway.  ABR_EVAL_GUARD=0: off.
EVAL_GUARD = os.environ.get("ABR_EVAL_GUARD", "1") != "0"


class EvalRangeGuard(object):
    """`forward(images)` = model(images) under the range guard; `.stats` = what it saw (bench_eval.py prints it)."""

    def __init__(self, model, log=None):
        from .. import ops
        self.ops, self.model = ops, model
        self.log = log or logging.getLogger("maskrcnn_benchmark_target_model.inference")
        self.stats = {"batches": 0, "reruns": 0, "small": 0, "seen": 0, "max_small_fraction": 0.0, "flags": 0}
        self.limit = float(os.environ.get("ABR_H3_MAX_SMALL_FRACTION", "0.05"))
        self._clean = False

    def active(self):
        return EVAL_GUARD and getattr(self.model, "conv_math", None) in ("f16x3", "bf16x6") and hasattr(self.model, "set_conv_math")

    def forward(self, images):
        ops, model = self.ops, self.model
        if not self.active() or not (images.tensors if hasattr(images, "tensors") else images).is_cuda:
            return model(images)
        if not self._clean:      # whatever an earlier phase of the process (training steps) left in the words is not this batch's
            ops.x6_range_flags(reset=True)
            ops.h3_range_stats(reset=True)
            self._clean = True
        for _attempt in range(3):
            out = model(images)
            math = model.conv_math
            flags = ops.x6_range_flags(reset=True)       # (synchronises the stream: the batch's kernels have all reported)
            small, seen = ops.h3_range_stats(reset=True) if math == "f16x3" else (0, 0)
            st = self.stats
            st["batches"] += 1
            st["flags"] |= flags
            st["small"] += small
            st["seen"] += seen
            frac = small / seen if seen else 0.0
            st["max_small_fraction"] = max(st["max_small_fraction"], frac)
            if flags & ops.H3_FLAG_STALE:
                raise RuntimeError("f16x3: a kernel was handed an amax word that did not carry the epoch it was told (a host plumbing bug, not data)")
            if flags & ops.X6_FLAG_NONFINITE:
                nxt = "f32"
                why = "an inf / nan operand"
            elif math == "f16x3" and frac > self.limit:
                nxt = "bf16x6"
                why = "{:.1%} of the operand elements more than 18 binades below their tensor's amax (limit {:.1%})".format(frac, self.limit)
            else:
                return out
            self.log.warning("eval range guard: {} in the {} arithmetic -- switching the model to {} and re-running the batch".format(why, math, nxt))
            model.set_conv_math(nxt)
            st["reruns"] += 1
        return out


def compute_on_dataset(model, data_loader, device, timer=None, guard=None):
    """-> ({image id: BoxList on cpu}, {image id: background BoxList})  (inference.py:43-109).
    Batches are (images, targets, img_ids) or the reference's 4-tuple (images, targets, proposals, img_ids)."""
    model.eval()
    results, results_background = {}, {}
    if guard is None:
        guard = EvalRangeGuard(model)
    for batch in data_loader:
        images, img_ids = batch[0], batch[-1]
        if hasattr(images, "to"):
            images = images.to(device)
        with torch.no_grad():
            t0 = time.perf_counter()
            output, _features, background = guard.forward(images)
            if timer is not None:
                torch.cuda.synchronize()
                timer["total"] = timer.get("total", 0.0) + time.perf_counter() - t0
        output = [o.to("cpu") for o in output]
        results.update({i: o for i, o in zip(img_ids, output)})
        # the reference keeps ONE background list per batch, keyed by the batch's first image id (inference.py:107)
        results_background[img_ids[0]] = background.to("cpu") if background is not None else None
    return results, results_background


def _accumulate_predictions_from_multiple_gpus(predictions_per_gpu):
    """Merge {image id: prediction} over ranks; rank 0 returns the list ordered by image id (inference.py:143-160)."""
    if get_world_size() > 1:
        gathered = [None] * get_world_size()
        dist.all_gather_object(gathered, predictions_per_gpu)
    else:
        gathered = [predictions_per_gpu]
    if not is_main_process():
        return None
    predictions = {}
    for p in gathered:
        predictions.update(p)
    image_ids = sorted(predictions.keys())
    if image_ids and len(image_ids) != image_ids[-1] + 1:
        logging.getLogger("maskrcnn_benchmark_target_model.inference").warning(
            "Number of images that were gathered from multiple processes is not a contiguous set. "
            "Some images might be missing from the evaluation")
    return [predictions[i] for i in image_ids]


def inference(model, data_loader, dataset_name, iou_types=("bbox",), box_only=False, device="cuda", expected_results=(),
              expected_results_sigma_tol=4, output_folder=None, external_proposal=False, alphabetical_order=True,
              summary_writer=None, save_predictions=False):
    """inference.py:163-213.  Returns the metric dict on rank 0 ({"ap", "map"} for VOC), None elsewhere."""
    if external_proposal:
        raise NotImplementedError("external (edge-box) proposals are outside the RPN path this package implements")
    logger = logging.getLogger("maskrcnn_benchmark_target_model.inference")
    dataset = data_loader.dataset
    n = max(len(dataset), 1)
    logger.info("Start evaluation on {} dataset({} images).".format(dataset_name, len(dataset)))
    timer = {}
    t0 = time.perf_counter()
    predictions, _background = compute_on_dataset(model, data_loader, torch.device(device), timer)
    synchronize()
    total = time.perf_counter() - t0
    world = get_world_size()
    logger.info("Total run time: {:.1f} s ({:.4f} s / img per device, on {} devices)".format(total, total * world / n, world))
    logger.info("Model inference time: {:.1f} s ({:.4f} s / img per device, on {} devices)".format(
        timer.get("total", 0.0), timer.get("total", 0.0) * world / n, world))
    predictions = _accumulate_predictions_from_multiple_gpus(predictions)
    if not is_main_process():
        return None
    if output_folder and save_predictions:
        torch.save(predictions, os.path.join(output_folder, "predictions.pth"))
    return evaluate(dataset=dataset, predictions=predictions, output_folder=output_folder, box_only=box_only, iou_types=iou_types,
                    expected_results=expected_results, expected_results_sigma_tol=expected_results_sigma_tol,
                    alphabetical_order=alphabetical_order)
