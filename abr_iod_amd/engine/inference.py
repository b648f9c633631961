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


def compute_on_dataset(model, data_loader, device, timer=None):
    """-> ({image id: BoxList on cpu}, {image id: background BoxList})  (inference.py:43-109).
    Batches are (images, targets, img_ids) or the reference's 4-tuple (images, targets, proposals, img_ids)."""
    model.eval()
    results, results_background = {}, {}
    for batch in data_loader:
        images, img_ids = batch[0], batch[-1]
        if hasattr(images, "to"):
            images = images.to(device)
        with torch.no_grad():
            t0 = time.perf_counter()
            output, _features, background = model(images)
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
