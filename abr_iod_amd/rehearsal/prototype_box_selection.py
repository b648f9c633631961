"""Prototype box selection driver (mirror of tools/prototype_box_selection.py:61-210).

`extract_bboxes_and_features` is the GPU part: the trained model's `generate_feature_logits_by_targets` (backbone + ROIAlign on
the ground-truth boxes + box head, all on the conv/ROIAlign kernels of the training path) and the per-(RoI, bin) channel mean of
the pooled features.  The reference moves the whole [n,1024,7,7] pooled tensor to the host and averages there
(:84 `torch.mean(roi_align_features.cpu(), dim=1)`, 200 KB per box over PCIe); here `abr_channel_mean` reduces it on the device
and 196 B per box cross the bus."""
import logging
import os
import time

import torch

from .. import ops
from ..layers._layout import as_nhwc
from .extract_memory import Mem

MIN_SIDE = 70  # boxes with width <= 70 AND height <= 70 (original pixels) are never stored (:97)


def extract_bboxes_and_features(model_source, data_loader, device, cfg):
    """-> per new class, a list of {'feature' [7][7], 'logits', 'image_path', 'box_class', 'box', 'mode'}.
    Batches are (images, targets, original_targets, idx): targets at network scale feed the model, original_targets (image
    scale) give the crop boxes.  Kept from the reference: the logits stored for box `ind` of image `img_n` are row
    `img_n + ind` of the batch's score matrix (:101), not the box's own row -- they are carried along but never used by any
    selection strategy."""
    n_old = len(cfg.MODEL.ROI_BOX_HEAD.NAME_OLD_CLASSES)
    n_new = len(cfg.MODEL.ROI_BOX_HEAD.NAME_NEW_CLASSES)
    logger = logging.getLogger("maskrcnn_benchmark_last_model.trainer")
    logger.info("Start sampling")
    model_source.eval()
    t0 = time.time()
    out = [[] for _ in range(n_new)]
    n_batches = 0
    for images, targets, original_targets, idx in data_loader:
        n_batches += 1
        images = images.to(device)
        targets = [t.to(device) for t in targets]
        with torch.no_grad():
            (scores, _), _, _, _, roi_align_features = model_source.generate_feature_logits_by_targets(images, targets)
            maps = ops.channel_mean(as_nhwc(roi_align_features))       # [n,7,7] on the device
        maps = maps.cpu().tolist()
        scores = scores.cpu()
        row = 0
        for img_n, target in enumerate(original_targets):
            labels = target.get_field("labels").cpu().tolist()
            boxes = target.bbox.cpu().tolist()
            for ind, (box, label) in enumerate(zip(boxes, labels)):
                row += 1
                if (box[2] - box[0]) <= MIN_SIDE and (box[3] - box[1]) <= MIN_SIDE:
                    continue
                out[label - n_old - 1].append({"feature": maps[row - 1], "logits": scores[img_n + ind], "image_path": idx[img_n],
                                               "box_class": label, "box": box, "mode": target.mode})
    dt = time.time() - t0
    logger.info("Total sampling time: {:.1f} s ({:.4f} s / it)".format(dt, dt / max(n_batches, 1)))
    return out


def selector(cfg_source, bbox_loader=None, model_source=None, image_root="data/VOCdevkit/VOC2007"):
    """:162-210.  Builds / loads the model unless one is passed, extracts features over `bbox_loader`, updates the memory folder
    `<OUTPUT_DIR>/<MEM_TYPE>_<MEM_BUFF>` and returns the list of files in it."""
    mem_dir = os.path.join(cfg_source.OUTPUT_DIR, "{}_{}".format(cfg_source.MEM_TYPE, cfg_source.MEM_BUFF))
    os.makedirs(mem_dir, exist_ok=True)
    step = cfg_source.STEP
    if step == 0 and len(os.listdir(mem_dir)) >= int(cfg_source.MEM_BUFF):
        info = None  # the first task's prototype boxes already exist
    else:
        if model_source is None:
            from ..modeling.detector.generalized_rcnn import build_detection_model
            from ..utils.checkpoint import DetectronCheckpointer
            model_source = build_detection_model(cfg_source)
            DetectronCheckpointer(cfg_source, model_source, save_dir=cfg_source.OUTPUT_DIR + "STEP{}".format(step)).load(cfg_source.MODEL.WEIGHT)
        if bbox_loader is None:
            raise ValueError("selector: pass the per-box data loader (the dataset pipeline is outside this package)")
        info = extract_bboxes_and_features(model_source, bbox_loader, torch.device(cfg_source.MODEL.DEVICE), cfg_source)
    mem = Mem(cfg_source, step, mem_dir, image_root=image_root)
    mem.update_memory(info)
    return os.listdir(mem.current_mem_path)
