"""Synthetic VOC-shaped workload for the benchmark / smoke test / parity tests:
configs for an incremental task, seeded random-init source + target models, 600x1000 image batches with a few GT boxes.
There is no network access for datasets or checkpoints; `data` is reported as "synthetic" by bench.py."""
import torch

from ..config import cfg as _default_cfg
from ..modeling.detector.generalized_rcnn import build_detection_model
from ..structures.bounding_box import BoxList

VOC_CLASSES = ["aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow", "diningtable", "dog",
               "horse", "motorbike", "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor"]

TASKS = {"15-5": (15, 5), "10-10": (10, 10), "19-1": (19, 1), "10-5": (10, 5)}


def make_cfgs(task="15-5", dist_type="id", feat="ard", alpha=0.5, beta=1.0, gamma=1.0, ims_per_batch=4, base_lr=0.002,
              overrides=()):
    """(cfg_source, cfg_target) as tools/train_incremental.py:421-466 derives them from configs/voc/<task>/*_RB_Target_model.yaml."""
    n_old, n_new = TASKS[task]
    base = _default_cfg.clone()
    base.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION = 7
    base.MODEL.ROI_BOX_HEAD.POOLER_SCALES = (0.0625,)
    base.MODEL.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO = 0
    base.MODEL.ROI_BOX_HEAD.NAME_OLD_CLASSES = VOC_CLASSES[:n_old]
    base.MODEL.ROI_BOX_HEAD.NAME_NEW_CLASSES = VOC_CLASSES[n_old:n_old + n_new]
    base.SOLVER.BASE_LR = base_lr
    base.SOLVER.WEIGHT_DECAY = 0.0001
    base.SOLVER.STEPS = (12500,)
    base.SOLVER.MAX_ITER = 15000
    base.SOLVER.IMS_PER_BATCH = ims_per_batch
    base.DIST.TYPE, base.DIST.FEAT = dist_type, feat
    base.DIST.ALPHA, base.DIST.BETA, base.DIST.GAMMA = alpha, beta, gamma
    base.INCREMENTAL = True
    if overrides:
        base.merge_from_list(list(overrides))
    cfg_source, cfg_target = base.clone(), base.clone()
    cfg_source.MODEL.ROI_BOX_HEAD.NUM_CLASSES = n_old + 1            # train_incremental.py:430-434
    cfg_target.MODEL.ROI_BOX_HEAD.NUM_CLASSES = n_old + n_new + 1    # :445-454
    return cfg_source, cfg_target


def randomize_frozen_bn(model, seed):
    """FrozenBN buffers default to identity (batch_norm.py:14-17); randomise them so the fused epilogue is exercised
    (w~U[.5,1.5], b~N(0,.1), mean~N(0,.1), var~U[.5,1.5]) -- SURVEY.md §8d."""
    from ..layers import FrozenBatchNorm2d
    g = torch.Generator().manual_seed(seed)
    for name, m in model.named_modules():
        if isinstance(m, FrozenBatchNorm2d):
            n = m.weight.numel()
            # A pretrained network's BN statistics keep activations O(1); a random-init one fed +-128 pixel values does not
            # and the first SGD step overflows.  Damp the stem (pixel scale) and every residual branch so that the synthetic
            # run trains at realistic magnitudes (same arithmetic, same kernels).
            damp = 1.0 / 64 if name.endswith("stem.bn1") else (0.25 if name.endswith("bn3") else 1.0)
            m.weight.copy_(((torch.rand(n, generator=g) + 0.5) * damp).to(m.weight.device))
            m.bias.copy_((torch.randn(n, generator=g) * 0.1).to(m.weight.device))
            m.running_mean.copy_((torch.randn(n, generator=g) * 0.1).to(m.weight.device))
            m.running_var.copy_((torch.rand(n, generator=g) + 0.5).to(m.weight.device))
            m.invalidate()


def build_models(cfg_source, cfg_target, seed=0, need_source=True):
    """Seeded random init; the target starts from the source's weights with the old-class rows of cls_score / bbox_pred copied
    into its larger head (utils/model_serialization.py:47-55), as loading model_trimmed.pth does in the reference."""
    torch.manual_seed(seed)
    model_target = build_detection_model(cfg_target)
    with torch.no_grad():
        randomize_frozen_bn(model_target, seed + 1)
    model_source = None
    if need_source:
        model_source = build_detection_model(cfg_source)
        with torch.no_grad():
            sd_t = dict(model_target.named_parameters())
            for name, p in model_source.named_parameters():
                q = sd_t[name]
                if p.shape == q.shape:
                    p.copy_(q)
                else:  # grown head: source holds the first rows
                    p.copy_(q[: p.shape[0]])
            bt = dict(model_target.named_buffers())
            for name, b in model_source.named_buffers():
                if name in bt and b.shape == bt[name].shape:
                    b.copy_(bt[name])
            from ..layers import FrozenBatchNorm2d
            for m in model_source.modules():
                if isinstance(m, FrozenBatchNorm2d):
                    m.invalidate()
        model_source.eval()
    model_target.train()
    return model_source, model_target


def synthetic_batch(batch, height=600, width=1000, seed=42, label_range=(16, 21), device="cuda", max_boxes=5):
    """images: uint8-valued U[0,255] BGR minus PIXEL_MEAN (transforms.py:161-165 + defaults.py:56-60);
    targets: 1..max_boxes GT boxes per image, w,h log-uniform in [32,480], labels over the task's NEW class ids."""
    g = torch.Generator().manual_seed(seed)
    mean = torch.tensor([102.9801, 115.9465, 122.7717]).view(1, 3, 1, 1)
    images = torch.randint(0, 256, (batch, 3, height, width), generator=g).float() - mean
    targets = []
    for _ in range(batch):
        n = int(torch.randint(1, max_boxes + 1, (1,), generator=g))
        w = torch.exp(torch.rand(n, generator=g) * (torch.log(torch.tensor(480.0)) - torch.log(torch.tensor(32.0))) + torch.log(torch.tensor(32.0)))
        h = torch.exp(torch.rand(n, generator=g) * (torch.log(torch.tensor(480.0)) - torch.log(torch.tensor(32.0))) + torch.log(torch.tensor(32.0)))
        x1 = torch.rand(n, generator=g) * (width - 33)
        y1 = torch.rand(n, generator=g) * (height - 33)
        boxes = torch.stack([x1, y1, (x1 + w).clamp(max=width - 1), (y1 + h).clamp(max=height - 1)], 1)
        labels = torch.randint(label_range[0], label_range[1], (n,), generator=g)
        t = BoxList(boxes.to(device), (width, height), mode="xyxy")
        t.add_field("labels", labels.to(device))
        targets.append(t)
    return images.to(device), targets
