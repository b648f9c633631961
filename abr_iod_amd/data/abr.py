"""Augmented Box Replay on the device (mirror of maskrcnn_benchmark/data/datasets/voc_abr.py:512-832).

What the reference does per training sample on a DataLoader worker: with probability 1/4 blend ("mixup") up to two rehearsal box
crops into the current image at a spot that overlaps no ground-truth box, with probability 1/4 replace the image by a 4-tile
mosaic of box crops on a 114-grey canvas, otherwise keep it; then Resize / flip / ToTensor / Normalize and zero-padded batching.
All pixel work there is Pillow + numpy on the host.

Here the DECISIONS (which boxes, where, scale, Lambda) stay on the host -- a few dozen scalar operations drawing from the same
python `random` / torch RNG calls in the same order as the reference, so a seeded run makes the same plan -- and the PIXELS never
leave HBM: the current image is uploaded once as uint8 (0.5 MB), the rehearsal crops are decoded once and stay resident on the
device (2000 crops ~ 0.2 GB of 288 GB), and resize / blend / paste / normalise / pad are the kernels of csrc/imgproc.hip, each
bit-identical to the Pillow / numpy / torch-CPU original (tests/test_gpu_data.py)."""
import os
import random

import numpy as np
import torch

from ..structures.bounding_box import FLIP_LEFT_RIGHT, BoxList
from ..structures.image_list import ImageList
from . import gpu_transforms as G


def compute_overlap(a, b):
    """voc_abr.py:922-947: (intersection / area(b), intersection/area(a) > 0.3 or intersection/area(b) > 0.3), boxes with +1 extents"""
    area_b = (b[2] - b[0] + 1) * (b[3] - b[1] + 1)
    iw = max(min(a[2], b[2]) - max(a[0], b[0]) + 1, 0)
    ih = max(min(a[3], b[3]) - max(a[1], b[1]) + 1, 0)
    area_a = (a[2] - a[0] + 1) * (a[3] - a[1] + 1)
    inter = iw * ih
    return inter / area_b, bool(inter / area_a > 0.3 or inter / area_b > 0.3)


def _as_gts(targets):
    """BoxList -> float64 [n,5] (x1,y1,x2,y2,label), like the reference's list-of-lists -> np.array (voc_abr.py:570-578)"""
    if isinstance(targets, np.ndarray):
        return targets
    boxes = targets.bbox.tolist()
    labels = targets.get_field("labels").tolist()
    return np.array([b + [l] for b, l in zip(boxes, labels)]) if boxes else np.zeros((0, 5))


class BoxRehearsalABR(object):
    """The ABR part of PascalVOCDataset_ABR: owns the rehearsal memory (file list + index pool) and turns (image, target) into the
    replayed (image, target).  Images are uint8 [H,W,3] CUDA tensors."""

    def __init__(self, mem_dir, file_names, batch_size, device="cuda", bg_size=0, shuffle=True):
        self.mem_dir = mem_dir
        self.BoxRehearsal_path = list(file_names)
        if shuffle:
            random.shuffle(self.BoxRehearsal_path)          # voc_abr.py:397
        self.boxes_index = list(range(len(self.BoxRehearsal_path)))
        self.batch_size, self.bg_size, self.device = batch_size, bg_size, device
        self._crops = {}  # file name -> resident uint8 crop on the device

    # ------------------------------------------------------------------ rehearsal memory
    def _crop(self, name):
        t = self._crops.get(name)
        if t is None:
            from PIL import Image
            t = self._crops[name] = G.to_device_u8(Image.open(os.path.join(self.mem_dir, name)).convert("RGB"), self.device)
        return t

    def _refill(self):
        if len(self.boxes_index) < self.batch_size:          # :594-596 / :726-728
            self.boxes_index = list(range(len(self.BoxRehearsal_path)))

    def _sample_per_bbox_from_boxrehearsal(self, i, im_shape):
        """:512-553 -> (crop uint8 [h,w,3] on the device, np.array([[0,0,w,h,class]]), memory index).
        `im_shape` is whatever the caller has: the mixup path passes numpy's (H,W,3) -- so the 3 is averaged in -- the mosaic path
        PIL's (W,H)."""
        name = self.BoxRehearsal_path[self.boxes_index[i]]
        crop = self._crop(name)
        cls_name = os.path.splitext(name)[0].split("_")[0]
        o_h, o_w = crop.shape[0], crop.shape[1]
        im_mean = np.mean(im_shape)
        box_mean = np.mean(np.array([int(o_w), int(o_h)]))
        if float(im_mean * 0.2) <= float(box_mean) <= float(im_mean * 0.7):
            scale = 1.0
        else:
            scale = random.uniform(float(im_mean * 0.4), float(im_mean * 0.6)) / float(box_mean)
        crop = G.resize(crop, int(scale * o_w), int(scale * o_h), G.BICUBIC)   # Image.resize default filter
        return crop, np.array([[0, 0, crop.shape[1], crop.shape[0], int(cls_name)]]), self.boxes_index[i]

    # ------------------------------------------------------------------ mixup
    def _place(self, c_gt, gts, img_shape):
        """Find a spot for a crop of size c_gt that overlaps no ground-truth box: up to 10 draws in the top-left 60% x 40% of the
        image, then up to 10 anchored by the bottom-right corner, else give up (:608-643).  Returns (box, tries)."""
        H, W = img_shape[0], img_shape[1]
        cw, ch = c_gt[2] - c_gt[0], c_gt[3] - c_gt[1]

        def top_left():
            px, py = random.randint(0, int(W * 0.6)), random.randint(0, int(H * 0.4))
            return [c_gt[0] + px, c_gt[1] + py, c_gt[2] + px, c_gt[3] + py]
        box = top_left()
        tries = 0
        while len(gts) and tries < 20:
            if not any(compute_overlap(g, box)[1] for g in gts):
                break
            if tries < 10:
                box = top_left()
            else:
                px, py = random.randint(int(W * 0.4), W), random.randint(int(H * 0.6), H)
                box = [px - cw, py - ch, px, py]
            tries += 1
        return box, tries

    def _start_mixup(self, image, targets, alpha=2.0, beta=5.0):
        """:555-698.  image: uint8 [H,W,3] device tensor (modified in place and returned)."""
        img_shape = tuple(image.shape)
        H, W = img_shape[0], img_shape[1]
        gts = _as_gts(targets)
        do_mix = True
        if gts.shape[0] == 1:   # a single object filling > 75 % of both sides: leave the image alone (:583-587)
            bw, bh = gts[0][2] - gts[0][0], gts[0][3] - gts[0][1]
            if (W - bw) < (W * 0.25) and (H - bh) < (H * 0.25):
                do_mix = False
        if do_mix:
            lam = torch.distributions.beta.Beta(alpha, beta).sample().item()
            self._refill()
            for i in range(2):  # the reference loops `num_mixup = 3` but leaves after the second attempt (:690-692)
                crop, c_gt, b_id = self._sample_per_bbox_from_boxrehearsal(i, img_shape)
                box, tries = self._place(c_gt[0], gts, img_shape)
                if tries >= 20:
                    continue
                # clip to the image; a,b = overhang at the bottom / right, c,d = at the left / top (:645-662)
                a = b = c = d = 0
                if box[3] >= H:
                    a, box[3] = box[3] - H, H
                if box[2] >= W:
                    b, box[2] = box[2] - W, W
                if box[0] < 0:
                    c, box[0] = -box[0], 0
                if box[1] < 0:
                    d, box[1] = -box[1], 0
                # which part of the crop lands there: the reference offsets by c/d only when nothing hangs over at the bottom/right
                off_x, off_y = (c, d) if (a == 0 and b == 0) else (0, 0)
                G.blend_paste_(image, crop, box[0], box[1], box[2], box[3], off_x, off_y, lam)
                row = c_gt.copy().astype(np.float64)
                row[0][:4] = box
                gts = row if gts.shape[0] == 0 else np.insert(gts, 0, values=row, axis=0)
                if b_id in self.boxes_index:
                    self.boxes_index.remove(b_id)
        target = BoxList(torch.as_tensor(gts[:, :4], dtype=torch.float32), (W, H))
        target.add_field("labels", torch.tensor(gts[:, 4]))
        return image, target

    # ------------------------------------------------------------------ mosaic
    def _start_boxes_mosaic(self, size_wh, num_boxes=4):
        """:700-816 with `targets=[]` (how transform_current_data_with_ABR calls it): a square canvas of side mean(w,h) filled with
        114, four rehearsal crops around a centre drawn in the middle 40-60 %."""
        s = int(np.mean(size_wh))
        yc = int(random.uniform(s * 0.4, s * 0.6))
        xc = int(random.uniform(s * 0.4, s * 0.6))
        self._refill()
        tiles = [self._sample_per_bbox_from_boxrehearsal(i, size_wh) for i in range(num_boxes)]
        canvas = G.full_canvas(s, s, 114, self.device)
        bg = self.bg_size
        gt4 = []
        for i, (crop, gts, b_id) in enumerate(tiles):
            h, w = crop.shape[0], crop.shape[1]
            if i % 4 == 0:    # top right
                xc_, yc_ = xc + bg, yc - bg
                x1a, y1a, x2a, y2a = xc_, max(yc_ - h, 0), min(xc_ + w, s), yc_
                x1b, y1b, x2b, y2b = 0, h - (y2a - y1a), min(w, x2a - x1a), h
            elif i % 4 == 1:  # bottom left
                xc_, yc_ = xc - bg, yc + bg
                x1a, y1a, x2a, y2a = max(xc_ - w, 0), yc_, xc_, min(s, yc_ + h)
                x1b, y1b, x2b, y2b = w - (x2a - x1a), 0, max(xc_, w), min(y2a - y1a, h)
            elif i % 4 == 2:  # bottom right
                xc_, yc_ = xc + bg, yc + bg
                x1a, y1a, x2a, y2a = xc_, yc_, min(xc_ + w, s), min(s, yc_ + h)
                x1b, y1b, x2b, y2b = 0, 0, min(w, x2a - x1a), min(y2a - y1a, h)
            else:             # top left
                xc_, yc_ = xc - bg, yc - bg
                x1a, y1a, x2a, y2a = max(xc_ - w, 0), max(yc_ - h, 0), xc_, yc_
                x1b, y1b, x2b, y2b = w - (x2a - x1a), h - (y2a - y1a), w, h
            # numpy slices clamp at the array edge (x2b = max(xc_, w) can exceed w): the pasted rectangle is the target's size
            G.copy_rect_(canvas, crop, x1a, y1a, x1b, y1b, x2a - x1a, y2a - y1a)
            g = np.array(gts, dtype=np.float64)
            g[:, [0, 2]] += x1a - x1b
            g[:, [1, 3]] += y1a - y1b
            gt4.append(g)
            if b_id in self.boxes_index:
                self.boxes_index.remove(b_id)
        gt4 = np.concatenate(gt4, 0)
        gt4[:, 0] = np.clip(gt4[:, 0], 0, s); gt4[:, 2] = np.clip(gt4[:, 2], 0, s)
        gt4[:, 1] = np.clip(gt4[:, 1], 0, s); gt4[:, 3] = np.clip(gt4[:, 3], 0, s)
        keep = ~(((gt4[:, 2] - gt4[:, 0]) <= 2.0) | ((gt4[:, 3] - gt4[:, 1]) <= 2.0))   # "delete too small objects" (:795-800)
        gt4 = gt4[keep]
        target = BoxList(torch.as_tensor(gt4[:, :4], dtype=torch.float32), (s, s))
        target.add_field("labels", torch.tensor(gt4[:, 4]))
        return canvas, target

    # ------------------------------------------------------------------ entry point
    def transform_current_data_with_ABR(self, img, target):
        """MIX : MOS : NEW = 1 : 1 : 2 (:821-838)"""
        is_mosaic = is_mixup = False
        if random.randint(0, 1) == 0:
            if random.randint(0, 1) == 0:
                is_mixup = True
            else:
                is_mosaic = True
        if is_mosaic:
            return self._start_boxes_mosaic((img.shape[1], img.shape[0]))
        if is_mixup:
            return self._start_mixup(img, target)
        return img, target


# ---------------------------------------------------------------------------------------------------- transforms + collate
class Resize(object):
    """transforms.py:64-108; the image is resized on the device with Pillow's BILINEAR (what torchvision's F.resize calls)."""

    def __init__(self, min_size, max_size):
        self.min_size = tuple(min_size) if isinstance(min_size, (list, tuple)) else (min_size,)
        self.max_size = max_size

    def get_size(self, image_size):
        w, h = image_size
        size = random.choice(self.min_size)
        if self.max_size is not None:
            lo, hi = float(min(w, h)), float(max(w, h))
            if hi / lo * size > self.max_size:
                size = int(round(self.max_size * lo / hi))
        if (w <= h and w == size) or (h <= w and h == size):
            return (h, w)
        if w < h:
            return (int(size * h / w), size)
        return (size, int(size * w / h))

    def __call__(self, image, target):
        oh, ow = self.get_size((image.shape[1], image.shape[0]))
        image = G.resize(image, ow, oh, G.BILINEAR)
        return image, (target.resize((ow, oh)) if target is not None else None)


class ColorJitter(object):
    """transforms.py:132-150: torchvision.transforms.ColorJitter(brightness, contrast, saturation, hue) on a device image.  Strength v > 0 means a
    factor drawn uniformly from [max(0, 1 - v), 1 + v] (hue: [-v, v], v <= 0.5); the active ops are applied in a shuffled order.  The draws follow
    torchvision 0.2-0.4's `get_params` (the reference's era): one `random.uniform` per active op in the order brightness, contrast, saturation, hue,
    then one `random.shuffle` -- Python's `random`, like the flip.  The pixel work is csrc/imgproc.hip (bit-exact against Pillow's ImageEnhance /
    HSV conversions, which is what torchvision's PIL path calls).  All strengths 0 (every configs/voc YAML): the identity, nothing drawn."""

    def __init__(self, brightness=None, contrast=None, saturation=None, hue=None):
        def interval(v, name, center=1.0, bound=None, clip0=True):
            v = float(v or 0.0)
            if v < 0:
                raise ValueError("If {} is a single number, it must be non negative.".format(name))
            if v == 0:
                return None
            lo, hi = center - v, center + v
            if clip0:
                lo = max(lo, 0.0)
            if bound is not None and not (bound[0] <= lo <= hi <= bound[1]):
                raise ValueError("{} values should be between {}".format(name, bound))
            return (lo, hi)
        self.spec = [("brightness", interval(brightness, "brightness")), ("contrast", interval(contrast, "contrast")),
                     ("saturation", interval(saturation, "saturation")), ("hue", interval(hue, "hue", 0.0, (-0.5, 0.5), False))]

    def get_params(self):
        ops_ = [(n, random.uniform(iv[0], iv[1])) for n, iv in self.spec if iv is not None]
        random.shuffle(ops_)
        return ops_

    def __call__(self, image, target=None):
        ops_ = self.get_params()
        if ops_:
            image = image.clone()       # (the dataset may hand out a cached device image)
            for name, factor in ops_:
                G.color_jitter_(image, name, factor)
        return image, target


class GPUTransform(object):
    """build_transforms(cfg, is_train) (transforms/build.py:5-41) for device images: ColorJitter -> Resize -> RandomHorizontalFlip -> ToTensor ->
    Normalize.  The flip and the normalisation are deferred into the batching kernel, so this returns (resized uint8 image,
    target, flip flag); `collate` finishes the job.  ColorJitter is the identity in every configs/voc YAML (all four strengths 0) and at test time."""

    def __init__(self, cfg, is_train=True):
        j = [float(getattr(cfg.INPUT, k, 0.0) or 0.0) if is_train else 0.0 for k in ("BRIGHTNESS", "CONTRAST", "SATURATION", "HUE")]   # build.py:10-21
        self.color_jitter = ColorJitter(*j)
        self.resize = Resize(cfg.INPUT.MIN_SIZE_TRAIN if is_train else cfg.INPUT.MIN_SIZE_TEST,
                             cfg.INPUT.MAX_SIZE_TRAIN if is_train else cfg.INPUT.MAX_SIZE_TEST)
        self.flip_prob = getattr(cfg.INPUT, "FLIP_PROB_TRAIN", 0.5) if is_train else 0
        self.mean, self.std, self.to_bgr255 = cfg.INPUT.PIXEL_MEAN, cfg.INPUT.PIXEL_STD, cfg.INPUT.TO_BGR255

    def __call__(self, image, target):
        image, target = self.color_jitter(image, target)      # first, on the original-size image (build.py:28-30)
        image, target = self.resize(image, target)
        flip = random.random() < self.flip_prob
        if flip and target is not None:
            target = target.transpose(FLIP_LEFT_RIGHT)
        return image, target, flip

    def collate(self, samples, size_divisible=0):
        """samples: list of (uint8 image, target, flip) -> (ImageList with the [B,3,Hmax,Wmax] fp32 batch, targets)
        (collate_batch.py:13-22 + image_list.py:57-70)"""
        hs, ws = [s[0].shape[0] for s in samples], [s[0].shape[1] for s in samples]
        HP, WP = max(hs), max(ws)
        if size_divisible > 0:
            HP = -(-HP // size_divisible) * size_divisible
            WP = -(-WP // size_divisible) * size_divisible
        batch = torch.empty((len(samples), 3, HP, WP), dtype=torch.float32, device=samples[0][0].device)
        for slot, (img, _, flip) in zip(batch, samples):
            G.normalize_into(img, slot, self.mean, self.std, self.to_bgr255, flip)
        return ImageList(batch, [(h, w) for h, w in zip(hs, ws)]), [s[1] for s in samples]
