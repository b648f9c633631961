#!/usr/bin/env python3
"""Throughput of the ABR data path: device pipeline (abr_iod_amd/data) vs the host pipeline the reference runs
in its DataLoader workers (Pillow + numpy + torch CPU, the same calls the reference makes), on VOC-sized synthetic images
(375x500 -> 600x800) with a synthetic rehearsal memory.  One process, one CPU core for the host leg.  GPU box only.
Prints one JSON line: images/s for both, per-stage device times."""
import json
import os
import random
import sys
import tempfile
import time
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abr_iod_amd.data.abr import BoxRehearsalABR, GPUTransform  # noqa: E402
from abr_iod_amd.data.gpu_transforms import to_device_u8  # noqa: E402
from abr_iod_amd.structures.bounding_box import BoxList  # noqa: E402


def main():
    from PIL import Image
    n_img, batch = 64, 4
    rs = np.random.RandomState(0)
    cfg = types.SimpleNamespace(INPUT=types.SimpleNamespace(MIN_SIZE_TRAIN=(600,), MAX_SIZE_TRAIN=1000, FLIP_PROB_TRAIN=0.5,
                                                             PIXEL_MEAN=[102.9801, 115.9465, 122.7717], PIXEL_STD=[1.0, 1.0, 1.0],
                                                             TO_BGR255=True, BRIGHTNESS=0.0, CONTRAST=0.0, SATURATION=0.0, HUE=0.0))
    imgs = [rs.randint(0, 256, (375, 500, 3), dtype=np.uint8) for _ in range(n_img)]
    with tempfile.TemporaryDirectory() as d:
        names = []
        for k in range(200):
            w, h = int(rs.randint(60, 300)), int(rs.randint(60, 300))
            name = "{}_{:05d}.jpg".format(1 + k % 15, k // 15)
            Image.fromarray(rs.randint(0, 256, (h, w, 3), dtype=np.uint8)).save(os.path.join(d, name), format="PNG")
            names.append(name)
        abr = BoxRehearsalABR(d, names, batch_size=batch, shuffle=False)
        tf = GPUTransform(cfg, True)

        def target():
            t = BoxList(torch.tensor([[30.0, 40.0, 180.0, 200.0], [250.0, 100.0, 400.0, 300.0]]), (500, 375), mode="xyxy")
            t.add_field("labels", torch.tensor([16, 18]))
            return t

        def device_pass():
            for b in range(0, n_img, batch):
                samples = []
                for k in range(batch):
                    img, t = abr.transform_current_data_with_ABR(to_device_u8(imgs[b + k]), target())
                    samples.append(tf(img, t))
                tf.collate(samples)
        random.seed(0); torch.manual_seed(0)
        device_pass()                      # warm-up: decodes + uploads the rehearsal crops, fills the coefficient cache
        torch.cuda.synchronize()
        random.seed(1); torch.manual_seed(1)
        t0 = time.perf_counter()
        device_pass()
        torch.cuda.synchronize()
        dt_dev = time.perf_counter() - t0

        # host leg: the same stages with Pillow / numpy / torch CPU (resize to 600x800 BILINEAR + ToTensor/normalise + pad; a BICUBIC
        # crop resize + float64 blend for the replayed quarter), single core like one DataLoader worker
        torch.set_num_threads(1)

        def pil_resize(a, w, h, res):
            return np.asarray(Image.fromarray(a).resize((w, h), res))

        def to_tensor_normalize(a, flip):
            if flip:
                a = a[:, ::-1]
            t = torch.from_numpy(np.ascontiguousarray(a)).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
            t = t[[2, 1, 0]] * 255
            return ((t - torch.tensor(cfg.INPUT.PIXEL_MEAN).view(-1, 1, 1)) / torch.tensor(cfg.INPUT.PIXEL_STD).view(-1, 1, 1)).numpy()

        crops = [np.asarray(Image.open(os.path.join(d, n)).convert("RGB")) for n in names[:32]]
        random.seed(1)
        t0 = time.perf_counter()
        for b in range(0, n_img, batch):
            outs = []
            for k in range(batch):
                im = imgs[b + k].copy()
                r = random.randint(0, 3)
                if r == 0:     # mixup: two crops
                    for c in crops[k:k + 2]:
                        c2 = pil_resize(c, 150, 150, Image.BICUBIC)
                        im[20:170, 20:170] = 0.3 * im[20:170, 20:170] + (1 - 0.3) * c2
                elif r == 1:   # mosaic: four crops on a 437-square canvas
                    canvas = np.full((437, 437, 3), 114.0, dtype=np.float32)
                    for q, c in enumerate(crops[k:k + 4]):
                        canvas[q // 2 * 218: q // 2 * 218 + 200, q % 2 * 218: q % 2 * 218 + 200] = pil_resize(c, 200, 200, Image.BICUBIC)
                    im = np.uint8(canvas)
                h, w = im.shape[:2]
                oh, ow = tf.resize.get_size((w, h))
                outs.append(to_tensor_normalize(pil_resize(im, ow, oh, Image.BILINEAR), k % 2 == 0))
            HP, WP = max(o.shape[1] for o in outs), max(o.shape[2] for o in outs)
            batch_t = torch.zeros((batch, 3, HP, WP))
            for slot, o in zip(batch_t, outs):
                slot[:, : o.shape[1], : o.shape[2]].copy_(torch.from_numpy(o))
        dt_host = time.perf_counter() - t0
    print(json.dumps({"metric": "ABR data path images/s (375x500 -> 600x800, MIX:MOS:NEW = 1:1:2)", "device_img_per_s": round(n_img / dt_dev, 1),
                      "host_1core_img_per_s": round(n_img / dt_host, 1), "images": n_img, "batch": batch,
                      "note": "device leg includes the H2D upload of each uint8 image; host leg = Pillow/numpy/torch CPU on one core"}))


if __name__ == "__main__":
    main()
