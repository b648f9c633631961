#!/usr/bin/env python3
"""F1 golden FROM THE REFERENCE (in-container only): the Augmented Box Replay methods of PascalVOCDataset_ABR
(maskrcnn_benchmark/data/datasets/voc_abr.py:512-838) run on synthetic images and a synthetic rehearsal memory with seeded RNGs.

The dataset object is created without its __init__ (which wants VOC on disk); only the attributes the ABR methods read are set.
voc_abr.py imports `Compose` through the data package at module import (a chain that needs torchvision, not installed here, and is
not used by any of the methods exercised), so that one package import is satisfied by an empty stand-in module.
Stored (tests/golden/abr_data.npz): the rehearsal crops (written as lossless PNG content under the reference's `<class>_<idx>.jpg`
names), the input images / targets, and for every case the reference's output image, boxes, labels and the index pool afterwards."""
import os
import random
import sys
import tempfile
import types

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402

rh.setup()
# voc_abr.py line 15 imports Compose through the data package, whose __init__ chain needs torchvision (not installed); none of the
# ABR methods use it, so the two package modules are pre-registered as empty stand-ins instead of being imported.
for n in ("maskrcnn_benchmark.data", "maskrcnn_benchmark.data.transforms"):
    sys.modules.setdefault(n, types.ModuleType(n))
sys.modules["maskrcnn_benchmark.data.transforms"].Compose = object
sys.path.insert(0, "/root/reference")

import importlib.util  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_voc_abr", "/root/reference/maskrcnn_benchmark/data/datasets/voc_abr.py")
voc_abr = importlib.util.module_from_spec(spec)
spec.loader.exec_module(voc_abr)
from maskrcnn_benchmark.structures.bounding_box import BoxList  # noqa: E402

CROPS = [("3", 0, 30, 44), ("3", 1, 95, 60), ("7", 0, 18, 22), ("7", 1, 130, 150), ("12", 0, 64, 40), ("12", 1, 52, 77),
         ("15", 0, 200, 90), ("15", 1, 75, 75)]  # (class, index, w, h)


def main():
    rs = np.random.RandomState(11)
    out = {}
    with tempfile.TemporaryDirectory() as d:
        names = []
        for k, (cls, idx, w, h) in enumerate(CROPS):
            arr = rs.randint(0, 256, (h, w, 3), dtype=np.uint8)
            arr[h // 4: h // 2] = 250  # flat bright band: bicubic overshoot / clipping gets exercised
            name = "{}_{:05d}.jpg".format(cls, idx)
            Image.fromarray(arr).save(os.path.join(d, name), format="PNG")
            out["crop_" + name] = arr
            names.append(name)
        out["names"] = np.array(names)
        cases = []
        for seed in range(6):
            H, W = [(120, 160), (150, 110), (96, 200)][seed % 3]
            img = rs.randint(0, 256, (H, W, 3), dtype=np.uint8)
            n = 1 + seed % 3
            x1 = rs.rand(n) * (W - 50); y1 = rs.rand(n) * (H - 50)
            boxes = np.stack([x1, y1, x1 + 15 + rs.rand(n) * 30, y1 + 15 + rs.rand(n) * 30], 1).astype(np.float32)
            if seed == 5:  # one object covering > 75 % of the image: mixup must leave the image alone
                boxes = np.array([[5., 5., W - 6., H - 6.]], np.float32)
            labels = rs.randint(16, 21, len(boxes))
            for mode in ("mixup", "mosaic", "abr"):
                ds = object.__new__(voc_abr.PascalVOCDataset_ABR)
                ds.PrototypeBoxSelection = types.SimpleNamespace(current_mem_path=d, first_mem_path=None)
                ds.BoxRehearsal_path = list(names)
                ds.boxes_index = list(range(len(names)))
                ds.batch_size, ds.bg_size = 4, 0
                target = BoxList(torch.from_numpy(boxes.copy()), (W, H), mode="xyxy")
                target.add_field("labels", torch.from_numpy(labels.copy()))
                random.seed(100 + seed); torch.manual_seed(100 + seed)
                pil = Image.fromarray(img.copy())
                if mode == "mixup":
                    oi, ot = ds._start_mixup(pil, target)
                elif mode == "mosaic":
                    oi, ot = ds._start_boxes_mosaic(pil, [], num_boxes=4)
                else:
                    oi, ot = ds.transform_current_data_with_ABR(pil, target)
                tag = "{}_{}".format(mode, seed)
                cases.append(tag)
                out[tag + "_img"] = np.asarray(oi)
                out[tag + "_boxes"] = ot.bbox.numpy().astype(np.float64)
                out[tag + "_labels"] = np.asarray(ot.get_field("labels").numpy(), dtype=np.float64)
                out[tag + "_size"] = np.array(ot.size)
                out[tag + "_pool"] = np.array(ds.boxes_index)
                print(tag, np.asarray(oi).shape, len(ot), ds.boxes_index)
            out["in_img_{}".format(seed)] = img
            out["in_boxes_{}".format(seed)] = boxes
            out["in_labels_{}".format(seed)] = labels
        out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "abr_data.npz"), **out)
    print("wrote abr_data.npz", os.path.getsize(os.path.join(HERE, "abr_data.npz")) / 1e6, "MB")


from mini_voc import make_mini_voc  # noqa: E402  (shared with the tests: no reference dependency)


def gold_voc_dataset():
    """Image lists and filtered ground truth of the reference's PascalVOCDataset (voc_abr.py:25-300) on the mini VOC directory, for
    training and testing, with / without difficult objects."""
    import json
    rs = np.random.RandomState(3)
    out = {}
    with tempfile.TemporaryDirectory() as d:
        make_mini_voc(d, rs)
        old, new, excl = ["dog", "person", "cat"], ["sofa", "train", "tvmonitor"], ["bird"]
        for tag, (is_train, split, diff) in {"train": (True, "trainval", False), "test": (False, "test", False),
                                             "test_difficult": (False, "test", True)}.items():
            ds = voc_abr.PascalVOCDataset(d, split, use_difficult=diff, transforms=None, old_classes=old, new_classes=new,
                                          excluded_classes=excl, is_train=is_train, is_father=True)
            rec = {"ids": list(ds.final_ids), "gt": []}
            for i in range(len(ds)):
                t = ds.get_groundtruth(i)
                rec["gt"].append({"boxes": t.bbox.tolist(), "labels": t.get_field("labels").tolist(),
                                  "difficult": [bool(v) for v in t.get_field("difficult").tolist()], "size": list(t.size),
                                  "info": ds.get_img_info(i)})
            out[tag] = rec
            print(tag, rec["ids"], [g["labels"] for g in rec["gt"]])
    with open(os.path.join(HERE, "voc_dataset.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
    gold_voc_dataset()
