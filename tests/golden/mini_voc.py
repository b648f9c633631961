"""A 6-image VOC-layout directory for the dataset tests and for tests/golden/make_golden_abr.py::gold_voc_dataset (which runs the
REFERENCE's PascalVOCDataset on it).  Pure numpy / PIL: importable without the reference tree."""
import os

import numpy as np


def make_mini_voc(root, rs):
    """A 6-image VOC-layout directory (Annotations / JPEGImages / ImageSets/Main) with old, new and excluded classes, difficult
    objects and the two-blank list format of the real dataset."""
    from PIL import Image as PILImage
    for sub in ("Annotations", "JPEGImages", os.path.join("ImageSets", "Main")):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    objs = {"000001": [("dog", 0, (10, 20, 100, 120)), ("person", 0, (30, 40, 90, 140))],
            "000002": [("sofa", 0, (5, 5, 150, 100)), ("dog", 1, (20, 30, 60, 80))],
            "000003": [("train", 0, (12, 14, 130, 90)), ("cat", 0, (40, 40, 80, 90)), ("tvmonitor", 0, (1, 1, 50, 50))],
            "000004": [("person", 0, (3, 4, 50, 110)), ("sheep", 1, (60, 20, 140, 100))],
            "000005": [("tvmonitor", 0, (50, 30, 120, 100)), ("sofa", 0, (10, 60, 159, 119)), ("bird", 0, (5, 5, 30, 30))],
            "000006": [("sheep", 0, (20, 20, 100, 100))]}
    for k, (img_id, ol) in enumerate(sorted(objs.items())):
        W, H = 160 + 8 * k, 120 + 4 * k
        PILImage.fromarray(rs.randint(0, 256, (H, W, 3), dtype=np.uint8)).save(os.path.join(root, "JPEGImages", img_id + ".jpg"))
        xml = ["<annotation><size><width>{}</width><height>{}</height><depth>3</depth></size>".format(W, H)]
        for name, diff, (x1, y1, x2, y2) in ol:
            xml.append("<object><name>{}</name><difficult>{}</difficult><bndbox><xmin>{}</xmin><ymin>{}</ymin><xmax>{}</xmax><ymax>{}</ymax>"
                       "</bndbox></object>".format(name, diff, x1, y1, x2, y2))
        xml.append("</annotation>")
        with open(os.path.join(root, "Annotations", img_id + ".xml"), "w") as f:
            f.write("".join(xml))
    for cls in ("dog", "person", "sofa", "train", "cat", "tvmonitor", "sheep", "bird"):
        for split in ("trainval", "test"):
            with open(os.path.join(root, "ImageSets", "Main", "{}_{}.txt".format(cls, split)), "w") as f:
                for img_id, ol in sorted(objs.items()):
                    mine = [d for n, d, _ in ol if n == cls]
                    if not mine:
                        f.write("{} -1\n".format(img_id))
                    elif all(mine):
                        f.write("{}  0\n".format(img_id))
                    else:
                        f.write("{}  1\n".format(img_id))
    return objs
