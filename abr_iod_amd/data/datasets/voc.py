"""PASCAL VOC in the class-incremental protocol (mirror of maskrcnn_benchmark/data/datasets/voc_abr.py:25-300, 300-510: the
dataset side of SURVEY.md §8f row F1).

What the reference's PascalVOCDataset / PascalVOCDataset_ABR decide, kept here:
  * which images a task sees: the per-class `ImageSets/Main/<class>_<split>.txt` lists of the NEW classes (training) or of new + old
    classes (testing); an entry `<id> -1` is skipped, a difficult-only entry `<id>  0` is skipped in training and kept in testing;
    first occurrence wins, order preserved (voc_abr.py:84-162);
  * which boxes an image contributes: `difficult` objects are dropped unless use_difficult, excluded classes always, OLD classes in
    training (their annotations are what incremental learning withholds), coordinates shifted to 0-based (voc_abr.py:239-287);
  * training samples go through Augmented Box Replay and the transforms.
Decoding (PIL) and XML parsing are host I/O; from the decoded uint8 image on, pixels live on the device (data/abr.py)."""
import os
import xml.etree.ElementTree as ET

import torch

from ...structures.bounding_box import BoxList
from ..gpu_transforms import to_device_u8

CLASSES = ("__background__ ", "aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow", "diningtable",
           "dog", "horse", "motorbike", "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor")


class PascalVOCDataset(object):
    CLASSES = CLASSES

    def __init__(self, data_dir, split, use_difficult=False, transforms=None, old_classes=(), new_classes=(), excluded_classes=(),
                 is_train=True, abr=None, device="cuda"):
        self.root, self.image_set, self.keep_difficult = data_dir, split, use_difficult
        self.transforms, self.abr, self.device = transforms, abr, device
        self.old_classes, self.new_classes, self.exclude_classes = list(old_classes), list(new_classes), list(excluded_classes)
        self.is_train = is_train
        self._annopath = os.path.join(self.root, "Annotations", "%s.xml")
        self._imgpath = os.path.join(self.root, "JPEGImages", "%s.jpg")
        self._imgsetpath = os.path.join(self.root, "ImageSets", "Main", "%s.txt")
        self.class_to_ind = dict(zip(CLASSES, range(len(CLASSES))))
        self.final_ids = self._image_ids(self.new_classes if is_train else self.new_classes + self.old_classes)
        self.id_to_img_map = dict(enumerate(self.final_ids))

    def _image_ids(self, categories):
        seen, out = set(), []
        for category in categories:
            with open(self._imgsetpath % "{0}_{1}".format(category, self.image_set)) as f:
                for line in f:
                    b = line.strip("\n").split(" ")   # "<id> -1" -> [id, '-1'];  "<id>  1" -> [id, '', '1'] (two blanks in VOC's files)
                    if b[1] == "-1":
                        continue
                    if b[2] == "0" and self.is_train:  # only difficult instances of the class
                        continue
                    if b[0] not in seen:
                        seen.add(b[0])
                        out.append(b[0])
        return out

    def __len__(self):
        return len(self.final_ids)

    def get_img_id(self, index):
        return self.final_ids[index]

    def map_class_id_to_class_name(self, class_id):
        return CLASSES[class_id]

    def get_img_info(self, index):
        size = ET.parse(self._annopath % self.final_ids[index]).getroot().find("size")
        return {"height": int(size.find("height").text), "width": int(size.find("width").text)}

    def _preprocess_annotation(self, root):
        boxes, labels, difficult_flags = [], [], []
        for obj in root.iter("object"):
            difficult = int(obj.find("difficult").text) == 1
            if difficult and not self.keep_difficult:
                continue
            name = obj.find("name").text.lower().strip()
            if name in self.exclude_classes or (self.is_train and name in self.old_classes):
                continue
            bb = obj.find("bndbox")
            boxes.append([int(bb.find(k).text) - 1 for k in ("xmin", "ymin", "xmax", "ymax")])  # 1-based pixels -> 0-based
            labels.append(self.class_to_ind[name])
            difficult_flags.append(difficult)
        size = root.find("size")
        return {"boxes": torch.tensor(boxes, dtype=torch.float32).reshape(-1, 4), "labels": torch.tensor(labels, dtype=torch.int64),
                "difficult": torch.tensor(difficult_flags, dtype=torch.bool), "im_info": (int(size.find("height").text), int(size.find("width").text))}

    def get_groundtruth(self, index):
        anno = self._preprocess_annotation(ET.parse(self._annopath % self.final_ids[index]).getroot())
        height, width = anno["im_info"]
        target = BoxList(anno["boxes"], (width, height), mode="xyxy")
        target.add_field("labels", anno["labels"])
        target.add_field("difficult", anno["difficult"])
        return target

    def __getitem__(self, index):
        """-> (uint8 device image after Resize, target, flip flag, image id / index): one element of GPUTransform.collate's input.
        Training with a rehearsal memory goes through Augmented Box Replay first (voc_abr.py:470-492)."""
        from PIL import Image
        img_id = self.final_ids[index]
        img = to_device_u8(Image.open(self._imgpath % img_id).convert("RGB"), self.device)
        target = self.get_groundtruth(index).clip_to_image(remove_empty=True)
        if self.is_train and self.abr is not None:
            img, target = self.abr.transform_current_data_with_ABR(img, target)
        if self.transforms is not None:
            img, target, flip = self.transforms(img, target)
        else:
            flip = False
        return img, target, flip, (img_id if self.is_train and self.abr is not None else index)


class BatchCollator(object):
    """collate_batch.py:4-22 for device samples: zero-padded fp32 batch through the normalise kernel."""

    def __init__(self, transforms, size_divisible=0):
        self.transforms, self.size_divisible = transforms, size_divisible

    def __call__(self, batch):
        images, targets = self.transforms.collate([b[:3] for b in batch], self.size_divisible)
        return images, targets, [b[3] for b in batch]
