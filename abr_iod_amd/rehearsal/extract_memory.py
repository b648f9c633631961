"""`Mem`: choose the prototype boxes of every new class and write their crops (mirror of tools/extract_memory.py:17-267).

Input: per new class a list of records {'feature': 7x7 channel-mean RoI map, 'logits', 'image_path', 'box_class', 'box', 'mode'}
(what prototype_box_selection.extract_bboxes_and_features collects).  Output: `<mem_type>_<mem_size>/<class>_<index:05d>.jpg`.
The selection arithmetic is a few thousand 49-element vectors per class: host numpy in the reference and here.  Behaviour kept
on purpose (each changes which boxes are chosen):
  Q1 mean sampling divides ALL feature maps of a class by one norm -- the Frobenius norm of the whole [N,7,7] stack -- while the
     class mean is normalised by its own norm (extract_memory.py:124-137);
  Q2 a class with fewer candidates than slots is topped up ONCE with its own first `deficit` records (:113-117);
  Q3 file index restarts at 0 for every class and the class id in the name is the integer label (:229).
Herding (:166-212) cannot run in the reference: `_ind_bbox_per_cls` is read before assignment (UnboundLocalError on the first
class).  It is implemented here as the code evidently intends (iCaRL herding on the un-normalised maps against the normalised
class mean) -- parity for `herding` is therefore UNPINNED; `mean` and `random` are pinned by tests/golden/rehearsal.npz."""
import math
import os
import random
import shutil

import numpy as np


class Mem(object):
    def __init__(self, cfg, step=0, current_mem_path=None, image_root="data/VOCdevkit/VOC2007"):
        self.new_classes = list(cfg.MODEL.ROI_BOX_HEAD.NAME_NEW_CLASSES)
        self.old_classes = list(cfg.MODEL.ROI_BOX_HEAD.NAME_OLD_CLASSES)
        self.all_classes = self.old_classes + self.new_classes
        self.cfg = cfg
        self.mem_type, self.mem_size, self.STEP = cfg.MEM_TYPE, cfg.MEM_BUFF, step
        self.root = image_root
        self._imgpath = os.path.join(self.root, "JPEGImages", "%s.jpg")
        self.current_mem_name = "{}_{}".format(self.mem_type, self.mem_size)
        self.current_mem_path = current_mem_path
        self.first_mem_path = None
        if step == 1:    # the first incremental step inherits the memory written next to the source checkpoint (:47-52)
            self.first_mem_path = os.path.join(os.path.split(cfg.MODEL.SOURCE_WEIGHT)[0], self.current_mem_name)
            self.exemplar = os.listdir(self.first_mem_path)
            assert len(self.exemplar) >= self.mem_size, "The selected rehearsals are not satisfied the setting size!"
        else:
            if step > 1:  # later steps update the run's own memory in place (:53-56)
                self.current_mem_path = os.path.join("output/{}/{}".format(cfg.TASK, cfg.NAME), self.current_mem_name)
            self.exemplar = os.listdir(self.current_mem_path)
        self.num_current_classes = len(self.new_classes)
        self.num_bbox_per_cls = math.ceil(self.mem_size / len(self.all_classes))
        self.current_mem_info, self.current_features, self.current_logits = [], [], []

    # ------------------------------------------------------------------ bookkeeping
    def get_fea_log_classes(self, mem_info):
        assert len(mem_info) == self.num_current_classes
        self.current_mem_info = mem_info
        return ([[r["feature"] for r in recs] for recs in mem_info], [[r["logits"] for r in recs] for recs in mem_info])

    def _top_up(self, i):
        """Q2"""
        deficit = self.num_bbox_per_cls - len(self.current_mem_info[i])
        if deficit > 0:
            for lst in (self.current_mem_info, self.current_features, self.current_logits):
                if lst and len(lst) > i:
                    lst[i].extend(lst[i][:deficit])

    def _write_class(self, i, order=None):
        recs = self.current_mem_info[i] if order is None else [self.current_mem_info[i][j] for j in order]
        self.current_mem_info[i] = recs
        for ind, rec in enumerate(recs[: self.num_bbox_per_cls]):  # Q3
            self.creat_and_save_box_image(rec, ind)

    def _done(self):
        files = os.listdir(self.current_mem_path)
        assert len(files) >= self.mem_size, "The selected rehearsals are not satisfied the setting size!"
        return files

    # ------------------------------------------------------------------ the three strategies
    def rnd_sampling(self):
        """:84-103 -- python `random.shuffle` per class (seed it for reproducibility), THEN the top-up"""
        for i in range(self.num_current_classes):
            random.shuffle(self.current_mem_info[i])
            short = self.num_bbox_per_cls - len(self.current_mem_info[i])
            if short > 0:
                self.current_mem_info[i].extend(self.current_mem_info[i][:short])
            self._write_class(i)
        return self._done()

    def mean_order(self, features):
        """Indices of one class's records by ascending distance to the class-mean map, Q1 normalisation (:119-140)."""
        fea = np.array(features)                       # [N,7,7] float64
        mu = fea.mean(axis=0)
        mu = mu / np.linalg.norm(mu)
        phi = fea / np.linalg.norm(fea)                # ONE norm for the whole stack
        return np.argsort(np.sqrt(((mu - phi) ** 2).sum(axis=(1, 2))))

    def mean_feature_sampling(self):
        for i in range(self.num_current_classes):
            self._top_up(i)
            self._write_class(i, self.mean_order(self.current_features[i])[: self.num_bbox_per_cls])
        return self._done()

    def herding_order(self, features):
        """Greedy herding (:177-199): pick, one at a time, the record that moves the running mean of the picks closest to the
        (normalised) class mean; already picked records are excluded."""
        fea = np.array(features)
        fea = fea.reshape(len(fea), -1)
        mu = fea.mean(axis=0)
        mu = mu / np.linalg.norm(mu)
        centre = np.zeros_like(mu)
        taken = np.zeros(len(fea), bool)
        order = []
        for f in range(len(fea)):
            cand = centre * f / (f + 1) + fea / (f + 1)
            d = ((cand - mu) ** 2).sum(axis=1)
            d[taken] = np.inf
            j = int(d.argmin())
            order.append(j)
            taken[j] = True
            centre = cand[j]
        return order

    def herding_feature_sampling(self):
        for i in range(self.num_current_classes):
            self._top_up(i)
            self._write_class(i, self.herding_order(self.current_features[i])[: self.num_bbox_per_cls])
        return self._done()

    # ------------------------------------------------------------------ output
    def creat_and_save_box_image(self, bbox_info, ind):
        """crop = PIL box with int()-truncated corners of the ORIGINAL-size box, saved as JPEG with PIL defaults (:222-230)"""
        from PIL import Image
        box = bbox_info["box"]
        im = Image.open(self._imgpath % bbox_info["image_path"][0]).convert("RGB")
        crop = im.crop((int(box[0]), int(box[1]), int(box[2]), int(box[3])))
        crop.save(os.path.join(self.current_mem_path, "{0}_{1:05d}.jpg".format(bbox_info["box_class"], ind)))

    def update_memory(self, input_bboxes_info):
        """:232-267 -- shrink / inherit the old classes' share, then select and write the new classes' boxes"""
        if self.STEP == 0 and input_bboxes_info is None:
            return
        keep_below = self.num_bbox_per_cls  # per-class quota after this step: indices 0 .. quota-1 stay
        if self.STEP == 1:
            for name in self.exemplar:
                src = os.path.join(self.first_mem_path, name)
                if os.path.isfile(src) and int(os.path.splitext(name)[0].split("_")[1]) < keep_below:
                    shutil.copy(src, self.current_mem_path)
        elif self.STEP > 1:
            for name in self.exemplar:
                path = os.path.join(self.current_mem_path, name)
                if os.path.isfile(path) and int(os.path.splitext(name)[0].split("_")[1]) >= keep_below:
                    os.remove(path)
        self.current_features, self.current_logits = self.get_fea_log_classes(input_bboxes_info)
        strategy = {"random": self.rnd_sampling, "mean": self.mean_feature_sampling, "herding": self.herding_feature_sampling}
        if self.mem_type in strategy:
            self.exemplar = strategy[self.mem_type]()
        return self.exemplar
