"""F1 dataset side: the class-incremental VOC dataset selects the images and ground-truth boxes the REFERENCE's
PascalVOCDataset does (tests/golden/voc_dataset.json, produced by running voc_abr.py on tests/golden/mini_voc.py's directory)."""
import json
import os
import random
import sys
import types

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from mini_voc import make_mini_voc  # noqa: E402

OLD, NEW, EXCL = ["dog", "person", "cat"], ["sofa", "train", "tvmonitor"], ["bird"]


@pytest.fixture(scope="module")
def voc_dir(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("voc"))
    make_mini_voc(d, np.random.RandomState(3))
    return d


def test_image_lists_and_groundtruth_equal_reference(voc_dir):
    from abr_iod_amd.data.datasets.voc import PascalVOCDataset
    gold = json.load(open(os.path.join(HERE, "golden", "voc_dataset.json")))
    for tag, (is_train, split, diff) in {"train": (True, "trainval", False), "test": (False, "test", False),
                                         "test_difficult": (False, "test", True)}.items():
        ds = PascalVOCDataset(voc_dir, split, use_difficult=diff, old_classes=OLD, new_classes=NEW, excluded_classes=EXCL,
                              is_train=is_train, device="cpu")
        g = gold[tag]
        assert ds.final_ids == g["ids"] and len(ds) == len(g["ids"]), tag
        for i, rec in enumerate(g["gt"]):
            t = ds.get_groundtruth(i)
            assert t.bbox.tolist() == rec["boxes"] and t.get_field("labels").tolist() == rec["labels"], (tag, i)
            assert [bool(v) for v in t.get_field("difficult").tolist()] == rec["difficult"]
            assert list(t.size) == rec["size"] and ds.get_img_info(i) == rec["info"]
        assert ds.map_class_id_to_class_name(12) == "dog" and ds.get_img_id(0) == g["ids"][0]


@pytest.mark.gpu
def test_getitem_abr_transform_collate(voc_dir, tmp_path):
    """A training sample: decode -> device -> Augmented Box Replay -> Resize/flip -> zero-padded normalised batch.  Without a
    rehearsal memory and with flips off the batch equals the host pipeline (PIL BILINEAR + ToTensor/Normalize) exactly."""
    from PIL import Image
    from abr_iod_amd.data.abr import BoxRehearsalABR, GPUTransform
    from abr_iod_amd.data.datasets.voc import BatchCollator, PascalVOCDataset
    from oracle import abr_data_ref as R
    cfg = types.SimpleNamespace(INPUT=types.SimpleNamespace(MIN_SIZE_TRAIN=(240,), MAX_SIZE_TRAIN=400, MIN_SIZE_TEST=240, MAX_SIZE_TEST=400,
                                                             FLIP_PROB_TRAIN=0.0, PIXEL_MEAN=[102.9801, 115.9465, 122.7717],
                                                             PIXEL_STD=[1.0, 1.0, 1.0], TO_BGR255=True, BRIGHTNESS=0.0, CONTRAST=0.0,
                                                             SATURATION=0.0, HUE=0.0))
    tf = GPUTransform(cfg, is_train=True)
    ds = PascalVOCDataset(voc_dir, "trainval", old_classes=OLD, new_classes=NEW, excluded_classes=EXCL, is_train=True, transforms=tf)
    batch = [ds[i] for i in range(len(ds))]
    images, targets, ids = BatchCollator(tf)(batch)
    assert ids == [0, 1, 2] and images.tensors.shape[0] == 3
    for i, (img_dev, t, flip, _) in enumerate(batch):
        src = np.asarray(Image.open(os.path.join(voc_dir, "JPEGImages", ds.final_ids[i] + ".jpg")).convert("RGB"))
        oh, ow = tf.resize.get_size((src.shape[1], src.shape[0]))
        host = R.to_tensor_normalize(R.pil_resize(src, ow, oh, R.BILINEAR), cfg.INPUT.PIXEL_MEAN, cfg.INPUT.PIXEL_STD, True, False)
        got = images.tensors[i].cpu().numpy()
        assert np.array_equal(got[:, :oh, :ow], host) and not got[:, oh:, :].any() and not got[:, :, ow:].any()
        assert tuple(images.image_sizes[i]) == (oh, ow) and t.size == (ow, oh) and len(t) == len(ds.get_groundtruth(i))
    # with a rehearsal memory the sample goes through Augmented Box Replay (image id returned instead of the index)
    mem = tmp_path / "mem"
    mem.mkdir()
    rs = np.random.RandomState(0)
    names = []
    for k in range(6):
        n = "{}_{:05d}.jpg".format(12 + k % 2, k)
        Image.fromarray(rs.randint(0, 256, (40 + 7 * k, 50 + 5 * k, 3), dtype=np.uint8)).save(str(mem / n), format="PNG")
        names.append(n)
    random.seed(1); torch.manual_seed(1)
    ds2 = PascalVOCDataset(voc_dir, "trainval", old_classes=OLD, new_classes=NEW, excluded_classes=EXCL, is_train=True, transforms=tf,
                           abr=BoxRehearsalABR(str(mem), names, batch_size=4, shuffle=False))  # pool refills below 4, as with IMS_PER_BATCH 4
    kinds = set()
    for _ in range(4):
        for i in range(len(ds2)):
            img, t, flip, img_id = ds2[i]
            assert img_id == ds2.final_ids[i] and img.dtype == torch.uint8 and img.is_cuda and len(t) >= 1
            labs = set(int(v) for v in t.get_field("labels").tolist())
            kinds.add("replayed" if labs & {12, 13} else "plain")
    assert kinds == {"replayed", "plain"}
