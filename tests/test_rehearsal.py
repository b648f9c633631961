"""F2: prototype box selection.  CPU: the selection order equals the REFERENCE's Mem on the committed fixture
(tests/golden/rehearsal.npz), crops / file names / step bookkeeping behave as tools/extract_memory.py.  GPU: channel-mean kernel
and the feature-extraction pass."""
import os
import random
import types

import numpy as np
import pytest
import torch

from abr_iod_amd.rehearsal.extract_memory import Mem


def _cfg(mem_type, mem_size=18, old=("a",), new=("b", "c", "d"), source_weight=""):
    return types.SimpleNamespace(MODEL=types.SimpleNamespace(ROI_BOX_HEAD=types.SimpleNamespace(
        NAME_OLD_CLASSES=list(old), NAME_NEW_CLASSES=list(new)), SOURCE_WEIGHT=source_weight), MEM_TYPE=mem_type, MEM_BUFF=mem_size,
        TASK="t", NAME="n")


def _info(g):
    counts = g["counts"]
    return [[{"feature": g[f"feat{c}"][j].tolist(), "logits": None, "image_path": ["x"], "box_class": 2 + c, "box": [0, 0, 99, 99],
              "mode": "xyxy", "rid": j} for j in range(n)] for c, n in enumerate(counts)]


@pytest.mark.parametrize("mem_type", ["mean", "random"])
def test_selection_equals_reference(gold, tmp_path, mem_type):
    g = gold("rehearsal")
    m = Mem(_cfg(mem_type), 0, str(tmp_path))
    picked = []

    def rec(self, info, ind):
        picked.append((info["box_class"], ind, info["rid"]))
        open(os.path.join(str(tmp_path), "{}_{:05d}_{}.jpg".format(info["box_class"], ind, len(picked))), "w").close()
    m.creat_and_save_box_image = types.MethodType(rec, m)
    random.seed(3)
    with pytest.raises(AssertionError):  # 15 files for an 18-slot memory: the old class's share is not there in this fixture
        m.update_memory(_info(g))
    assert np.array_equal(np.array(picked), g[f"{mem_type}_picked"])


def test_herding_is_greedy_mean_matching():
    rs = np.random.RandomState(0)
    fea = np.abs(rs.randn(30, 7, 7))
    m = Mem.__new__(Mem)
    order = m.herding_order(fea.tolist())
    assert sorted(order) == list(range(30))
    flat = fea.reshape(30, -1)
    mu = flat.mean(0) / np.linalg.norm(flat.mean(0))
    # the first pick is the single record closest to the normalised class mean; every prefix mean is the best greedy extension
    assert order[0] == int(((flat - mu) ** 2).sum(1).argmin())
    c1 = flat[order[0]]
    best2 = min((j for j in range(30) if j != order[0]), key=lambda j: (((c1 + flat[j]) / 2 - mu) ** 2).sum())
    assert order[1] == best2


def test_crops_names_and_step_bookkeeping(tmp_path):
    from PIL import Image
    root = tmp_path / "VOC2007"
    (root / "JPEGImages").mkdir(parents=True)
    rs = np.random.RandomState(1)
    Image.fromarray(rs.randint(0, 255, (120, 160, 3), dtype=np.uint8)).save(str(root / "JPEGImages" / "000007.jpg"), quality=95)
    # step 0: 2 classes (1 old + 1 new), 6 slots -> 3 per class
    d0 = tmp_path / "src" / "mean_6"
    d0.mkdir(parents=True)
    for k in range(3):
        (d0 / "1_{:05d}.jpg".format(k)).write_bytes(b"old")
    rec = lambda j, box: {"feature": (np.ones((7, 7)) * (j + 1)).tolist(), "logits": None, "image_path": ["000007"], "box_class": 2,
                          "box": box, "mode": "xyxy"}
    info = [[rec(0, [10.9, 20.2, 90.7, 100.99]), rec(1, [0.0, 0.0, 159.0, 119.0]), rec(2, [5.5, 5.5, 80.5, 80.5]), rec(3, [1, 2, 75, 76])]]
    m = Mem(_cfg("mean", 6, old=("a",), new=("b",)), 0, str(d0), image_root=str(root))
    files = m.update_memory(info)
    assert sorted(files) == ["1_00000.jpg", "1_00001.jpg", "1_00002.jpg", "2_00000.jpg", "2_00001.jpg", "2_00002.jpg"]
    sizes = sorted(Image.open(str(d0 / f)).size for f in files if f.startswith("2_"))
    assert set(sizes) <= {(80, 80), (159, 119), (75, 75), (74, 74)}  # int()-truncated corners: (10,20,90,100) -> 80x80 ...
    # the crop is exactly PIL's crop + default-quality JPEG of the same pixels
    im = Image.open(str(root / "JPEGImages" / "000007.jpg")).convert("RGB")
    which = [r for r in m.current_mem_info[0][:3]]
    ref = im.crop(tuple(int(v) for v in which[0]["box"]))
    ref.save(str(tmp_path / "ref.jpg"))
    assert (tmp_path / "ref.jpg").read_bytes() == (d0 / "2_00000.jpg").read_bytes()
    # step 1: 3 classes, 6 slots -> 2 per class: inherits indices 0..1 of every old class from the folder next to SOURCE_WEIGHT
    d1 = tmp_path / "step1" / "mean_6"
    d1.mkdir(parents=True)
    m1 = Mem(_cfg("mean", 6, old=("a", "b"), new=("c",), source_weight=str(tmp_path / "src" / "model_trimmed.pth")), 1, str(d1),
             image_root=str(root))
    info1 = [[dict(rec(j, [0, 0, 100, 100]), box_class=3) for j in range(5)]]
    files1 = m1.update_memory(info1)
    assert sorted(files1) == ["1_00000.jpg", "1_00001.jpg", "2_00000.jpg", "2_00001.jpg", "3_00000.jpg", "3_00001.jpg"]


@pytest.mark.gpu
def test_channel_mean_kernel():
    from abr_iod_amd import ops
    g = torch.Generator().manual_seed(0)
    for shape in ((37, 7, 7, 1024), (5, 49, 130), (3, 6)):
        x = torch.randn(shape, generator=g)
        got = ops.channel_mean(x.cuda()).cpu()
        np.testing.assert_allclose(got.numpy(), x.double().mean(-1).float().numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_extract_bboxes_and_features():
    """The GPU pass: per-GT-box 7x7 channel-mean maps equal torch.mean over the model's own pooled features; small boxes are
    dropped; records land in the bucket of their class (label - n_old - 1)."""
    from abr_iod_amd.engine.synthetic import build_models, make_cfgs, synthetic_batch
    from abr_iod_amd.rehearsal.prototype_box_selection import extract_bboxes_and_features
    tiny = ["MODEL.RESNETS.STEM_OUT_CHANNELS", 16, "MODEL.RESNETS.RES2_OUT_CHANNELS", 32, "MODEL.RESNETS.WIDTH_PER_GROUP", 8,
            "MODEL.RESNETS.BACKBONE_OUT_CHANNELS", 128]
    cfg_s, cfg_t = make_cfgs("15-5", dist_type="id", feat="ard", alpha=0.5, overrides=tiny)
    _, mt = build_models(cfg_s, cfg_t, seed=0, need_source=False)
    images, targets = synthetic_batch(2, 160, 224, seed=3)      # labels in 16..20 = the 5 new classes of task 15-5
    originals = [t.resize((448, 320)) for t in targets]          # image-scale boxes (2x)
    info = extract_bboxes_and_features(mt, [(images, targets, originals, [["a"], ["b"]])], torch.device("cuda"), cfg_t)
    assert len(info) == 5
    mt.eval()
    with torch.no_grad():
        (_, _), _, _, _, raf = mt.generate_feature_logits_by_targets(images, targets)
    ref = raf.float().mean(dim=1).cpu().numpy()
    row, seen = 0, 0
    for img_n, t in enumerate(originals):
        for box, lab in zip(t.bbox.cpu().tolist(), t.get_field("labels").cpu().tolist()):
            small = (box[2] - box[0]) <= 70 and (box[3] - box[1]) <= 70
            bucket = info[lab - 15 - 1]
            hit = [r for r in bucket if r["box"] == box]
            assert bool(hit) != small
            if hit:
                seen += 1
                np.testing.assert_allclose(np.array(hit[0]["feature"]), ref[row], rtol=1e-5, atol=1e-6)
                assert hit[0]["box_class"] == lab and hit[0]["image_path"] == [["a"], ["b"]][img_n] and hit[0]["mode"] == "xyxy"
            row += 1
    assert seen == sum(len(b) for b in info) > 0
