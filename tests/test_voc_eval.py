"""F4: the VOC mAP of abr_iod_amd equals the reference's voc_eval.py on the committed fixture
(tests/golden/voc_eval.npz, produced by running the reference's eval_detection_voc: tests/golden/make_golden.py gold_voc_eval)."""
import numpy as np
import torch

from abr_iod_amd.data.datasets.evaluation import evaluate
from abr_iod_amd.data.datasets.evaluation.voc.voc_eval import calc_detection_voc_ap, eval_detection_voc
from abr_iod_amd.structures.bounding_box import BoxList


def _lists(g):
    preds, gts = [], []
    for i in range(int(g["n_images"])):
        size = tuple(int(v) for v in g[f"size{i}"])
        p = BoxList(torch.from_numpy(g[f"db{i}"]), size)
        p.add_field("labels", torch.from_numpy(g[f"dl{i}"])); p.add_field("scores", torch.from_numpy(g[f"ds{i}"]))
        t = BoxList(torch.from_numpy(g[f"gb{i}"]), size)
        t.add_field("labels", torch.from_numpy(g[f"gl{i}"])); t.add_field("difficult", torch.from_numpy(g[f"gd{i}"]))
        preds.append(p); gts.append(t)
    return preds, gts


def test_voc_map_equals_reference(gold):
    g = gold("voc_eval")
    preds, gts = _lists(g)
    for tag, m07 in (("area", False), ("voc07", True)):
        r = eval_detection_voc(preds, gts, iou_thresh=0.5, use_07_metric=m07)
        np.testing.assert_allclose(r["ap"], g[f"ap_{tag}"], rtol=0, atol=1e-12, equal_nan=True)
        assert abs(r["map"] - float(g[f"map_{tag}"])) < 1e-12


def test_ap_edge_cases():
    ap = calc_detection_voc_ap([None, np.array([1.0, 0.5, 2 / 3]), np.array([np.nan, 1.0])], [None, np.array([0.5, 0.5, 1.0]), None])
    assert np.isnan(ap[0]) and np.isnan(ap[2])
    assert abs(ap[1] - (0.5 * 1.0 + 0.5 * 2 / 3)) < 1e-12
    ap07 = calc_detection_voc_ap([np.array([1.0, 0.5])], [np.array([0.5, 0.5])], use_07_metric=True)
    assert abs(ap07[0] - 6 / 11) < 1e-12  # recall thresholds 0..0.5 see precision 1, the rest 0


class _FakeVOC(object):
    def __init__(self, gts, scale):
        self.gts, self.scale = gts, scale

    def get_img_info(self, i):
        return {"width": self.gts[i].size[0], "height": self.gts[i].size[1]}

    def get_groundtruth(self, i):
        return self.gts[i]

    def map_class_id_to_class_name(self, i):
        return "class{}".format(i)


def test_do_voc_evaluation_resizes_and_writes(gold, tmp_path, capsys):
    """predictions arrive at the network's input scale and are resized to the image's original size first (voc_eval.py:16-21)."""
    g = gold("voc_eval")
    preds, gts = _lists(g)
    scaled = [p.resize((p.size[0] * 2, p.size[1] * 2)) for p in preds]
    r = evaluate(_FakeVOC(gts, 2), scaled, str(tmp_path), box_only=False, iou_types=("bbox",))
    np.testing.assert_allclose(r["ap"], g["ap_area"], atol=1e-12, equal_nan=True)
    text = (tmp_path / "result.txt").read_text().splitlines()
    assert text[0] == "mAP: {:.4f}".format(float(g["map_area"]))
    assert text[1].startswith("class1") and len(text) == 22 and text[-1].startswith("nan,")
    assert "mAP" in capsys.readouterr().out
