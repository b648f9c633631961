"""CPU: the oracle restatement at FULL size against the reference itself (tests/golden/e2e_full_15-5.npz, written by
tests/golden/make_golden_e2e_full.py from /root/reference's own forward of two 600x1000 images through the full-width R50-C4,
= BASELINE.json configs[0]).  Weights are regenerated (seeded CPU init of this package + tests/e2e_common.perturb_trainable).
Pins the oracle the GPU tests compare with at the benchmark geometry: 38x63 C4 map, 35 910 anchors, 12000 -> 2000 proposals."""
import numpy as np
import torch

from e2e_common import CONFIGS, match_fraction, needs_source, perturb_trainable
from oracle import ops as O
from oracle import torch_ref as R
from oracle.model_ref import RefModel

H, W = 600, 1000


def regenerated_state_dicts(name, device="cpu"):
    from abr_iod_amd.engine.synthetic import build_models, make_cfgs
    from abr_iod_amd.utils.checkpoint import reference_state_dict
    task, dist_type, feat, alpha, beta, gamma, label_range, n_old = CONFIGS[name]
    cfg_s, cfg_t = make_cfgs(task, dist_type=dist_type, feat=feat, alpha=alpha, beta=beta, gamma=gamma, overrides=["MODEL.DEVICE", device])
    ms, mt = build_models(cfg_s, cfg_t, seed=0, need_source=needs_source(name))
    sd_t = {k: v.cpu() for k, v in reference_state_dict(mt).items()}
    perturb_trainable(sd_t, [n for n, p in mt.named_parameters() if p.requires_grad])
    sd_s = {k: v.cpu() for k, v in reference_state_dict(ms).items()} if ms is not None else None
    return sd_s, sd_t, ms, mt, cfg_s, cfg_t


def _close(a, b, tol=1e-4):
    return abs(a - b) <= tol * max(1.0, abs(b))


import pytest


@pytest.mark.parametrize("fixture", ["15-5", "15-5_b4"])
def test_oracle_reproduces_reference_full_size_step(gold, fixture):
    """(`15-5_b4`: the same configuration at B = 4, the benchmarked batch)"""
    from abr_iod_amd.engine.synthetic import synthetic_batch
    name = fixture.split("_b")[0]
    g = gold("e2e_full_" + fixture)
    NB = int(g["batch"]) if "batch" in g else 2
    _, dist_type, _, alpha, beta, gamma, label_range, n_old = CONFIGS[name]
    sd_s, sd_t, _, _, _, _ = regenerated_state_dicts(name)
    images, _ = synthetic_batch(NB, H, W, seed=int(g["image_seed"]), label_range=label_range, device="cpu")
    torch.set_num_threads(8)
    mt, ms = RefModel(sd_t, trainable_prefixes=()), RefModel(sd_s, trainable_prefixes=())
    with torch.no_grad():
        ft, fs = mt.backbone(images), ms.backbone(images)
        amax = float(g["feat_t_absmax"])
        np.testing.assert_allclose(ft[:, ::97, ::7, ::11].numpy(), g["feat_t_spot"], rtol=0, atol=1e-4 * amax)
        obj, reg = mt.rpn_head(ft)
        np.testing.assert_allclose(obj[:, :, ::5, ::9].numpy(), g["rpn_obj_spot"], rtol=0, atol=1e-4 * float(g["rpn_obj_absmax"]))
        fh, fw = ft.shape[-2:]
        assert (fh, fw) == (38, 63)
        anchors, vis = O.grid_anchors(O.cell_anchors(), fh, fw, 16, (H, W))
        assert anchors.shape[0] == 35910
        gts = [g[f"gt{i}"] for i in range(NB)]
        # proposals: 12000 -> NMS -> 2000 (+GT).  fp32 re-association (oneDNN here vs oneDNN there is identical; this guards the logic)
        props = R.rpn_post_process(obj, reg, [anchors] * NB, [(H, W)] * NB, 12000, 2000, gt_boxes=gts)
        for i in range(NB):   # as sets (e2e_common.match_fraction): the losses below use the reference's own lists
            assert abs(props[i][0].shape[0] - g[f"tgt_props{i}"].shape[0]) <= 2
            assert match_fraction(g[f"tgt_props{i}"], props[i][0]) >= 0.98, match_fraction(g[f"tgt_props{i}"], props[i][0])
        # RPN loss with the reference's sampler draw
        n = anchors.shape[0]
        labs, tgts, posm, negm = [], [], torch.zeros(NB, n, dtype=torch.bool), torch.zeros(NB, n, dtype=torch.bool)
        for i in range(NB):
            lab, tgt, _ = R.rpn_prepare_targets(anchors, vis, gts[i])
            labs.append(torch.from_numpy(lab)); tgts.append(torch.from_numpy(tgt))
            posm[i, torch.from_numpy(g[f"rpn_pos{i}"]).long()] = True
            negm[i, torch.from_numpy(g[f"rpn_neg{i}"]).long()] = True
        lo, lb = R.rpn_loss(obj, reg, torch.stack(labs), torch.stack(tgts), posm, negm)
        assert _close(float(lo), float(g["loss_objectness"])) and _close(float(lb), float(g["loss_rpn_box_reg"]))
        # box head on the reference's sampled proposals
        rois, labels, rts = [], [], []
        for i in range(NB):
            boxes = g[f"tgt_props{i}"]
            m = O.matcher(O.box_iou(gts[i], boxes), 0.5, 0.5, False)
            lab = g[f"gt_labels{i}"][np.clip(m, 0, None)].astype(np.int64)
            lab[m == -1] = 0; lab[m == -2] = -1
            tgt = O.box_encode(gts[i][np.clip(m, 0, None)], boxes, (10.0, 10.0, 5.0, 5.0))
            sel = g[f"head_sel{i}"].astype(np.int64)
            assert len(sel) == 512
            np.testing.assert_array_equal(lab[sel], g[f"det_labels{i}"].astype(np.int64))
            rois.append(np.concatenate([np.full((len(sel), 1), i, np.float32), boxes[sel]], 1))
            labels.append(lab[sel]); rts.append(tgt[sel])
        _, logits, boxreg = mt.box_head(ft, torch.from_numpy(np.concatenate(rois)))
        np.testing.assert_allclose(logits[:16].numpy(), g["det_logits_head"], rtol=1e-4, atol=1e-5)
        lc, lbox = R.box_head_loss(logits, boxreg, torch.from_numpy(np.concatenate(labels)), torch.from_numpy(np.concatenate(rts)), dist_type, n_old)
        assert _close(float(lc), float(g["loss_classifier"])) and _close(float(lbox), float(g["loss_box_reg"]))
        # distillation pass on the reference's 64 picks of its top-128
        rois64 = torch.from_numpy(np.concatenate([np.concatenate([np.full((64, 1), i, np.float32), g[f"src_top128_{i}"][g[f"soften_sel{i}"]]], 1)
                                                  for i in range(NB)]))
        ps, zs, bs = ms.box_head(fs, rois64)
        pt, zt, bt = mt.box_head(ft, rois64)
        np.testing.assert_allclose(zs[:8].numpy(), g["soften_scores_head"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(zt[:8].numpy(), g["target_scores_head"], rtol=1e-4, atol=1e-5)
        l_id = R.roi_distillation_loss(zs, bs.view(-1, n_old + 1, 4), zt, bt.view(-1, 21, 4), dist_type)
        l_ard = R.ard_loss(ps, pt, gamma)
        assert _close(float(l_id), float(g["loss_id"])) and _close(float(l_ard), float(g["loss_ard"]))
