"""GPU: the ABR data path through the C-ABI kernels of csrc/imgproc.hip -- bit-exact against
(a) Pillow itself (resize), (b) the REFERENCE's PascalVOCDataset_ABR mixup / mosaic outputs (tests/golden/abr_data.npz, produced by
tests/golden/make_golden_abr.py from voc_abr.py with seeded RNGs) and (c) the oracle's numpy / torch-CPU restatements."""
import os
import random
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_resize_bit_exact_vs_pillow():
    from abr_iod_amd.data import gpu_transforms as G
    from oracle import abr_data_ref as R
    rs = np.random.RandomState(0)
    for H, W, ow, oh in [(37, 53, 80, 64), (200, 333, 91, 77), (120, 160, 120, 300), (64, 64, 64, 64), (375, 500, 800, 600),
                         (50, 70, 23, 70), (9, 7, 40, 3), (300, 220, 47, 31), (500, 375, 450, 600)]:
        img = rs.randint(0, 256, (H, W, 3), dtype=np.uint8)
        img[: H // 3] = 255
        for name in (G.BILINEAR, G.BICUBIC):
            got = G.resize(torch.from_numpy(img).cuda(), ow, oh, name).cpu().numpy()
            ref = R.pil_resize(img, ow, oh, name)
            assert got.shape == ref.shape and np.array_equal(got, ref), (name, H, W, ow, oh)


def test_blend_copy_fill_vs_numpy_and_errors():
    from abr_iod_amd.data import gpu_transforms as G
    from oracle import abr_data_ref as R
    rs = np.random.RandomState(1)
    img = rs.randint(0, 256, (90, 120, 3), dtype=np.uint8)
    crop = rs.randint(0, 256, (40, 50, 3), dtype=np.uint8)
    for lam in (float(np.float32(0.2871)), 0.0, 1.0, float(np.float32(0.9999))):
        for (x0, y0, x1, y1, ox, oy) in [(10, 5, 60, 45, 0, 0), (0, 0, 30, 20, 20, 20), (100, 70, 120, 90, 0, 0)]:
            exp = R.blend_paste(img.copy(), crop, x0, y0, x1, y1, ox, oy, lam)
            got = G.blend_paste_(torch.from_numpy(img.copy()).cuda(), torch.from_numpy(crop).cuda(), x0, y0, x1, y1, ox, oy, lam).cpu().numpy()
            assert np.array_equal(got, exp), (lam, x0, y0)
    canvas = G.full_canvas(30, 40, 114, "cuda")
    G.copy_rect_(canvas, torch.from_numpy(crop).cuda(), 5, 3, 10, 20, 20, 15)
    exp = np.full((30, 40, 3), 114, np.uint8)
    exp[3:18, 5:25] = crop[20:35, 10:30]
    assert np.array_equal(canvas.cpu().numpy(), exp)
    with pytest.raises(RuntimeError):   # numpy would raise a shape mismatch; the kernel launcher refuses out-of-range rectangles
        G.blend_paste_(torch.from_numpy(img.copy()).cuda(), torch.from_numpy(crop).cuda(), 100, 70, 130, 90, 0, 0, 0.5)
    with pytest.raises(RuntimeError):
        G.copy_rect_(canvas, torch.from_numpy(crop).cuda(), 0, 0, 30, 30, 30, 15)


def _memory(g, tmp_path):
    from PIL import Image
    names = [str(n) for n in g["names"]]
    for n in names:
        Image.fromarray(g["crop_" + n]).save(os.path.join(str(tmp_path), n), format="PNG")  # lossless content under the .jpg name
    return names


def test_abr_mixup_mosaic_equal_reference(gold, tmp_path):
    """Same seeds, same rehearsal memory, same inputs -> the same replayed image (every pixel), boxes, labels and index pool as
    the reference's _start_mixup / _start_boxes_mosaic / transform_current_data_with_ABR."""
    from abr_iod_amd.data.abr import BoxRehearsalABR
    from abr_iod_amd.data.gpu_transforms import to_device_u8
    from abr_iod_amd.structures.bounding_box import BoxList
    g = gold("abr_data")
    names = _memory(g, tmp_path)
    seen = set()
    for tag in [str(c) for c in g["cases"]]:
        mode, seed = tag.split("_")[0], int(tag.split("_")[1])
        img = g["in_img_{}".format(seed)]
        H, W = img.shape[:2]
        target = BoxList(torch.from_numpy(g["in_boxes_{}".format(seed)].copy()), (W, H), mode="xyxy")
        target.add_field("labels", torch.from_numpy(g["in_labels_{}".format(seed)].copy()))
        abr = BoxRehearsalABR(str(tmp_path), names, batch_size=4, shuffle=False)
        random.seed(100 + seed); torch.manual_seed(100 + seed)
        dev = to_device_u8(img.copy())
        if mode == "mixup":
            oi, ot = abr._start_mixup(dev, target)
        elif mode == "mosaic":
            oi, ot = abr._start_boxes_mosaic((W, H))
        else:
            oi, ot = abr.transform_current_data_with_ABR(dev, target)
        exp = g[tag + "_img"]
        got = oi.cpu().numpy()
        assert got.shape == exp.shape, tag
        assert np.array_equal(got, exp), (tag, int((got != exp).sum()))
        assert tuple(ot.size) == tuple(int(v) for v in g[tag + "_size"]), tag
        np.testing.assert_array_equal(ot.bbox.cpu().numpy().astype(np.float64), g[tag + "_boxes"])
        np.testing.assert_array_equal(np.asarray(ot.get_field("labels").cpu().numpy(), dtype=np.float64), g[tag + "_labels"])
        assert abr.boxes_index == [int(v) for v in g[tag + "_pool"]], tag
        seen.add((mode, not np.array_equal(exp, img) if exp.shape == img.shape else True))
    assert ("mixup", True) in seen and ("mixup", False) in seen and ("abr", False) in seen  # blended, refused (one big object), untouched


def test_transform_and_collate_vs_pillow_torch():
    """Resize (BILINEAR, min 600 / max 1000) + flip + ToTensor + BGR255 + mean/std + zero-padded batch == the host pipeline."""
    from abr_iod_amd.data.abr import GPUTransform
    from abr_iod_amd.data.gpu_transforms import to_device_u8
    from abr_iod_amd.structures.bounding_box import BoxList
    from oracle import abr_data_ref as R
    cfg = types.SimpleNamespace(INPUT=types.SimpleNamespace(MIN_SIZE_TRAIN=(600,), MAX_SIZE_TRAIN=1000, MIN_SIZE_TEST=600, MAX_SIZE_TEST=1000,
                                                             FLIP_PROB_TRAIN=0.5, PIXEL_MEAN=[102.9801, 115.9465, 122.7717],
                                                             PIXEL_STD=[1.0, 1.0, 1.0], TO_BGR255=True, BRIGHTNESS=0.0, CONTRAST=0.0,
                                                             SATURATION=0.0, HUE=0.0))
    tf = GPUTransform(cfg, is_train=True)
    rs = np.random.RandomState(3)
    random.seed(7)
    samples, hosts = [], []
    for (H, W) in [(375, 500), (500, 333), (300, 300), (200, 640)]:
        img = rs.randint(0, 256, (H, W, 3), dtype=np.uint8)
        t = BoxList(torch.tensor([[10.0, 20.0, 110.0, 150.0]]), (W, H), mode="xyxy")
        t.add_field("labels", torch.tensor([17]))
        state = random.getstate()
        out_img, out_t, flip = tf(to_device_u8(img), t)
        random.setstate(state)                       # replay the host pipeline with the same draws
        oh, ow = tf.resize.get_size((W, H))
        flip_h = random.random() < 0.5
        assert flip == flip_h and tuple(out_img.shape[:2]) == (oh, ow)
        host = R.pil_resize(img, ow, oh, R.BILINEAR)
        assert np.array_equal(out_img.cpu().numpy(), host)
        exp_t = t.resize((ow, oh))
        if flip_h:
            exp_t = exp_t.transpose(0)
        assert torch.equal(out_t.bbox, exp_t.bbox) and out_t.size == (ow, oh)
        samples.append((out_img, out_t, flip))
        hosts.append(R.to_tensor_normalize(host, cfg.INPUT.PIXEL_MEAN, cfg.INPUT.PIXEL_STD, True, flip_h))
    assert any(s[2] for s in samples) and not all(s[2] for s in samples)
    images, targets = tf.collate(samples)
    HP, WP = max(h.shape[1] for h in hosts), max(h.shape[2] for h in hosts)
    assert tuple(images.tensors.shape) == (4, 3, HP, WP)
    for i, h in enumerate(hosts):
        exp = np.zeros((3, HP, WP), np.float32)
        exp[:, : h.shape[1], : h.shape[2]] = h
        assert np.array_equal(images.tensors[i].cpu().numpy(), exp), i
        assert tuple(images.image_sizes[i]) == (h.shape[1], h.shape[2])
    images32, _ = tf.collate(samples, size_divisible=32)
    assert images32.tensors.shape[2] % 32 == 0 and images32.tensors.shape[3] % 32 == 0


def test_color_jitter_ops_bit_exact_vs_pillow_restatement():
    """csrc/imgproc.hip's ColorJitter ops == the oracle's restatements of Pillow's ImageEnhance / HSV arithmetic (pinned against Pillow itself by
    tests/test_oracle_data.py): brightness / contrast / saturation for factors inside and outside [0, 1], hue shifts with wrap-around, saturated and
    black regions, a gray image (the hue of a gray pixel is 0 whatever the shift)."""
    from abr_iod_amd.data import gpu_transforms as G
    from oracle import abr_data_ref as R
    rs = np.random.RandomState(5)
    img = rs.randint(0, 256, (157, 211, 3), dtype=np.uint8)
    img[:20] = 255
    img[20:40] = 0
    img[40:60] = rs.randint(0, 256, (20, 211, 1), dtype=np.uint8)      # gray band
    fn = {"brightness": R.adjust_brightness, "contrast": R.adjust_contrast, "saturation": R.adjust_saturation, "hue": R.adjust_hue}
    for op, factors in (("brightness", (0.0, 0.3, 1.0, 1.37, 2.5)), ("contrast", (0.0, 0.45, 1.0, 1.8)), ("saturation", (0.0, 0.7, 1.0, 1.6, 3.0)),
                        ("hue", (-0.5, -0.23, 0.0, 0.004, 0.31, 0.5))):
        for f in factors:
            got = G.color_jitter_(G.to_device_u8(img), op, f).cpu().numpy()
            want = fn[op](img, f)
            assert np.array_equal(got, want), (op, f, int(np.abs(got.astype(int) - want.astype(int)).max()), int((got != want).sum()))
    # every colour of a 64^3 sub-cube through the hue path (the conversions' branches: which channel is the maximum, all six sectors)
    v = np.arange(0, 256, 4, dtype=np.uint8)
    cube = np.ascontiguousarray(np.stack(np.meshgrid(v, v, v, indexing="ij"), -1).reshape(512, 512, 3))
    for f in (0.0, 0.17, -0.4):
        assert np.array_equal(G.color_jitter_(G.to_device_u8(cube), "hue", f).cpu().numpy(), R.adjust_hue(cube, f)), f
    with pytest.raises(RuntimeError):
        G.color_jitter_(G.to_device_u8(img), "hue", 0.7)


def test_color_jitter_transform_draws_and_order():
    """data/abr.py::ColorJitter (transforms.py:132-150): factors drawn and ops shuffled as torchvision 0.2-0.4's get_params does, applied in that order;
    GPUTransform applies it before the resize, only in training, and is the identity (same tensor object, nothing drawn) at zero strengths."""
    from abr_iod_amd.data.abr import ColorJitter, GPUTransform
    from abr_iod_amd.data.gpu_transforms import to_device_u8
    from oracle import abr_data_ref as R
    rs = np.random.RandomState(9)
    img = rs.randint(0, 256, (120, 160, 3), dtype=np.uint8)
    cj = ColorJitter(0.4, 0.3, 0.5, 0.1)
    for seed in (1, 2, 3):
        random.seed(seed)
        want_ops = R.color_jitter_params(0.4, 0.3, 0.5, 0.1, random)
        random.seed(seed)
        dev = to_device_u8(img)
        out, _ = cj(dev, None)
        assert len(want_ops) == 4 and out is not dev and np.array_equal(dev.cpu().numpy(), img)     # the input image is not written
        assert np.array_equal(out.cpu().numpy(), R.color_jitter_apply(img, want_ops)), (seed, want_ops)
    base = dict(MIN_SIZE_TRAIN=(120,), MAX_SIZE_TRAIN=200, MIN_SIZE_TEST=120, MAX_SIZE_TEST=200, FLIP_PROB_TRAIN=0.0,
                PIXEL_MEAN=[102.9801, 115.9465, 122.7717], PIXEL_STD=[1.0, 1.0, 1.0], TO_BGR255=True)
    cfg0 = types.SimpleNamespace(INPUT=types.SimpleNamespace(BRIGHTNESS=0.0, CONTRAST=0.0, SATURATION=0.0, HUE=0.0, **base))
    cfg1 = types.SimpleNamespace(INPUT=types.SimpleNamespace(BRIGHTNESS=0.2, CONTRAST=0.0, SATURATION=0.0, HUE=0.05, **base))
    dev = to_device_u8(img)
    state = random.getstate()
    plain, _, _ = GPUTransform(cfg0, is_train=True)(dev, None)
    assert np.array_equal(plain.cpu().numpy(), img)                       # 120x160 is already the target size: untouched
    random.setstate(state)
    random.seed(11)
    ops_ = R.color_jitter_params(0.2, 0.0, 0.0, 0.05, random)
    random.seed(11)
    jit, _, _ = GPUTransform(cfg1, is_train=True)(dev, None)
    assert [n for n, _ in sorted(ops_)] == ["brightness", "hue"] and np.array_equal(jit.cpu().numpy(), R.color_jitter_apply(img, ops_))
    test_time, _, _ = GPUTransform(cfg1, is_train=False)(dev, None)       # build.py:15-21: no jitter at test time
    assert np.array_equal(test_time.cpu().numpy(), img)
    with pytest.raises(ValueError):
        ColorJitter(hue=0.6)
