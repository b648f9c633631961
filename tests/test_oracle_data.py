"""CPU: the oracle's restatement of Pillow's 8-bit resampler is bit-exact against PIL.Image.resize (the dependency the reference
calls: voc_abr.py:548 default BICUBIC, transforms.py:99 BILINEAR), so the HIP resampler can be checked against either."""
import numpy as np
import pytest

from oracle import abr_data_ref as R

CASES = [(37, 53, 80, 64), (200, 333, 91, 77), (120, 160, 120, 300), (64, 64, 64, 64), (375, 500, 600, 800), (50, 70, 23, 70),
         (9, 7, 40, 3), (300, 220, 47, 31)]


@pytest.mark.parametrize("name", [R.BILINEAR, R.BICUBIC])
def test_resample_restatement_equals_pillow(name):
    rs = np.random.RandomState(0)
    for H, W, ow, oh in CASES:
        img = rs.randint(0, 256, (H, W, 3), dtype=np.uint8)
        img[: H // 3] = 255  # saturated areas: bicubic overshoot must clip exactly like clip8
        img[H // 3: H // 2, : W // 2] = 0
        got = R.resample_u8(img, ow, oh, name)
        ref = R.pil_resize(img, ow, oh, name)
        assert got.shape == ref.shape
        assert np.array_equal(got, ref), (name, H, W, ow, oh, int(np.abs(got.astype(int) - ref.astype(int)).max()))


def test_blend_and_normalize_restatements():
    rs = np.random.RandomState(1)
    img = rs.randint(0, 256, (40, 50, 3), dtype=np.uint8)
    crop = rs.randint(0, 256, (20, 30, 3), dtype=np.uint8)
    lam = float(np.float32(0.2871))
    out = R.blend_paste(img.copy(), crop, 10, 5, 40, 25, 0, 0, lam)
    exp = np.floor(lam * img[5:25, 10:40].astype(np.float64) + (1 - lam) * crop.astype(np.float64)).astype(np.uint8)
    assert np.array_equal(out[5:25, 10:40], exp) and np.array_equal(out[:5], img[:5])
    t = R.to_tensor_normalize(img, [102.9801, 115.9465, 122.7717], [1.0, 1.0, 1.0])
    assert t.shape == (3, 40, 50) and abs(float(t[0, 0, 0]) - (float(img[0, 0, 2]) - 102.9801)) < 1e-4


def test_color_jitter_restatements_equal_pillow():
    """ColorJitter's four pixel ops (transforms.py:132-150 -> torchvision -> Pillow): the oracle's restatements against Pillow ITSELF --
    ImageEnhance.Brightness / Contrast / Color for factors inside and outside [0, 1], the RGB -> HSV -> RGB conversions over 2^21 colours of the
    2^24 (every 8th: the full cube was compared when the restatement was written, tests/golden note in oracle/abr_data_ref.py), hue shifts with wrap."""
    from PIL import Image, ImageEnhance
    rs = np.random.RandomState(2)
    img = rs.randint(0, 256, (61, 83, 3), dtype=np.uint8)
    img[:10] = 255
    img[10:20] = 0
    pil = Image.fromarray(img)
    for f in (0.0, 0.25, 0.6, 1.0, 1.4, 1.999, 0.123456789):
        assert np.array_equal(R.adjust_brightness(img, f), np.asarray(ImageEnhance.Brightness(pil).enhance(f))), f
        assert np.array_equal(R.adjust_contrast(img, f), np.asarray(ImageEnhance.Contrast(pil).enhance(f))), f
        assert np.array_equal(R.adjust_saturation(img, f), np.asarray(ImageEnhance.Color(pil).enhance(f))), f
    assert np.array_equal(R.rgb_to_l(img), np.asarray(pil.convert("L")))
    v = np.arange(0, 256, dtype=np.uint8)
    cube = np.stack(np.meshgrid(v, v[::2], v[::4], indexing="ij"), -1).reshape(256, -1, 3)
    assert np.array_equal(R.rgb_to_hsv_u8(cube), np.asarray(Image.fromarray(cube, "RGB").convert("HSV")))
    assert np.array_equal(R.hsv_to_rgb_u8(cube), np.asarray(Image.fromarray(cube, "HSV").convert("RGB")))
    for hf in (-0.5, -0.1, 0.0, 0.07, 0.5):
        h, s, vv = pil.convert("HSV").split()
        nh = (np.asarray(h).astype(np.int32) + int(hf * 255)).astype(np.uint8)           # torchvision: np_h += np.uint8(hue_factor * 255), wrapping
        want = np.asarray(Image.merge("HSV", (Image.fromarray(nh, "L"), s, vv)).convert("RGB"))
        assert np.array_equal(R.adjust_hue(img, hf), want), hf
    with pytest.raises(ValueError):
        R.adjust_hue(img, 0.6)


def test_color_jitter_parameter_draws():
    """the draw sequence of torchvision 0.2-0.4's ColorJitter.get_params (the reference's era): uniform per active op in the order b, c, s, h, then one
    shuffle -- restated from the published source (torchvision is not importable here: this half of the transform is NOT pinned by execution)"""
    import random
    rng = random.Random(7)
    ops_ = R.color_jitter_params(0.4, 0.0, 0.3, 0.1, rng)
    chk = random.Random(7)
    want = [("brightness", chk.uniform(0.6, 1.4)), ("saturation", chk.uniform(0.7, 1.3)), ("hue", chk.uniform(-0.1, 0.1))]
    chk.shuffle(want)
    assert ops_ == want and all(n != "contrast" for n, _ in ops_)
    assert R.color_jitter_params(0, 0, 0, 0, random.Random(1)) == []
    assert R.color_jitter_params(2.0, None, None, None, random.Random(1))[0][1] >= 0.0          # brightness interval clipped at 0
