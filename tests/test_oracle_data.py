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
