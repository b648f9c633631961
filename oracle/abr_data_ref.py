"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the ABR data path's pixel arithmetic.

The reference (maskrcnn_benchmark/data/datasets/voc_abr.py:512-816, data/transforms/transforms.py:64-165) does its pixel work
with two third-party libraries: Pillow (`Image.resize`: BICUBIC by default for the box crops, voc_abr.py:548; BILINEAR through
torchvision's `F.resize`, transforms.py:99) and numpy (blend / paste).  Pillow is importable in this image and on the GPU box,
so it is the anchor: `resample_u8` below restates Pillow's antialiased separable resampler (src/libImaging/Resample.c, 8 bits
per channel: double-precision filter weights -> 22-bit fixed point -> integer convolution, horizontal pass first) and
tests/test_oracle_data.py checks it BIT-EXACT against `PIL.Image.resize`.  The blend / mosaic / normalise functions restate the
numpy / torch expressions of the reference line by line.
Imported by tests/ and tools/bench_data.py's CPU baseline only."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
BILINEAR, BICUBIC = "bilinear", "bicubic"


def _filter(name):
    if name == BILINEAR:
        def f(x):
            x = np.abs(x)
            return np.where(x < 1.0, 1.0 - x, 0.0)
        return f, 1.0
    if name == BICUBIC:
        a = -0.5

        def f(x):
            x = np.abs(x)
            return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1, np.where(x < 2.0, (((x - 5) * x + 8) * x - 4) * a, 0.0))
        return f, 2.0
    raise ValueError(name)


def precompute_coeffs(in_size, out_size, name):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the full-image box (in0 = 0, in1 = in_size).
    -> (bounds int32 [out,2] = (first input index, tap count), coeffs int32 [out,ksize], ksize)"""
    filt, support0 = _filter(name)
    scale = float(in_size) / out_size  # (double)(in1 - in0) / outSize with float in0/in1
    filterscale = max(scale, 1.0)
    support = support0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        x = np.arange(xmax, dtype=np.float64)
        w = filt((x + xmin - center + 0.5) * ss)
        ww = 0.0
        for v in w:          # sequential double accumulation, as the C loop
            ww += v
        if ww != 0.0:
            w = w / ww
        kk[xx, :xmax] = w
        bounds[xx] = (xmin, xmax)
    fixed = np.where(kk < 0, np.trunc(-0.5 + kk * (1 << PRECISION_BITS)), np.trunc(0.5 + kk * (1 << PRECISION_BITS))).astype(np.int32)
    return bounds, fixed, ksize


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resample_u8(img, out_w, out_h, name):
    """img uint8 [H,W,C] -> uint8 [out_h,out_w,C]; horizontal pass (if the width changes) then vertical pass, uint8 in between."""
    H, W, C = img.shape
    cur = img
    if out_w != W:
        b, k, _ = precompute_coeffs(W, out_w, name)
        out = np.empty((H, out_w, C), np.uint8)
        for xx in range(out_w):
            x0, n = b[xx]
            acc = (cur[:, x0:x0 + n, :].astype(np.int64) * k[xx, :n].astype(np.int64)[None, :, None]).sum(1) + (1 << (PRECISION_BITS - 1))
            out[:, xx, :] = _clip8(acc)
        cur = out
    if out_h != H:
        b, k, _ = precompute_coeffs(H, out_h, name)
        out = np.empty((out_h, cur.shape[1], C), np.uint8)
        for yy in range(out_h):
            y0, n = b[yy]
            acc = (cur[y0:y0 + n].astype(np.int64) * k[yy, :n].astype(np.int64)[:, None, None]).sum(0) + (1 << (PRECISION_BITS - 1))
            out[yy] = _clip8(acc)
        cur = out
    return cur


def pil_resize(img, out_w, out_h, name):
    """The dependency itself: what the reference calls."""
    from PIL import Image
    res = {BILINEAR: Image.BILINEAR, BICUBIC: Image.BICUBIC}[name]
    return np.asarray(Image.fromarray(img).resize((out_w, out_h), res))


def blend_paste(image, crop, x0, y0, x1, y1, off_x, off_y, lam):
    """voc_abr.py:664-683: image[y0:y1, x0:x1] = Lambda * image[...] + (1 - Lambda) * crop[off_y:off_y+h, off_x:off_x+w], float64
    arithmetic assigned into the uint8 array (C cast = truncation).  In place; returns image."""
    h, w = y1 - y0, x1 - x0
    img1 = lam * image[y0:y1, x0:x1]
    c = (1 - lam) * crop
    image[y0:y1, x0:x1] = img1 + c[off_y:off_y + h, off_x:off_x + w]
    return image


def mosaic_canvas(size, pastes):
    """voc_abr.py:739-765: float32 canvas filled with 114, each tile img4[y1a:y2a, x1a:x2a] = crop[y1b:y2b, x1b:x2b], np.uint8 at the
    end.  pastes: list of (crop uint8, (x1a,y1a,x2a,y2a), (x1b,y1b,x2b,y2b))."""
    img4 = np.full((size, size, 3), 114.0, dtype=np.float32)
    for crop, (x1a, y1a, x2a, y2a), (x1b, y1b, x2b, y2b) in pastes:
        img4[y1a:y2a, x1a:x2a] = crop[y1b:y2b, x1b:x2b]
    return np.uint8(img4)


def to_tensor_normalize(img, mean, std, to_bgr255=True, flip=False):
    """transforms.py:129-165 on a uint8 HWC RGB image: (optional hflip), ToTensor (/255 in fp32), [2,1,0] * 255, (x - mean) / std
    -> float32 [3,H,W]."""
    import torch
    if flip:
        img = img[:, ::-1]
    t = torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    if to_bgr255:
        t = t[[2, 1, 0]] * 255
    mean_t = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
    std_t = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)
    return ((t - mean_t) / std_t).numpy()


# ---------------------------------------------------------------------------------------------------------------------------------
# ColorJitter (maskrcnn_benchmark/data/transforms/transforms.py:132-150 -> torchvision.transforms.ColorJitter on PIL images: the first
# transform of build.py:28-36, on the ORIGINAL-size image).  torchvision is not importable in this image; its PIL path is four calls into
# Pillow -- ImageEnhance.Brightness / Contrast / Color (= Image.blend with a degenerate image) and an RGB -> HSV -> RGB round trip with the
# hue byte shifted -- and Pillow IS here, so the functions below restate Pillow's C arithmetic (src/libImaging/Blend.c, Convert.c::rgb2l,
# rgb2hsv_row, hsv2rgb_row) and tests/test_oracle_data.py pins them bit-exact against Pillow itself (the conversions over all 2^24 colours).
def _blend_u8(in1, in2, alpha):
    """Image.blend(im1, im2, alpha): float32 arithmetic, truncation; outside [0, 1] the result is clipped first (Blend.c)"""
    al = np.float32(alpha)
    t = in1.astype(np.float32) + al * (in2.astype(np.float32) - in1.astype(np.float32))
    if 0.0 <= alpha <= 1.0:
        return t.astype(np.int32).astype(np.uint8)
    return np.where(t <= 0, 0, np.where(t >= 255, 255, t.astype(np.int32))).astype(np.uint8)


def rgb_to_l(img):
    """Image.convert("L"): ITU-R 601-2 luma in 16-bit fixed point (Convert.c rgb2l)"""
    r, g, b = (img[..., c].astype(np.int64) for c in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def adjust_brightness(img, factor):
    """F.adjust_brightness: ImageEnhance.Brightness(img).enhance(factor) = blend(black, img, factor)"""
    return _blend_u8(np.zeros_like(img), img, factor)


def adjust_contrast(img, factor):
    """F.adjust_contrast: the degenerate image is the constant int(mean(L) + 0.5)"""
    mean = int(float(rgb_to_l(img).astype(np.float64).sum() / (img.shape[0] * img.shape[1])) + 0.5)
    return _blend_u8(np.full_like(img, mean), img, factor)


def adjust_saturation(img, factor):
    """F.adjust_saturation: ImageEnhance.Color, the degenerate image is convert("L").convert("RGB")"""
    return _blend_u8(np.repeat(rgb_to_l(img)[..., None], 3, axis=2), img, factor)


def rgb_to_hsv_u8(rgb):
    """Image.convert("HSV") (Convert.c rgb2hsv_row: float quotients, the hue offset and the fmod in double, truncation)"""
    r, g, b = (rgb[..., c].astype(np.int32) for c in range(3))
    maxc, minc = np.maximum(r, np.maximum(g, b)), np.minimum(r, np.minimum(g, b))
    cr = (maxc - minc).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        s = cr / maxc.astype(np.float32)
        rc, gc, bc = ((maxc - c).astype(np.float32) / cr for c in (r, g, b))
        h0 = (bc - gc).astype(np.float32)
        h1 = (2.0 + rc.astype(np.float64) - bc.astype(np.float64)).astype(np.float32)
        h2 = (4.0 + gc.astype(np.float64) - rc.astype(np.float64)).astype(np.float32)
        h = np.where(r == maxc, h0, np.where(g == maxc, h1, h2))
        hd = np.fmod(h.astype(np.float64) / 6.0 + 1.0, 1.0).astype(np.float32)
        uh = np.clip((hd.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
        us = np.clip((s.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    gray = minc == maxc
    return np.stack([np.where(gray, 0, uh), np.where(gray, 0, us), maxc], -1).astype(np.uint8)


def hsv_to_rgb_u8(hsv):
    """Image.convert("RGB") of an HSV image (Convert.c hsv2rgb_row: sector and remainder from (float)h * 6.0 / 255.0 in double, C round())"""
    h, s, v = hsv[..., 0].astype(np.float32), hsv[..., 1], hsv[..., 2]
    hh = h.astype(np.float64) * 6.0 / 255.0
    i = np.floor(hh).astype(np.int32)
    f = (hh - i.astype(np.float32).astype(np.float64)).astype(np.float32).astype(np.float64)
    fs = (s.astype(np.float32).astype(np.float64) / 255.0).astype(np.float32).astype(np.float64)
    vf = v.astype(np.float32).astype(np.float64)

    def c_round(x):   # half away from zero
        return np.where(x >= 0, np.floor(x + 0.5), np.ceil(x - 0.5)).astype(np.int32)
    p, q, t = (np.clip(c_round(x), 0, 255) for x in (vf * (1.0 - fs), vf * (1.0 - fs * f), vf * (1.0 - fs * (1.0 - f))))
    vi, k = v.astype(np.int32), i % 6
    R, G, B = np.choose(k, [vi, q, p, p, t, vi]), np.choose(k, [t, vi, vi, q, p, p]), np.choose(k, [p, p, t, vi, vi, q])
    gray = s == 0
    return np.stack([np.where(gray, vi, R), np.where(gray, vi, G), np.where(gray, vi, B)], -1).astype(np.uint8)


def hue_shift_u8(hue_factor):
    """the byte torchvision adds to the H channel: uint8(hue_factor * 255) -- truncation toward zero, then modulo 256"""
    return int(hue_factor * 255) % 256


def adjust_hue(img, hue_factor):
    """F.adjust_hue on a PIL image: H += uint8(hue_factor * 255) with wrap-around, S and V untouched"""
    if not -0.5 <= hue_factor <= 0.5:
        raise ValueError("hue_factor ({}) is not in [-0.5, 0.5].".format(hue_factor))
    hsv = rgb_to_hsv_u8(img)
    hsv[..., 0] = (hsv[..., 0].astype(np.int32) + hue_shift_u8(hue_factor)).astype(np.uint8)
    return hsv_to_rgb_u8(hsv)


def color_jitter_params(brightness, contrast, saturation, hue, rng):
    """torchvision.transforms.ColorJitter of the reference's era (0.2-0.4: ColorJitter.get_params): the interval of each strength (None = off),
    one `random.uniform` per active op in the order brightness, contrast, saturation, hue, then `random.shuffle` of the op list.
    -> [(op name, factor)] in application order.  `rng`: the `random` module or a random.Random."""
    def interval(v, center=1.0, clip0=True):
        if v is None or v == 0:
            return None
        lo, hi = center - v, center + v
        return (max(lo, 0.0) if clip0 else lo, hi)
    spec = [("brightness", interval(brightness)), ("contrast", interval(contrast)), ("saturation", interval(saturation)),
            ("hue", interval(hue, 0.0, False))]
    ops_ = [(n, rng.uniform(iv[0], iv[1])) for n, iv in spec if iv is not None]
    rng.shuffle(ops_)
    return ops_


def color_jitter_apply(img, ops_):
    fn = {"brightness": adjust_brightness, "contrast": adjust_contrast, "saturation": adjust_saturation, "hue": adjust_hue}
    for name, factor in ops_:
        img = fn[name](img, factor)
    return img
