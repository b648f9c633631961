"""Filter-weight tables for `abr_img_resample_u8`: Pillow's `precompute_coeffs` + `normalize_coeffs_8bpc`
(src/libImaging/Resample.c of the Pillow the reference calls through `Image.resize`, voc_abr.py:548 / transforms.py:99), vectorised
over the output pixels.  Double precision on the host, in the same operation order as the C code (the weight sum is accumulated tap
by tap), so the 22-bit fixed-point weights -- and therefore every output pixel -- are identical to Pillow's."""
import functools
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
BILINEAR, BICUBIC = "bilinear", "bicubic"
_SUPPORT = {BILINEAR: 1.0, BICUBIC: 2.0}


def _weights(name, x):
    x = np.abs(x)
    if name == BILINEAR:
        return np.where(x < 1.0, 1.0 - x, 0.0)
    a = -0.5
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1, np.where(x < 2.0, (((x - 5) * x + 8) * x - 4) * a, 0.0))


@functools.lru_cache(maxsize=4096)
def coeff_table(in_size, out_size, name):
    """-> (bounds int32 [out,2] = (first input index, tap count), coeffs int32 [out,ksize], ksize)"""
    if name not in _SUPPORT:
        raise ValueError("unsupported resampling filter: {}".format(name))
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = _SUPPORT[name] * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)   # (int) cast truncates toward zero, then the clamp at 0
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    taps = np.arange(ksize, dtype=np.float64)[None, :]
    w = _weights(name, (taps + xmin[:, None] - center[:, None] + 0.5) * ss)
    w = np.where(taps < xmax[:, None], w, 0.0)
    ww = np.zeros(out_size, np.float64)
    for t in range(ksize):  # sequential accumulation in tap order, like the C loop
        ww = ww + w[:, t]
    w = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    fixed = np.where(w < 0, np.trunc(-0.5 + w * (1 << PRECISION_BITS)), np.trunc(0.5 + w * (1 << PRECISION_BITS))).astype(np.int32)
    bounds = np.stack([xmin, xmax], 1).astype(np.int32)
    return bounds, fixed, ksize
