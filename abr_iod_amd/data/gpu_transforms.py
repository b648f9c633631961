"""Device-side image ops of the ABR data path (uint8 HWC RGB tensors on the GPU) over csrc/imgproc.hip.
Mirrors what maskrcnn_benchmark/data/transforms/transforms.py and voc_abr.py do with Pillow / numpy on the host."""
import ctypes as C

import numpy as np
import torch

from .. import _lib as L
from .resample import BICUBIC, BILINEAR, coeff_table  # noqa: F401

_tables = {}


def _dev_table(in_size, out_size, name, device):
    key = (in_size, out_size, name, device)
    t = _tables.get(key)
    if t is None:
        if len(_tables) > 8192:
            _tables.clear()
        b, k, ksize = coeff_table(in_size, out_size, name)
        t = _tables[key] = (torch.from_numpy(b).to(device), torch.from_numpy(k).to(device), ksize)
    return t


def to_device_u8(img, device="cuda"):
    """PIL image / numpy [H,W,3] uint8 -> device tensor (the one host->device copy of a sample: ~0.5 MB)"""
    arr = np.asarray(img)
    if arr.ndim != 3 or arr.shape[2] != 3 or arr.dtype != np.uint8:
        raise ValueError("expected an RGB uint8 image, got {} {}".format(arr.shape, arr.dtype))
    if not (arr.flags.writeable and arr.flags.c_contiguous):
        arr = np.array(arr, order="C")  # PIL hands out read-only views
    return torch.from_numpy(arr).to(device, non_blocking=True)


def resize(img, out_w, out_h, resample=BILINEAR):
    """`PIL.Image.resize((out_w, out_h), resample)` on a device image, bit-exact."""
    L.require_cuda(img)
    H, W, _ = img.shape
    out_w, out_h = int(out_w), int(out_h)
    out = torch.empty((out_h, out_w, 3), dtype=torch.uint8, device=img.device)
    nil = (None, None, 0)
    bh, kh, nh = _dev_table(W, out_w, resample, img.device) if out_w != W else nil
    bv, kv, nv = _dev_table(H, out_h, resample, img.device) if out_h != H else nil
    tmp = torch.empty((H, out_w, 3), dtype=torch.uint8, device=img.device) if (out_w != W and out_h != H) else None
    L.check(L.lib().abr_img_resample_u8(L.ptr(img), H, W, L.ptr(out), out_h, out_w, L.ptr(bh), L.ptr(kh), nh, L.ptr(bv), L.ptr(kv), nv,
                                        L.ptr(tmp), L.stream()), "img_resample")
    return out


def blend_paste_(img, crop, x0, y0, x1, y1, off_x, off_y, lam):
    """img[y0:y1, x0:x1] = uint8(lam*img[...] + (1-lam)*crop[off_y:.., off_x:..])  in place (voc_abr.py:664-683)"""
    H, W, _ = img.shape
    CH, CW, _ = crop.shape
    L.check(L.lib().abr_img_blend_paste_u8(L.ptr(img), H, W, L.ptr(crop), CH, CW, int(x0), int(y0), int(x1 - x0), int(y1 - y0), int(off_x),
                                           int(off_y), float(lam), L.stream()), "img_blend_paste")
    return img


def copy_rect_(dst, src, dx, dy, sx, sy, rw, rh):
    L.check(L.lib().abr_img_copy_rect_u8(L.ptr(dst), dst.shape[0], dst.shape[1], L.ptr(src), src.shape[0], src.shape[1], int(dx), int(dy),
                                         int(sx), int(sy), int(rw), int(rh), L.stream()), "img_copy_rect")
    return dst


def full_canvas(h, w, value, device):
    out = torch.empty((h, w, 3), dtype=torch.uint8, device=device)
    L.check(L.lib().abr_img_fill_u8(L.ptr(out), out.numel(), int(value), L.stream()), "img_fill")
    return out


def normalize_into(img, out_slot, mean, std, to_bgr255=True, flip=False):
    """(hflip) + ToTensor + to_bgr255 + Normalize of `img` [h,w,3] into `out_slot` = one [3,HP,WP] fp32 image of the batch tensor,
    zero outside [h,w] (transforms.py:108-165, image_list.py:57-70)."""
    h, w, _ = img.shape
    _, HP, WP = out_slot.shape
    m = (C.c_float * 3)(*[float(v) for v in mean])
    s = (C.c_float * 3)(*[float(v) for v in std])
    L.check(L.lib().abr_img_normalize_to_batch(L.ptr(img), h, w, int(bool(flip)), int(bool(to_bgr255)), m, s, L.ptr(out_slot), HP, WP,
                                               L.stream()), "img_normalize")
    return out_slot


_JITTER_OPS = {"brightness": 0, "contrast": 1, "saturation": 2, "hue": 3}
_jitter_scratch = {}


def color_jitter_(img, op, factor):
    """One ColorJitter op of torchvision's PIL path, in place on a device image (uint8 HWC RGB), bit-exact against Pillow:
    F.adjust_brightness / adjust_contrast / adjust_saturation (= ImageEnhance: Image.blend with a degenerate image) and F.adjust_hue
    (H byte shifted through Pillow's RGB <-> HSV conversions).  op: a name of _JITTER_OPS."""
    L.require_cuda(img)
    if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3 or not img.is_contiguous():
        raise ValueError("expected a contiguous uint8 HWC RGB device image")
    scratch = _jitter_scratch.get(img.device)
    if scratch is None:
        scratch = _jitter_scratch[img.device] = torch.zeros(1, dtype=torch.int64, device=img.device)
    L.check(L.lib().abr_img_color_jitter_u8(L.ptr(img), img.shape[0], img.shape[1], _JITTER_OPS[op], float(factor), L.ptr(scratch), L.stream()),
            "img_color_jitter")
    return img
