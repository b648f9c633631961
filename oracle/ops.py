"""TEST INFRASTRUCTURE ONLY -- ctypes/numpy wrappers over liboracle.so (oracle.c).

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by abr_iod_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "oracle"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.abr_oracle_nms.restype = C.c_int64
        _lib.abr_oracle_smooth_l1_sum.restype = C.c_double
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def roi_align_forward(feat, rois, scale, ph, pw, sr):
    feat, rois = _f32(feat), _f32(rois)
    B, Ch, H, W = feat.shape
    K = rois.shape[0]
    out = np.empty((K, Ch, ph, pw), np.float32)
    lib().abr_oracle_roi_align_forward(_p(feat), _p(rois), K, Ch, H, W, C.c_float(scale), ph, pw, sr, _p(out))
    return out


def roi_align_backward(grad, rois, scale, ph, pw, B, Ch, H, W, sr):
    grad, rois = _f32(grad), _f32(rois)
    K = rois.shape[0]
    out = np.empty((B, Ch, H, W), np.float32)
    lib().abr_oracle_roi_align_backward(_p(grad), _p(rois), K, B, Ch, H, W, C.c_float(scale), ph, pw, sr, _p(out))
    return out


def roi_align_taps(rois, H, W, scale, ph, pw, sr, max_s):
    rois = _f32(rois)
    K = rois.shape[0]
    idx = np.empty((K, ph * pw, max_s, 4), np.int32)
    grid = np.empty((K, 2), np.int32)
    lib().abr_oracle_roi_align_taps(_p(rois), K, H, W, C.c_float(scale), ph, pw, sr, max_s, _p(idx), _p(grid))
    return idx, grid


def nms(boxes, scores, thr, strict_gt=False):
    boxes, scores = _f32(boxes), _f32(scores)
    n = boxes.shape[0]
    keep = np.empty((max(n, 1),), np.int64)
    k = lib().abr_oracle_nms(_p(boxes), _p(scores), C.c_int64(n), C.c_float(thr), int(strict_gt), _p(keep))
    return keep[:k].copy()


def box_iou(a, b):
    a, b = _f32(a), _f32(b)
    out = np.empty((a.shape[0], b.shape[0]), np.float32)
    lib().abr_oracle_box_iou(_p(a), a.shape[0], _p(b), b.shape[0], _p(out))
    return out


def matcher(iou, hi, lo, allow_low_quality):
    iou = _f32(iou)
    G, n = iou.shape
    out = np.empty((n,), np.int64)
    lib().abr_oracle_matcher(_p(iou), G, n, C.c_float(hi), C.c_float(lo), int(allow_low_quality), _p(out))
    return out


def box_encode(gt, ex, weights):
    gt, ex, w = _f32(gt), _f32(ex), _f32(weights)
    out = np.empty_like(gt)
    lib().abr_oracle_box_encode(_p(gt), _p(ex), gt.shape[0], _p(w), _p(out))
    return out


def box_decode(deltas, boxes, weights):
    deltas, boxes, w = _f32(deltas), _f32(boxes), _f32(weights)
    n, k4 = deltas.shape
    out = np.empty_like(deltas)
    lib().abr_oracle_box_decode(_p(deltas), _p(boxes), n, k4 // 4, _p(w), _p(out))
    return out


def cell_anchors(stride=16, sizes=(32, 64, 128, 256, 512), ratios=(0.5, 1.0, 2.0)):
    s = np.asarray(sizes, np.float64)
    r = np.asarray(ratios, np.float64)
    out = np.empty((len(s) * len(r), 4), np.float32)
    lib().abr_oracle_cell_anchors(stride, _p(s), len(s), _p(r), len(r), _p(out))
    return out


def grid_anchors(cell, H, W, stride, image_hw, straddle=0):
    cell = _f32(cell)
    A = cell.shape[0]
    out = np.empty((H * W * A, 4), np.float32)
    vis = np.empty((H * W * A,), np.uint8)
    lib().abr_oracle_grid_anchors(_p(cell), A, H, W, stride, int(image_hw[0]), int(image_hw[1]), straddle, _p(out), _p(vis))
    return out, vis.astype(bool)


def sigmoid_focal_forward(logits, targets, gamma, alpha):
    logits = _f32(logits)
    targets = np.ascontiguousarray(targets, np.int32)
    out = np.empty_like(logits)
    lib().abr_oracle_sigmoid_focal_forward(_p(logits), _p(targets), logits.shape[0], logits.shape[1],
                                           C.c_float(gamma), C.c_float(alpha), _p(out))
    return out


def sigmoid_focal_backward(logits, targets, d_losses, gamma, alpha):
    logits, d_losses = _f32(logits), _f32(d_losses)
    targets = np.ascontiguousarray(targets, np.int32)
    out = np.empty_like(logits)
    lib().abr_oracle_sigmoid_focal_backward(_p(logits), _p(targets), _p(d_losses), logits.shape[0], logits.shape[1],
                                            C.c_float(gamma), C.c_float(alpha), _p(out))
    return out


def smooth_l1(x, t, beta, size_average):
    x, t = _f32(x), _f32(t)
    g = np.empty_like(x)
    s = lib().abr_oracle_smooth_l1_sum(_p(x), _p(t), C.c_int64(x.size), C.c_float(beta), _p(g))
    if size_average:
        return np.float32(s / x.size), g / np.float32(x.size)
    return np.float32(s), g
