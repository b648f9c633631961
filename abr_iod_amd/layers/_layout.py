"""Layout convention of the host side.

Every activation the model hands to user code has the REFERENCE's logical shape [B,C,H,W] but
channels-last memory (it is `nhwc.permute(0,3,1,2)`), so shapes/indexing match maskrcnn_benchmark while
the HIP kernels see their native NHWC.  Tensors that arrive in plain NCHW memory are converted once with
the library's tiled transpose kernel."""
from .. import ops


def as_nhwc(t):
    """logical [B,C,H,W] -> contiguous [B,H,W,C] (view when already channels-last)."""
    v = t.permute(0, 2, 3, 1)
    if v.is_contiguous():
        return ops.amax_carry(v, t)   # (the same elements: an amax tag of the f16x3 arithmetic stays valid)
    return ops.nchw_to_nhwc(t.contiguous())


def from_nhwc(x):
    """contiguous [B,H,W,C] -> logical [B,C,H,W] view (channels-last memory)."""
    return ops.amax_carry(x.permute(0, 3, 1, 2), x)
