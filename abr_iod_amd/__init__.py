"""abr_iod_amd — MI355X-native (gfx950) implementation of the Faster R-CNN R50-C4 + Attentive RoI
Distillation training hot path of YuyangSunshine/ABR_IOD.  Hand-written HIP kernels behind the C ABI
of include/abr_iod_hip.h; this package is the Python host side mirroring the reference's
maskrcnn_benchmark layers / modeling / distillation interface for that path."""
__version__ = "0.1.0"
