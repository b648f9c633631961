// TEST INFRASTRUCTURE ONLY -- force-included (-include) when oracle/Makefile compiles the
// reference's own CPU sources (maskrcnn_benchmark/csrc/vision.cpp, cpu/ROIAlign_cpu.cpp,
// cpu/nms_cpu.cpp) where they lie under /root/reference.  Those sources were written for
// torch 1.x, whose AT_DISPATCH_FLOATING_TYPES accepted `tensor.type()`
// (at::DeprecatedTypeProperties); torch 2.10 in this image only accepts at::ScalarType.
// This header re-states that one macro so the unmodified reference sources compile against
// the image's own torch headers.  It replaces no header/library that the image lacks and
// contains no reference code.
#pragma once
#include <torch/extension.h>
namespace abr_oracle_compat {
inline at::ScalarType st(const at::DeprecatedTypeProperties& t) { return t.scalarType(); }
inline at::ScalarType st(at::ScalarType t) { return t; }
}  // namespace abr_oracle_compat
#undef AT_DISPATCH_FLOATING_TYPES
#define AT_DISPATCH_FLOATING_TYPES(TYPE, NAME, ...) \
  AT_DISPATCH_SWITCH(::abr_oracle_compat::st(TYPE), NAME, AT_DISPATCH_CASE_FLOATING_TYPES(__VA_ARGS__))
