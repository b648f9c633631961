// Layout transforms, pooling, ReLU backward and the fused multi-tensor SGD step (all HBM-bound).
#include <algorithm>

#include "common.h"

namespace {

// [B,C,H,W] -> [B,H,W,Cpad] (extra channels zero).  Tiled through LDS so both sides are coalesced.
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, int C, int HW, int Cpad,
                                                            float* __restrict__ out) {
    __shared__ float t[32][33];
    const int b = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, p = p0 + tx;
        t[i][tx] = (c < C && p < HW) ? x[((size_t)b * C + c) * HW + p] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int p = p0 + i, c = c0 + tx;
        if (p < HW && c < Cpad) out[((size_t)b * HW + p) * Cpad + c] = t[tx][i];
    }
}

// The image batch: C <= 4 planes -> [B,H,W,4] (missing channels zero).  One pixel per thread: every plane is read as consecutive floats, the
// pixel leaves as ONE 16 B store.  (The 32 x 32 LDS-tiled transpose above spends 29 of its 32 channel rows on nothing here and writes four
// lanes per store: 65 us for a 4 x 3 x 600 x 1000 batch, twice per step; this form moves the same 67 MB in ~15 us.)
__global__ __launch_bounds__(256) void nchw_to_nhwc4_kernel(const float* __restrict__ x, int C, int HW, float* __restrict__ out) {
    const int b = blockIdx.y;
    const float* xb = x + (size_t)b * C * HW;
    float4* ob = reinterpret_cast<float4*>(out) + (size_t)b * HW;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        v.x = xb[p];
        if (C > 1) v.y = xb[(size_t)HW + p];
        if (C > 2) v.z = xb[(size_t)2 * HW + p];
        if (C > 3) v.w = xb[(size_t)3 * HW + p];
        ob[p] = v;
    }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ x, int C, int HW, float* __restrict__ out) {
    __shared__ float t[32][33];
    const int b = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int p = p0 + i, c = c0 + tx;
        t[i][tx] = (c < C && p < HW) ? x[((size_t)b * HW + p) * C + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, p = p0 + tx;
        if (p < HW && c < C) out[((size_t)b * C + c) * HW + p] = t[tx][i];
    }
}

// max_pool2d(kernel 3, stride 2, padding 1) NHWC, 16 B per lane along C  (resnet.py:367)
__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ x, int B, int H, int W, int C, int Ho, int Wo,
                                                       float* __restrict__ out) {
    const int cv = C / 4;
    const int64_t total = (int64_t)B * Ho * Wo * cv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = i % cv;
        const int64_t pix = i / cv;
        const int wo = pix % Wo, ho = (pix / Wo) % Ho, b = pix / ((int64_t)Wo * Ho);
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
        for (int dy = 0; dy < 3; dy++) {
            const int hi = ho * 2 - 1 + dy;
            if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; dx++) {
                const int wi = wo * 2 - 1 + dx;
                if ((unsigned)wi >= (unsigned)W) continue;
                const float4 v = reinterpret_cast<const float4*>(x)[(((size_t)b * H + hi) * W + wi) * cv + c];
                // (torch's max_pool2d keeps a NaN: `val > max || isnan(val)`, aten/src/ATen/native/cuda/DilatedMaxPool2d.cu; fmaxf would drop it)
                m.x = (v.x > m.x || v.x != v.x) ? v.x : m.x; m.y = (v.y > m.y || v.y != v.y) ? v.y : m.y;
                m.z = (v.z > m.z || v.z != v.z) ? v.z : m.z; m.w = (v.w > m.w || v.w != v.w) ? v.w : m.w;
            }
        }
        reinterpret_cast<float4*>(out)[i] = m;
    }
}

// x [N,HW,C] -> out [N,C] mean over HW
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const float* __restrict__ x, int64_t N, int HW, int C,
                                                           float* __restrict__ out) {
    const int cv = C / 4;
    const int64_t total = N * cv;
    const float inv = 1.f / (float)HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / cv;
        const int c = i % cv;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int p = 0; p < HW; p++) {
            const float4 v = reinterpret_cast<const float4*>(x)[(n * HW + p) * cv + c];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        a.x *= inv; a.y *= inv; a.z *= inv; a.w *= inv;
        reinterpret_cast<float4*>(out)[i] = a;
    }
}

// x [rows, C] (NHWC rows = (roi, bin)) -> out[row] = mean over the C channels.  One wave per row: 16 B per lane per step, then a
// wave reduction.  The rehearsal-buffer builder's `torch.mean(roi_align_features.cpu(), dim=1)` (tools/prototype_box_selection.py:84)
// without shipping the [n,1024,7,7] tensor over PCIe first.
__global__ __launch_bounds__(256) void channel_mean_kernel(const float* __restrict__ x, int64_t rows, int C, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    const int cv = C / 4;
    for (int64_t r = wave0; r < rows; r += nwaves) {
        const float4* p = reinterpret_cast<const float4*>(x + r * C);
        float a = 0.f;
        for (int c = lane; c < cv; c += 64) {
            const float4 v = p[c];
            a += (v.x + v.y) + (v.z + v.w);
        }
        for (int c = cv * 4 + lane; c < C; c += 64) a += x[r * C + c];
        a = abr::wave_sum(a);
        if (lane == 0) out[r] = a / (float)C;
    }
}

__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ g, int64_t N, int HW, int C,
                                                           float* __restrict__ gx) {
    const int cv = C / 4;
    const int64_t total = N * HW * cv;
    const float inv = 1.f / (float)HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = i % cv;
        const int64_t n = i / ((int64_t)HW * cv);
        float4 v = reinterpret_cast<const float4*>(g)[n * cv + c];
        v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
        reinterpret_cast<float4*>(gx)[i] = v;
    }
}

// the same through the ReLU that produced y (the pooled tensor): gx = y > 0 ? g / HW : 0 -- the pooling's and the ReLU's backward in one pass
// over the [N, HW, C] tensor instead of a write + (two reads + a write)
// amax (optional): max |gx| into that amax word (the f16x3 consumers of gx -- layer4's last dgrad and weight gradient -- read it)
__global__ __launch_bounds__(256) void avgpool_bwd_masked_kernel(const float* __restrict__ g, const float* __restrict__ y, int64_t N, int HW, int C,
                                                                  float* __restrict__ gx, unsigned long long* amax, unsigned epoch) {
    unsigned am = 0;
    const int cv = C / 4;
    const int64_t total = N * HW * cv;
    const float inv = 1.f / (float)HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = i % cv;
        const int64_t n = i / ((int64_t)HW * cv);
        float4 v = reinterpret_cast<const float4*>(g)[n * cv + c];
        const float4 yv = reinterpret_cast<const float4*>(y)[i];
        v.x = yv.x > 0.f ? v.x * inv : 0.f; v.y = yv.y > 0.f ? v.y * inv : 0.f;
        v.z = yv.z > 0.f ? v.z * inv : 0.f; v.w = yv.w > 0.f ? v.w * inv : 0.f;
        reinterpret_cast<float4*>(gx)[i] = v;
        if (amax)
            am = max(max(am, max(__float_as_uint(v.x) & 0x7FFFFFFFu, __float_as_uint(v.y) & 0x7FFFFFFFu)),
                     max(__float_as_uint(v.z) & 0x7FFFFFFFu, __float_as_uint(v.w) & 0x7FFFFFFFu));
    }
    if (amax) abr::h3_amax_emit(amax, epoch, am);
}

// out may alias g (in place): no __restrict__ on those two
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* g, const float* __restrict__ y, int64_t n4, int64_t n, float* out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 gv = reinterpret_cast<const float4*>(g)[i];
        const float4 yv = reinterpret_cast<const float4*>(y)[i];
        gv.x = yv.x > 0.f ? gv.x : 0.f; gv.y = yv.y > 0.f ? gv.y : 0.f;
        gv.z = yv.z > 0.f ? gv.z : 0.f; gv.w = yv.w > 0.f ? gv.w : 0.f;
        reinterpret_cast<float4*>(out)[i] = gv;
    }
    if (blockIdx.x == 0)
        for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) out[i] = y[i] > 0.f ? g[i] : 0.f;
}

__global__ __launch_bounds__(256) void add_kernel(float* __restrict__ a, const float* __restrict__ b, int64_t n4, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 av = reinterpret_cast<float4*>(a)[i];
        const float4 bv = reinterpret_cast<const float4*>(b)[i];
        av.x += bv.x; av.y += bv.y; av.z += bv.z; av.w += bv.w;
        reinterpret_cast<float4*>(a)[i] = av;
    }
    if (blockIdx.x == 0)
        for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) a[i] += b[i];
}

__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ x, int64_t n, float s, const float* __restrict__ s_dev) {
    const float k = s * (s_dev ? *s_dev : 1.f);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] *= k;
}

// Fused SGD(momentum) over ALL tensors in one launch.  Reference: one param group PER TENSOR
// (solver/build.py:7-21) -> 52 x (wd add, momentum mul/add, update) tiny launches; here one streaming pass:
// reads p,g,m and writes p,m = 5 x 4 B per element.  torch.optim.SGD semantics (dampening 0, no nesterov):
//   d = g + wd*p ; m = first ? d : mu*m + d ; p -= lr*m.
// seg_end is ascending and every segment starts on a 64-float boundary (modeling/_flat.py ALIGN), so a float4 never straddles two
// tensors: 16 B accesses, one segment lookup per float4, the (<= 256-entry) tables staged in LDS once per workgroup.
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   int64_t total, const int64_t* __restrict__ seg_end,
                                                   const float* __restrict__ lr, const float* __restrict__ wd, int n_seg,
                                                   float mu, float gscale, int first) {
    __shared__ int64_t s_end[256];
    __shared__ float s_lr[256], s_wd[256];
    const bool staged = n_seg <= 256;
    if (staged)
        for (int i = threadIdx.x; i < n_seg; i += blockDim.x) { s_end[i] = seg_end[i]; s_lr[i] = lr[i]; s_wd[i] = wd[i]; }
    __syncthreads();
    const int64_t* se = staged ? s_end : seg_end;
    const float* lrs = staged ? s_lr : lr;
    const float* wds = staged ? s_wd : wd;
    const int64_t n4 = total / 4;
    for (int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i4 < n4; i4 += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = i4 * 4;
        int lo = 0, hi = n_seg - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (i < se[mid]) hi = mid; else lo = mid + 1;
        }
        if (i + 4 > se[lo]) {  // a float4 that straddles two tensors (only with unaligned segment tables): element by element
            for (int e = 0; e < 4; e++) {
                int sg = lo;
                while (i + e >= se[sg] && sg < n_seg - 1) sg++;
                const float pe = p[i + e];
                const float d = g[i + e] * gscale + wds[sg] * pe;
                const float me = first ? d : mu * m[i + e] + d;
                m[i + e] = me;
                p[i + e] = pe - lrs[sg] * me;
            }
            continue;
        }
        const float w_ = wds[lo], l_ = lrs[lo];
        float4 pv = reinterpret_cast<float4*>(p)[i4];
        const float4 gv = reinterpret_cast<const float4*>(g)[i4];
        float4 mv = first ? make_float4(0.f, 0.f, 0.f, 0.f) : reinterpret_cast<const float4*>(m)[i4];
        const float dx = gv.x * gscale + w_ * pv.x, dy = gv.y * gscale + w_ * pv.y, dz = gv.z * gscale + w_ * pv.z, dw = gv.w * gscale + w_ * pv.w;
        mv.x = first ? dx : mu * mv.x + dx; mv.y = first ? dy : mu * mv.y + dy; mv.z = first ? dz : mu * mv.z + dz; mv.w = first ? dw : mu * mv.w + dw;
        pv.x -= l_ * mv.x; pv.y -= l_ * mv.y; pv.z -= l_ * mv.z; pv.w -= l_ * mv.w;
        reinterpret_cast<float4*>(m)[i4] = mv;
        reinterpret_cast<float4*>(p)[i4] = pv;
    }
    if (blockIdx.x == 0)  // tail (total is a multiple of 64 for flat buffers; kept for arbitrary callers)
        for (int64_t i = n4 * 4 + threadIdx.x; i < total; i += blockDim.x) {
            int lo = 0, hi = n_seg - 1;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (i < se[mid]) hi = mid; else lo = mid + 1;
            }
            const float pv = p[i];
            const float d = g[i] * gscale + wds[lo] * pv;
            const float mv = first ? d : mu * m[i] + d;
            m[i] = mv;
            p[i] = pv - lrs[lo] * mv;
        }
}

}  // namespace

extern "C" int abr_nchw_to_nhwc_pad(const float* x, int B, int C, int H, int W, int Cpad, float* out, void* stream) {
    ABR_REQUIRE(x && out && B > 0 && C > 0 && Cpad >= C, "nchw_to_nhwc_pad: bad args");
    if (Cpad == 4) {   // the image batch (3 -> 4 channels)
        const int HW = H * W;
        nchw_to_nhwc4_kernel<<<dim3((unsigned)std::min((HW + 255) / 256, 4096), (unsigned)B), 256, 0, abr::as_stream(stream)>>>(x, C, HW, out);
        ABR_CHECK_LAUNCH("nchw_to_nhwc_pad");
        return ABR_OK;
    }
    dim3 grid((H * W + 31) / 32, (Cpad + 31) / 32, B);
    nchw_to_nhwc_kernel<<<grid, 256, 0, abr::as_stream(stream)>>>(x, C, H * W, Cpad, out);
    ABR_CHECK_LAUNCH("nchw_to_nhwc_pad");
    return ABR_OK;
}
extern "C" int abr_nchw_to_nhwc(const float* x, int B, int C, int H, int W, float* out, void* stream) {
    return abr_nchw_to_nhwc_pad(x, B, C, H, W, C, out, stream);
}
extern "C" int abr_nhwc_to_nchw(const float* x, int B, int C, int H, int W, float* out, void* stream) {
    ABR_REQUIRE(x && out && B > 0 && C > 0, "nhwc_to_nchw: bad args");
    dim3 grid((H * W + 31) / 32, (C + 31) / 32, B);
    nhwc_to_nchw_kernel<<<grid, 256, 0, abr::as_stream(stream)>>>(x, C, H * W, out);
    ABR_CHECK_LAUNCH("nhwc_to_nchw");
    return ABR_OK;
}

extern "C" int abr_maxpool3x3s2(const float* x, int B, int H, int W, int C, float* out, void* stream) {
    ABR_REQUIRE(x && out && C % 4 == 0, "maxpool3x3s2: bad args (C % 4 == 0)");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int64_t total = (int64_t)B * Ho * Wo * (C / 4);
    maxpool_kernel<<<(unsigned)std::min<int64_t>((total + 255) / 256, 16384), 256, 0, abr::as_stream(stream)>>>(x, B, H, W, C,
                                                                                                                  Ho, Wo, out);
    ABR_CHECK_LAUNCH("maxpool3x3s2");
    return ABR_OK;
}

extern "C" int abr_avgpool_forward(const float* x, int N, int HW, int C, float* out, void* stream) {
    ABR_REQUIRE(N >= 0 && HW > 0 && C % 4 == 0, "avgpool_forward: bad args");
    if (N == 0) return ABR_OK;
    const int64_t total = (int64_t)N * (C / 4);
    avgpool_fwd_kernel<<<(unsigned)std::min<int64_t>((total + 255) / 256, 8192), 256, 0, abr::as_stream(stream)>>>(x, N, HW, C, out);
    ABR_CHECK_LAUNCH("avgpool_forward");
    return ABR_OK;
}
extern "C" int abr_channel_mean(const float* x, int64_t rows, int C, float* out, void* stream) {
    ABR_REQUIRE(rows >= 0 && C > 0, "channel_mean: bad args");
    if (rows == 0) return ABR_OK;
    ABR_REQUIRE(x && out, "channel_mean: null pointer");
    channel_mean_kernel<<<(unsigned)std::min<int64_t>((rows + 3) / 4, 16384), 256, 0, abr::as_stream(stream)>>>(x, rows, C, out);
    ABR_CHECK_LAUNCH("channel_mean");
    return ABR_OK;
}
extern "C" int abr_avgpool_backward(const float* g, int N, int HW, int C, float* gx, void* stream) {
    ABR_REQUIRE(N >= 0 && HW > 0 && C % 4 == 0, "avgpool_backward: bad args");
    if (N == 0) return ABR_OK;
    const int64_t total = (int64_t)N * HW * (C / 4);
    avgpool_bwd_kernel<<<(unsigned)std::min<int64_t>((total + 255) / 256, 8192), 256, 0, abr::as_stream(stream)>>>(g, N, HW, C, gx);
    ABR_CHECK_LAUNCH("avgpool_backward");
    return ABR_OK;
}

extern "C" int abr_avgpool_relu_backward_amax(const float* g, const float* y, int N, int HW, int C, float* gx, uint64_t* amax, uint32_t amax_epoch,
                                              void* stream) {
    ABR_REQUIRE(N >= 0 && HW > 0 && C % 4 == 0, "avgpool_relu_backward: bad args");
    if (N == 0) return ABR_OK;
    ABR_REQUIRE(g && y && gx, "avgpool_relu_backward: null pointer");
    const int64_t total = (int64_t)N * HW * (C / 4);
    avgpool_bwd_masked_kernel<<<(unsigned)std::min<int64_t>((total + 255) / 256, 8192), 256, 0, abr::as_stream(stream)>>>(
        g, y, N, HW, C, gx, reinterpret_cast<unsigned long long*>(amax), amax_epoch);
    ABR_CHECK_LAUNCH("avgpool_relu_backward");
    return ABR_OK;
}
extern "C" int abr_avgpool_relu_backward(const float* g, const float* y, int N, int HW, int C, float* gx, void* stream) {
    return abr_avgpool_relu_backward_amax(g, y, N, HW, C, gx, nullptr, 0, stream);
}

extern "C" int abr_relu_backward(const float* g, const float* y, int64_t n, float* out, void* stream) {
    if (n == 0) return ABR_OK;
    ABR_REQUIRE(g && y && out && n > 0, "relu_backward: bad args");
    relu_bwd_kernel<<<(unsigned)std::min<int64_t>((n / 4 + 255) / 256 + 1, 8192), 256, 0, abr::as_stream(stream)>>>(g, y, n / 4, n, out);
    ABR_CHECK_LAUNCH("relu_backward");
    return ABR_OK;
}
extern "C" int abr_add_inplace(float* a, const float* b, int64_t n, void* stream) {
    if (n == 0) return ABR_OK;
    ABR_REQUIRE(a && b && n > 0, "add_inplace: bad args");
    add_kernel<<<(unsigned)std::min<int64_t>((n / 4 + 255) / 256 + 1, 8192), 256, 0, abr::as_stream(stream)>>>(a, b, n / 4, n);
    ABR_CHECK_LAUNCH("add_inplace");
    return ABR_OK;
}

namespace {
struct LossTerms { const float* p[8]; float w[8]; int group[8]; };
// out[0] = sum_i w_i * *p_i ; out[1 + g] = the same restricted to group g (0 / 1): the trainer's total loss and its two logged parts
__global__ void loss_sum_kernel(const LossTerms t, int n, float* __restrict__ total, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float tot = 0.f, g0 = 0.f, g1 = 0.f;
        for (int i = 0; i < n; i++) {
            const float v = t.w[i] * *t.p[i];
            tot += v;
            if (t.group[i] == 0) g0 += v; else g1 += v;
        }
        *total = tot;
        out[0] = tot; out[1] = g0; out[2] = g1;
    }
}
// grads[i] = w_i * *g (the upstream gradient of the total, a device scalar)
__global__ void loss_sum_bwd_kernel(const LossTerms t, int n, const float* __restrict__ g, float* __restrict__ grads) {
    if (threadIdx.x < n && blockIdx.x == 0) grads[threadIdx.x] = t.w[threadIdx.x] * *g;
}
}  // namespace

extern "C" int abr_loss_sum(const float* const* terms_host, const float* weights_host, const int32_t* groups_host, int n, float* total, float* out,
                            void* stream) {
    ABR_REQUIRE(n >= 1 && n <= 8 && terms_host && weights_host && groups_host && total && out, "loss_sum: 1..8 terms");
    LossTerms t;
    for (int i = 0; i < 8; i++) { t.p[i] = i < n ? terms_host[i] : nullptr; t.w[i] = i < n ? weights_host[i] : 0.f; t.group[i] = i < n ? groups_host[i] : 0; }
    for (int i = 0; i < n; i++) ABR_REQUIRE(t.p[i], "loss_sum: null term");
    loss_sum_kernel<<<1, 64, 0, abr::as_stream(stream)>>>(t, n, total, out);
    ABR_CHECK_LAUNCH("loss_sum");
    return ABR_OK;
}
extern "C" int abr_loss_sum_backward(const float* weights_host, int n, const float* g, float* grads, void* stream) {
    ABR_REQUIRE(n >= 1 && n <= 8 && weights_host && g && grads, "loss_sum_backward: 1..8 terms");
    LossTerms t;
    for (int i = 0; i < 8; i++) { t.p[i] = nullptr; t.w[i] = i < n ? weights_host[i] : 0.f; t.group[i] = 0; }
    loss_sum_bwd_kernel<<<1, 64, 0, abr::as_stream(stream)>>>(t, n, g, grads);
    ABR_CHECK_LAUNCH("loss_sum_backward");
    return ABR_OK;
}

extern "C" int abr_scale_inplace(float* x, int64_t n, float s, const float* s_dev, void* stream) {
    if (n == 0) return ABR_OK;
    ABR_REQUIRE(x && n > 0, "scale_inplace: bad args");
    scale_kernel<<<(unsigned)std::min<int64_t>((n + 255) / 256, 4096), 256, 0, abr::as_stream(stream)>>>(x, n, s, s_dev);
    ABR_CHECK_LAUNCH("scale_inplace");
    return ABR_OK;
}

extern "C" int abr_sgd_momentum(float* p, const float* g, float* m, int64_t total, const int64_t* seg_end_dev,
                                const float* lr_dev, const float* wd_dev, int n_seg, float momentum, float gscale,
                                int first_step, void* stream) {
    if (total == 0) return ABR_OK;
    ABR_REQUIRE(p && g && m && seg_end_dev && lr_dev && wd_dev && n_seg > 0, "sgd_momentum: bad args");
    sgd_kernel<<<(unsigned)std::min<int64_t>((total / 4 + 255) / 256 + 1, 8192), 256, 0, abr::as_stream(stream)>>>(
        p, g, m, total, seg_end_dev, lr_dev, wd_dev, n_seg, momentum, gscale, first_step);
    ABR_CHECK_LAUNCH("sgd_momentum");
    return ABR_OK;
}
