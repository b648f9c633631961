// ROIAlign forward / backward for gfx950.
//
// Reference semantics: maskrcnn_benchmark/csrc/cpu/ROIAlign_cpu.cpp:17-219 (forward, the oracle),
// csrc/cuda/ROIAlign_cuda.cu:65-122 (forward) and :125-254 (backward; there is no CPU backward).
//
// Design (MI355X-first, not a translation of the one-thread-per-output-element CUDA kernel):
//   * native layout is NHWC: the 4 bilinear taps of a sample are 4 contiguous C-vectors (4 KB at
//     C=1024), read as 16 B per lane -> every tap is one fully coalesced row;
//   * a 256-thread workgroup owns `bpb` bins of ONE RoI; the per-sample integer tap indices and
//     weights depend only on (roi, bin, sample), so they are computed once per workgroup into a small
//     LDS table (one entry per thread per chunk) and then broadcast-read by all channel lanes;
//   * coordinate arithmetic is compiled with FP contraction OFF and in the reference's association
//     order, so (int)y / (int)x -- the tap indices -- are bit-identical to the CPU reference; the value
//     accumulation uses the reference's order too (w1*v1 + w2*v2 + w3*v3 + w4*v4, then /count);
//   * blockIdx is remapped so each XCD (private L2) walks a contiguous range of (roi, bin) tiles:
//     neighbouring bins/RoIs re-read the same feature rows from that XCD's L2;
//   * bin_step=2 computes only the even bins -- the only ones layer4's stride-2 1x1 conv reads.
// Bound: HBM (output write / grad read) with L2-resident tap re-reads; see DESIGN.md.
#include <stdlib.h>

#include "common.h"

namespace {

struct RoiGeom {
    float y0, x0, bh, bw;
    int gh, gw, b;
};

struct Tap {
    int p0, p1, p2, p3;  // flat y*W+x, or -1 when the sample is rejected
    float w0, w1, w2, w3;
};

#pragma clang fp contract(off)
__device__ __forceinline__ RoiGeom roi_geom(const float* __restrict__ r, float scale, int PH, int PW, int sr) {
    RoiGeom g;
    g.b = (int)r[0];
    const float sw = r[1] * scale, sh = r[2] * scale, ew = r[3] * scale, eh = r[4] * scale;  // no rounding
    const float rw = fmaxf(ew - sw, 1.f), rh = fmaxf(eh - sh, 1.f);                          // malformed -> 1x1
    g.x0 = sw;
    g.y0 = sh;
    g.bh = rh / (float)PH;
    g.bw = rw / (float)PW;
    g.gh = sr > 0 ? sr : (int)ceilf(rh / (float)PH);
    g.gw = sr > 0 ? sr : (int)ceilf(rw / (float)PW);
    return g;
}

#pragma clang fp contract(off)
__device__ __forceinline__ Tap make_tap(const RoiGeom& g, int H, int W, int ph, int pw, int iy, int ix) {
    // start + ph*bin + ((i+.5)*bin)/grid  -- keep exactly this association (ROIAlign_cpu.cpp:39-45)
    float y = g.y0 + ph * g.bh + (float)(iy + .5f) * g.bh / (float)g.gh;
    float x = g.x0 + pw * g.bw + (float)(ix + .5f) * g.bw / (float)g.gw;
    Tap t;
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) {
        t.p0 = t.p1 = t.p2 = t.p3 = -1;
        t.w0 = t.w1 = t.w2 = t.w3 = 0.f;
        return t;
    }
    if (y <= 0) y = 0;
    if (x <= 0) x = 0;
    int yl = (int)y, xl = (int)x, yh, xh;
    if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
    if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
    const float ly = y - yl, lx = x - xl;
    const float hy = 1.f - ly, hx = 1.f - lx;
    t.w0 = hy * hx; t.w1 = hy * lx; t.w2 = ly * hx; t.w3 = ly * lx;
    t.p0 = yl * W + xl; t.p1 = yl * W + xh; t.p2 = yh * W + xl; t.p3 = yh * W + xh;
    return t;
}

// ---------------------------------------------------------------------------------------------------
// NHWC forward.  grid.x = K * blocks_per_roi, block = 256 = bpb bins x tx channel-lanes.
// ---------------------------------------------------------------------------------------------------
template <int VEC>
struct VecT;
template <>
struct VecT<4> { using type = float4; };
template <>
struct VecT<1> { using type = float; };

#pragma clang fp contract(off)
template <int VEC>
__global__ __launch_bounds__(256) void roi_align_fwd_nhwc(const float* __restrict__ feat, const float* __restrict__ rois,
                                                           int K, int C, int H, int W, float scale, int PH, int PW,
                                                           int sr, int step, int PHo, int PWo, int bpb, int tx,
                                                           int blocks_per_roi, float* __restrict__ out) {
    __shared__ int4 s_idx[256];
    __shared__ float4 s_w[256];
    const unsigned nblk = gridDim.x;
    const unsigned bid = abr::xcd_remap(blockIdx.x, nblk);
    const int n = bid / blocks_per_roi;
    const int bin0 = (bid % blocks_per_roi) * bpb;
    const int nbins = PHo * PWo;
    const int tch = 256 / bpb;  // samples per chunk per bin

    const RoiGeom g = roi_geom(rois + 5 * (size_t)n, scale, PH, PW, sr);
    const int ns = g.gh * g.gw;
    const float count = (float)ns;

    const int bl = threadIdx.x / tx;  // local bin
    const int cl = threadIdx.x % tx;  // channel lane
    const int bin = bin0 + bl;
    const bool bin_ok = bin < nbins;
    const int ph = (bin / PWo) * step, pw = (bin % PWo) * step;
    const int cvecs = C / VEC;
    const float* fb = feat + (size_t)g.b * H * W * C;

    // entry computed by this thread in every chunk: (local bin eb, sample-in-chunk es)
    const int eb = threadIdx.x / tch, es = threadIdx.x % tch;
    const int ebin = bin0 + eb;
    const int eph = (ebin / PWo) * step, epw = (ebin % PWo) * step;

    using V = typename VecT<VEC>::type;
    // up to 4 channel vectors per lane are kept in registers per pass (C <= 4*tx*VEC per pass)
    for (int c0 = 0; c0 < cvecs; c0 += tx) {
        const int cv = c0 + cl;
        const bool c_ok = bin_ok && cv < cvecs;
        float acc[VEC];
#pragma unroll
        for (int i = 0; i < VEC; i++) acc[i] = 0.f;
        for (int sb = 0; sb < ns; sb += tch) {
            __syncthreads();
            {
                const int s = sb + es;
                Tap t;
                if (s < ns && ebin < nbins) {
                    t = make_tap(g, H, W, eph, epw, s / g.gw, s % g.gw);
                } else {
                    t.p0 = t.p1 = t.p2 = t.p3 = -1;
                    t.w0 = t.w1 = t.w2 = t.w3 = 0.f;
                }
                s_idx[threadIdx.x] = make_int4(t.p0, t.p1, t.p2, t.p3);
                s_w[threadIdx.x] = make_float4(t.w0, t.w1, t.w2, t.w3);
            }
            __syncthreads();
            if (c_ok) {
                const int lim = min(tch, ns - sb);
                for (int s = 0; s < lim; s++) {
                    const int4 p = s_idx[bl * tch + s];
                    if (p.x < 0) continue;  // uniform across the bin's lanes
                    const float4 w = s_w[bl * tch + s];
                    const V v0 = *reinterpret_cast<const V*>(fb + (size_t)p.x * C + cv * VEC);
                    const V v1 = *reinterpret_cast<const V*>(fb + (size_t)p.y * C + cv * VEC);
                    const V v2 = *reinterpret_cast<const V*>(fb + (size_t)p.z * C + cv * VEC);
                    const V v3 = *reinterpret_cast<const V*>(fb + (size_t)p.w * C + cv * VEC);
                    const float* a0 = reinterpret_cast<const float*>(&v0);
                    const float* a1 = reinterpret_cast<const float*>(&v1);
                    const float* a2 = reinterpret_cast<const float*>(&v2);
                    const float* a3 = reinterpret_cast<const float*>(&v3);
#pragma unroll
                    for (int i = 0; i < VEC; i++)
                        acc[i] += w.x * a0[i] + w.y * a1[i] + w.z * a2[i] + w.w * a3[i];
                }
            }
        }
        if (c_ok) {
            V o;
            float* op = reinterpret_cast<float*>(&o);
#pragma unroll
            for (int i = 0; i < VEC; i++) op[i] = acc[i] / count;
            *reinterpret_cast<V*>(out + ((size_t)n * nbins + bin) * C + cv * VEC) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// NHWC backward: scatter (g*w_k)/count to the 4 taps with hardware fp32 atomics (global_atomic_add_f32).
// ---------------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
template <int VEC>
__global__ __launch_bounds__(256) void roi_align_bwd_nhwc(const float* __restrict__ grad, const float* __restrict__ rois,
                                                           int K, int C, int H, int W, float scale, int PH, int PW,
                                                           int sr, int step, int PHo, int PWo, int bpb, int tx,
                                                           int blocks_per_roi, float* __restrict__ gfeat) {
    __shared__ int4 s_idx[256];
    __shared__ float4 s_w[256];
    const unsigned bid = abr::xcd_remap(blockIdx.x, gridDim.x);
    const int n = bid / blocks_per_roi;
    const int bin0 = (bid % blocks_per_roi) * bpb;
    const int nbins = PHo * PWo;
    const int tch = 256 / bpb;

    const RoiGeom g = roi_geom(rois + 5 * (size_t)n, scale, PH, PW, sr);
    const int ns = g.gh * g.gw;
    const float count = (float)ns;

    const int bl = threadIdx.x / tx, cl = threadIdx.x % tx;
    const int bin = bin0 + bl;
    const bool bin_ok = bin < nbins;
    const int cvecs = C / VEC;
    float* gb = gfeat + (size_t)g.b * H * W * C;

    const int eb = threadIdx.x / tch, es = threadIdx.x % tch;
    const int ebin = bin0 + eb;
    const int eph = (ebin / PWo) * step, epw = (ebin % PWo) * step;

    using V = typename VecT<VEC>::type;
    for (int c0 = 0; c0 < cvecs; c0 += tx) {
        const int cv = c0 + cl;
        const bool c_ok = bin_ok && cv < cvecs;
        V gv;
        float* gp = reinterpret_cast<float*>(&gv);
        if (c_ok) gv = *reinterpret_cast<const V*>(grad + ((size_t)n * nbins + bin) * C + cv * VEC);
        for (int sb = 0; sb < ns; sb += tch) {
            __syncthreads();
            {
                const int s = sb + es;
                Tap t;
                if (s < ns && ebin < nbins) {
                    t = make_tap(g, H, W, eph, epw, s / g.gw, s % g.gw);
                } else {
                    t.p0 = t.p1 = t.p2 = t.p3 = -1;
                    t.w0 = t.w1 = t.w2 = t.w3 = 0.f;
                }
                s_idx[threadIdx.x] = make_int4(t.p0, t.p1, t.p2, t.p3);
                s_w[threadIdx.x] = make_float4(t.w0, t.w1, t.w2, t.w3);
            }
            __syncthreads();
            if (c_ok) {
                const int lim = min(tch, ns - sb);
                for (int s = 0; s < lim; s++) {
                    const int4 p = s_idx[bl * tch + s];
                    if (p.x < 0) continue;
                    const float4 w = s_w[bl * tch + s];
                    float* d0 = gb + (size_t)p.x * C + cv * VEC;
                    float* d1 = gb + (size_t)p.y * C + cv * VEC;
                    float* d2 = gb + (size_t)p.z * C + cv * VEC;
                    float* d3 = gb + (size_t)p.w * C + cv * VEC;
#pragma unroll
                    for (int i = 0; i < VEC; i++) {
                        unsafeAtomicAdd(d0 + i, gp[i] * w.x / count);
                        unsafeAtomicAdd(d1 + i, gp[i] * w.y / count);
                        unsafeAtomicAdd(d2 + i, gp[i] * w.z / count);
                        unsafeAtomicAdd(d3 + i, gp[i] * w.w / count);
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// NHWC backward, separable per-RoI form (the one the hot path uses).
//
// The direct form above issues one atomic per (bin, sample, tap, channel): K*49*gh*gw*4*C of them -- at
// K=2048, C=1024 that is ~3.7e9 atomics and the kernel is bound by the L2 atomic rate (25 ms measured).
// Bilinear weights and the sample grid are both products of a y-part and an x-part, and so is the
// reject test (y out of range OR x out of range), hence for one RoI
//     dFeat[y][x][c] += sum_ph Wy[ph][y] * sum_pw Wx[pw][x] * g[ph][pw][c] / count
// with Wy[ph][y] = sum of the y-weights that bin-row ph's samples put on feature row y (same for Wx).
// A workgroup builds the two small tables in LDS with the SAME coordinate code as the forward (so the
// integer taps are identical), then walks the RoI's pixel footprint once: per row it folds the <=7 bin
// rows into 7 register vectors, per pixel it folds the bin columns and issues ONE vector atomic.
// Atomics drop from 4*gh*gw*49 to ~(7*bh+1)*(7*bw+1) per RoI and channel (~9x fewer on RPN proposals).
// ---------------------------------------------------------------------------------------------------
constexpr int kMaxPo = 8;  // pooled bins per axis handled by the separable kernel

#pragma clang fp contract(off)
__device__ __forceinline__ void axis_tap(float start, float bin, int grid, int p, int i, int L, int* lo, int* hi,
                                         float* wlo, float* whi, bool* ok) {
    float v = start + p * bin + (float)(i + .5f) * bin / (float)grid;
    if (v < -1.0f || v > (float)L) { *ok = false; return; }
    *ok = true;
    if (v <= 0) v = 0;
    int l = (int)v, h;
    if (l >= L - 1) { h = l = L - 1; v = (float)l; } else h = l + 1;
    const float lw = v - l;
    *lo = l; *hi = h; *wlo = 1.f - lw; *whi = lw;
}

template <int VEC>
__global__ __launch_bounds__(256) void roi_align_bwd_nhwc_sep(const float* __restrict__ grad, const float* __restrict__ rois,
                                                               int K, int C, int H, int W, float scale, int PH, int PW,
                                                               int sr, int step, int PHo, int PWo, int tx, int cchunks,
                                                               float* __restrict__ gfeat) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Wy = sm;                 // [PHo][H]
    float* Wx = sm + PHo * H;       // [PWo][W]
    int* rng = reinterpret_cast<int*>(Wx + PWo * W);  // ymin, ymax, xmin, xmax
    const unsigned bid = abr::xcd_remap(blockIdx.x, gridDim.x);
    const int n = bid / cchunks, chunk = bid % cchunks;
    const RoiGeom g = roi_geom(rois + 5 * (size_t)n, scale, PH, PW, sr);
    const float inv_count = 1.f / (float)(g.gh * g.gw);

    for (int i = threadIdx.x; i < PHo * H + PWo * W; i += 256) sm[i] = 0.f;
    if (threadIdx.x == 0) { rng[0] = H; rng[1] = -1; rng[2] = W; rng[3] = -1; }
    __syncthreads();
    for (int t = threadIdx.x; t < PHo * g.gh; t += 256) {
        const int pi = t / g.gh, iy = t % g.gh;
        int lo, hi; float wl, wh; bool ok;
        axis_tap(g.y0, g.bh, g.gh, pi * step, iy, H, &lo, &hi, &wl, &wh, &ok);
        if (ok) {
            atomicAdd(&Wy[pi * H + lo], wl);
            atomicAdd(&Wy[pi * H + hi], wh);
            atomicMin(&rng[0], lo); atomicMax(&rng[1], hi);
        }
    }
    for (int t = threadIdx.x; t < PWo * g.gw; t += 256) {
        const int pi = t / g.gw, ix = t % g.gw;
        int lo, hi; float wl, wh; bool ok;
        axis_tap(g.x0, g.bw, g.gw, pi * step, ix, W, &lo, &hi, &wl, &wh, &ok);
        if (ok) {
            atomicAdd(&Wx[pi * W + lo], wl);
            atomicAdd(&Wx[pi * W + hi], wh);
            atomicMin(&rng[2], lo); atomicMax(&rng[3], hi);
        }
    }
    __syncthreads();
    const int ymin = rng[0], ymax = rng[1], xmin = rng[2], xmax = rng[3];
    if (ymax < ymin || xmax < xmin) return;

    using V = typename VecT<VEC>::type;
    const int cvecs = C / VEC;
    const int cl = threadIdx.x % tx, rl = threadIdx.x / tx, nrl = 256 / tx;
    const int cv = chunk * tx + cl;
    if (cv >= cvecs) return;
    const float* gr = grad + (size_t)n * PHo * PWo * C + cv * VEC;
    float* gb = gfeat + (size_t)g.b * H * W * C + cv * VEC;

    for (int y = ymin + rl; y <= ymax; y += nrl) {
        float T[kMaxPo][VEC];
#pragma unroll
        for (int j = 0; j < kMaxPo; j++)
#pragma unroll
            for (int i = 0; i < VEC; i++) T[j][i] = 0.f;
        bool any = false;
        for (int pi = 0; pi < PHo; pi++) {
            const float wy = Wy[pi * H + y] * inv_count;
            if (wy == 0.f) continue;
            any = true;
#pragma unroll
            for (int j = 0; j < kMaxPo; j++) {
                if (j < PWo) {
                    const V v = *reinterpret_cast<const V*>(gr + ((size_t)pi * PWo + j) * C);
                    const float* a = reinterpret_cast<const float*>(&v);
#pragma unroll
                    for (int i = 0; i < VEC; i++) T[j][i] += wy * a[i];
                }
            }
        }
        if (!any) continue;
        for (int x = xmin; x <= xmax; x++) {
            float o[VEC];
#pragma unroll
            for (int i = 0; i < VEC; i++) o[i] = 0.f;
            bool hit = false;
#pragma unroll
            for (int j = 0; j < kMaxPo; j++) {
                if (j < PWo) {
                    const float wx = Wx[j * W + x];
                    if (wx != 0.f) {
                        hit = true;
#pragma unroll
                        for (int i = 0; i < VEC; i++) o[i] += wx * T[j][i];
                    }
                }
            }
            if (hit) {
                float* d = gb + ((size_t)y * W + x) * C;
#pragma unroll
                for (int i = 0; i < VEC; i++) unsafeAtomicAdd(d + i, o[i]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// NCHW compatibility kernels (the reference's own tensor layout; drop-in for `_C.roi_align_*`).
// One thread per (n, c, ph, pw) like the reference; pw fastest so a wave reads neighbouring taps.
// ---------------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void roi_align_fwd_nchw(const float* __restrict__ feat, const float* __restrict__ rois,
                                                           int64_t total, int C, int H, int W, float scale, int PH,
                                                           int PW, int sr, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int pw = i % PW, ph = (i / PW) % PH, c = (i / PW / PH) % C, n = i / PW / PH / C;
        const RoiGeom g = roi_geom(rois + 5 * (size_t)n, scale, PH, PW, sr);
        const float* plane = feat + ((size_t)g.b * C + c) * H * W;
        float acc = 0.f;
        for (int iy = 0; iy < g.gh; iy++)
            for (int ix = 0; ix < g.gw; ix++) {
                const Tap t = make_tap(g, H, W, ph, pw, iy, ix);
                if (t.p0 < 0) continue;
                acc += t.w0 * plane[t.p0] + t.w1 * plane[t.p1] + t.w2 * plane[t.p2] + t.w3 * plane[t.p3];
            }
        out[i] = acc / (float)(g.gh * g.gw);
    }
}

#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void roi_align_bwd_nchw(const float* __restrict__ grad, const float* __restrict__ rois,
                                                           int64_t total, int C, int H, int W, float scale, int PH,
                                                           int PW, int sr, float* __restrict__ gfeat) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int pw = i % PW, ph = (i / PW) % PH, c = (i / PW / PH) % C, n = i / PW / PH / C;
        const RoiGeom g = roi_geom(rois + 5 * (size_t)n, scale, PH, PW, sr);
        float* plane = gfeat + ((size_t)g.b * C + c) * H * W;
        const float gv = grad[i];
        const float count = (float)(g.gh * g.gw);
        for (int iy = 0; iy < g.gh; iy++)
            for (int ix = 0; ix < g.gw; ix++) {
                const Tap t = make_tap(g, H, W, ph, pw, iy, ix);
                if (t.p0 < 0) continue;
                unsafeAtomicAdd(plane + t.p0, gv * t.w0 / count);
                unsafeAtomicAdd(plane + t.p1, gv * t.w1 / count);
                unsafeAtomicAdd(plane + t.p2, gv * t.w2 / count);
                unsafeAtomicAdd(plane + t.p3, gv * t.w3 / count);
            }
    }
}

__global__ void roi_align_taps_kernel(const float* __restrict__ rois, int K, int H, int W, float scale, int PH, int PW,
                                      int sr, int max_s, int32_t* __restrict__ idx, int32_t* __restrict__ grid) {
    const int64_t total = (int64_t)K * PH * PW * max_s;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int s = i % max_s, pw = (i / max_s) % PW, ph = (i / max_s / PW) % PH, n = i / max_s / PW / PH;
        const RoiGeom g = roi_geom(rois + 5 * (size_t)n, scale, PH, PW, sr);
        if (s == 0 && ph == 0 && pw == 0) { grid[2 * n] = g.gh; grid[2 * n + 1] = g.gw; }
        int32_t* o = idx + i * 4;
        if (s >= g.gh * g.gw) { o[0] = o[1] = o[2] = o[3] = -2; continue; }
        const Tap t = make_tap(g, H, W, ph, pw, s / g.gw, s % g.gw);
        o[0] = t.p0; o[1] = t.p1; o[2] = t.p2; o[3] = t.p3;
    }
}

// choose (tx, bpb): tx = channel lanes per bin (power of two >= #vectors, <=256), bpb = 256/tx capped at 8
void pick_shape(int cvecs, int nbins, int* tx, int* bpb) {
    int t = 1;
    while (t < cvecs && t < 256) t <<= 1;
    if (t < 32) t = 32;
    int b = 256 / t;
    if (b > 8) b = 8;
    if (b > nbins) {  // never more local bins than bins
        b = 1;
        while (b * 2 <= nbins && b * 2 <= 8 && 256 / (b * 2) >= t) b *= 2;
    }
    *tx = 256 / b;  // widen lanes so that tx*bpb == 256 exactly
    *bpb = b;
}

}  // namespace

extern "C" int abr_roi_align_forward(const float* feat, const float* rois, int K, int B, int C, int H, int W,
                                     float scale, int PH, int PW, int sr, int bin_step, int layout, float* out,
                                     void* stream) {
    ABR_REQUIRE(K >= 0 && B > 0 && C > 0 && H > 0 && W > 0 && PH > 0 && PW > 0, "roi_align_forward: bad shape");
    ABR_REQUIRE(bin_step >= 1, "roi_align_forward: bin_step must be >= 1");
    ABR_REQUIRE(layout == ABR_NCHW || layout == ABR_NHWC, "roi_align_forward: bad layout");
    if (K == 0) return ABR_OK;
    ABR_REQUIRE(feat && rois && out, "roi_align_forward: null pointer");
    hipStream_t st = abr::as_stream(stream);
    if (layout == ABR_NCHW) {
        ABR_REQUIRE(bin_step == 1, "roi_align_forward: bin_step>1 needs NHWC");
        const int64_t total = (int64_t)K * C * PH * PW;
        const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 32);
        roi_align_fwd_nchw<<<grid, 256, 0, st>>>(feat, rois, total, C, H, W, scale, PH, PW, sr, out);
    } else {
        const int PHo = (PH + bin_step - 1) / bin_step, PWo = (PW + bin_step - 1) / bin_step;
        const int nbins = PHo * PWo;
        int tx, bpb;
        if (C % 4 == 0) {
            pick_shape(C / 4, nbins, &tx, &bpb);
            const int bpr = (nbins + bpb - 1) / bpb;
            roi_align_fwd_nhwc<4><<<(unsigned)(K * bpr), 256, 0, st>>>(feat, rois, K, C, H, W, scale, PH, PW, sr,
                                                                        bin_step, PHo, PWo, bpb, tx, bpr, out);
        } else {
            pick_shape(C, nbins, &tx, &bpb);
            const int bpr = (nbins + bpb - 1) / bpb;
            roi_align_fwd_nhwc<1><<<(unsigned)(K * bpr), 256, 0, st>>>(feat, rois, K, C, H, W, scale, PH, PW, sr,
                                                                        bin_step, PHo, PWo, bpb, tx, bpr, out);
        }
    }
    ABR_CHECK_LAUNCH("roi_align_forward");
    return ABR_OK;
}

extern "C" int abr_roi_align_backward(const float* grad, const float* rois, int K, int B, int C, int H, int W,
                                      float scale, int PH, int PW, int sr, int bin_step, int layout, int accumulate,
                                      float* gfeat, void* stream) {
    ABR_REQUIRE(K >= 0 && B > 0 && C > 0 && H > 0 && W > 0 && PH > 0 && PW > 0, "roi_align_backward: bad shape");
    ABR_REQUIRE(bin_step >= 1, "roi_align_backward: bin_step must be >= 1");
    ABR_REQUIRE(layout == ABR_NCHW || layout == ABR_NHWC, "roi_align_backward: bad layout");
    ABR_REQUIRE(gfeat, "roi_align_backward: null output");
    hipStream_t st = abr::as_stream(stream);
    if (!accumulate) {
        if (hipMemsetAsync(gfeat, 0, sizeof(float) * (size_t)B * C * H * W, st) != hipSuccess) {
            abr::set_error("roi_align_backward: memset failed");
            return ABR_E_LAUNCH;
        }
    }
    if (K == 0) return ABR_OK;
    ABR_REQUIRE(grad && rois, "roi_align_backward: null pointer");
    if (layout == ABR_NCHW) {
        ABR_REQUIRE(bin_step == 1, "roi_align_backward: bin_step>1 needs NHWC");
        const int64_t total = (int64_t)K * C * PH * PW;
        const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 32);
        roi_align_bwd_nchw<<<grid, 256, 0, st>>>(grad, rois, total, C, H, W, scale, PH, PW, sr, gfeat);
    } else {
        const int PHo = (PH + bin_step - 1) / bin_step, PWo = (PW + bin_step - 1) / bin_step;
        const int nbins = PHo * PWo;
        int tx, bpb;
        const size_t sep_lds = sizeof(float) * ((size_t)PHo * H + (size_t)PWo * W) + 16;
        if (PHo <= kMaxPo && PWo <= kMaxPo && sep_lds <= 60 * 1024 && !getenv("ABR_ROIALIGN_BWD_DIRECT")) {
            const int vec = (C % 4 == 0) ? 4 : 1;
            const int cvecs = C / vec;
            int t = 1;
            while (t < cvecs && t < 256) t <<= 1;
            if (t < 16) t = 16;
            const int cchunks = (cvecs + t - 1) / t;
            if (vec == 4)
                roi_align_bwd_nhwc_sep<4><<<(unsigned)(K * cchunks), 256, sep_lds, st>>>(grad, rois, K, C, H, W, scale, PH, PW, sr,
                                                                                        bin_step, PHo, PWo, t, cchunks, gfeat);
            else
                roi_align_bwd_nhwc_sep<1><<<(unsigned)(K * cchunks), 256, sep_lds, st>>>(grad, rois, K, C, H, W, scale, PH, PW, sr,
                                                                                        bin_step, PHo, PWo, t, cchunks, gfeat);
        } else if (C % 4 == 0) {
            pick_shape(C / 4, nbins, &tx, &bpb);
            const int bpr = (nbins + bpb - 1) / bpb;
            roi_align_bwd_nhwc<4><<<(unsigned)(K * bpr), 256, 0, st>>>(grad, rois, K, C, H, W, scale, PH, PW, sr,
                                                                        bin_step, PHo, PWo, bpb, tx, bpr, gfeat);
        } else {
            pick_shape(C, nbins, &tx, &bpb);
            const int bpr = (nbins + bpb - 1) / bpb;
            roi_align_bwd_nhwc<1><<<(unsigned)(K * bpr), 256, 0, st>>>(grad, rois, K, C, H, W, scale, PH, PW, sr,
                                                                        bin_step, PHo, PWo, bpb, tx, bpr, gfeat);
        }
    }
    ABR_CHECK_LAUNCH("roi_align_backward");
    return ABR_OK;
}

extern "C" int abr_roi_align_taps(const float* rois, int K, int H, int W, float scale, int PH, int PW, int sr,
                                  int max_s, int32_t* idx, int32_t* grid, void* stream) {
    ABR_REQUIRE(K >= 0 && max_s > 0, "roi_align_taps: bad shape");
    if (K == 0) return ABR_OK;
    const int64_t total = (int64_t)K * PH * PW * max_s;
    roi_align_taps_kernel<<<(unsigned)std::min<int64_t>((total + 255) / 256, 8192), 256, 0, abr::as_stream(stream)>>>(
        rois, K, H, W, scale, PH, PW, sr, max_s, idx, grid);
    ABR_CHECK_LAUNCH("roi_align_taps");
    return ABR_OK;
}
