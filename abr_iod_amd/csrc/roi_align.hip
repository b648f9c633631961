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
                                                           int blocks_per_roi, int cslices, float* __restrict__ out) {
    __shared__ int4 s_idx[256];
    __shared__ float4 s_w[256];
    const unsigned nblk = gridDim.x;
    // cslices == 8: consecutive workgroups land on consecutive XCDs, so workgroup b takes channel slice b % 8 of tile b / 8 -- every
    // XCD then reads ONE eighth of the channels of every image (1.2 MB per image at C = 1024: resident in its 4 MiB L2, where a
    // whole image's map, 9.7 MB, is not) and the taps that neighbouring bins / RoIs share are L2 hits instead of fabric fetches
    const unsigned bid = cslices > 1 ? blockIdx.x / (unsigned)cslices : abr::xcd_remap(blockIdx.x, nblk);
    const int slice = cslices > 1 ? (int)(blockIdx.x % (unsigned)cslices) : 0;
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
    const int cps = C / VEC / cslices;                     // channel vectors of this workgroup's slice
    const int cv_begin = slice * cps, cvecs = cv_begin + cps;
    const float* fb = feat + (size_t)g.b * H * W * C;

    // entry computed by this thread in every chunk: (local bin eb, sample-in-chunk es)
    const int eb = threadIdx.x / tch, es = threadIdx.x % tch;
    const int ebin = bin0 + eb;
    const int eph = (ebin / PWo) * step, epw = (ebin % PWo) * step;

    using V = typename VecT<VEC>::type;
    // up to 4 channel vectors per lane are kept in registers per pass (C <= 4*tx*VEC per pass)
    for (int c0 = cv_begin; c0 < cvecs; c0 += tx) {
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
// NHWC backward, atomic-free GATHER form (what the training step uses).
//
// The scatter forms above are bound by the L2 atomic rate (3.1 ms per step at B=4).  The same separable identity lets every
// feature pixel GATHER its gradient instead:
//     dFeat[b][y][x][:] = sum over RoIs r of image b:  sum_ph Wy_r[ph][y] * sum_pw Wx_r[pw][x] * g_r[ph][pw][:]
// Pass 1 (one small workgroup per RoI) writes the RoI's weight tables Wy_r [PHo][H], Wx_r [PWo][Wp] (1/count folded into
// Wy) and its footprint rectangle, using the forward's coordinate code -> identical integer taps.
// Pass 2: a workgroup owns one feature row y and XT=8 consecutive pixels for all channels (16 B per lane); it walks the RoI
// list with wave-uniform (scalar) range tests, and for an overlapping RoI folds the <=2 active bin rows into PWo register
// vectors and then the bin columns into the 8 pixel accumulators.  Every dFeat element is written exactly once, coalesced,
// in a fixed summation order (deterministic; no zero-fill pass, `accumulate` is a read-add-write).
// ---------------------------------------------------------------------------------------------------
struct RoiRect { int b, ymin, ymax, xmin, xmax, pad0, pad1, pad2; };

__global__ __launch_bounds__(64) void roi_bwd_tables_kernel(const float* __restrict__ rois, int K, int H, int W, int Wp,
                                                             float scale, int PH, int PW, int sr, int step, int PHo, int PWo,
                                                             float* __restrict__ Wy, float* __restrict__ Wx,
                                                             RoiRect* __restrict__ rect) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sy = sm;             // [PHo][H]
    float* sx = sm + PHo * H;   // [PWo][Wp]
    int* rng = reinterpret_cast<int*>(sx + PWo * Wp);
    const int n = blockIdx.x;
    const RoiGeom g = roi_geom(rois + 5 * (size_t)n, scale, PH, PW, sr);
    const float inv_count = 1.f / (float)(g.gh * g.gw);
    for (int i = threadIdx.x; i < PHo * H + PWo * Wp; i += 64) sm[i] = 0.f;
    if (threadIdx.x == 0) { rng[0] = H; rng[1] = -1; rng[2] = W; rng[3] = -1; }
    __syncthreads();
    // one thread per bin walks the bin's samples IN ORDER and owns the bin's table row (round 5: the former one-thread-per-sample form added
    // into the row with LDS float atomics, whose order -- and the tables' last bits, and with them the feature gradient's -- changed from run to run)
    if (threadIdx.x < PHo) {
        const int pi = threadIdx.x;
        for (int iy = 0; iy < g.gh; iy++) {
            int lo, hi; float wl, wh; bool ok;
            axis_tap(g.y0, g.bh, g.gh, pi * step, iy, H, &lo, &hi, &wl, &wh, &ok);
            if (ok) {
                sy[pi * H + lo] += wl;
                sy[pi * H + hi] += wh;
                atomicMin(&rng[0], lo); atomicMax(&rng[1], hi);
            }
        }
    } else if (threadIdx.x >= 32 && threadIdx.x < 32 + PWo) {
        const int pi = threadIdx.x - 32;
        for (int ix = 0; ix < g.gw; ix++) {
            int lo, hi; float wl, wh; bool ok;
            axis_tap(g.x0, g.bw, g.gw, pi * step, ix, W, &lo, &hi, &wl, &wh, &ok);
            if (ok) {
                sx[pi * Wp + lo] += wl;
                sx[pi * Wp + hi] += wh;
                atomicMin(&rng[2], lo); atomicMax(&rng[3], hi);
            }
        }
    }
    __syncthreads();
    float* oy = Wy + (size_t)n * PHo * H;
    float* ox = Wx + (size_t)n * PWo * Wp;
    for (int i = threadIdx.x; i < PHo * H; i += 64) oy[i] = sy[i] * inv_count;
    for (int i = threadIdx.x; i < PWo * Wp; i += 64) ox[i] = sx[i];
    if (threadIdx.x == 0) {
        RoiRect r;
        r.b = g.b; r.ymin = rng[0]; r.ymax = rng[1]; r.xmin = rng[2]; r.xmax = rng[3]; r.pad0 = r.pad1 = r.pad2 = 0;
        rect[n] = r;
    }
}

constexpr int kXT = 8;  // pixels per workgroup along x

// Pass 1b: for every gather workgroup -- (image b, feature row y, x-tile of kXT pixels) -- the ORDERED list (ascending RoI index) of the RoIs
// of image b whose footprint touches that row and that x-tile.  The gather then walks only RoIs it has work for: with one list per ROW
// (round 2) a workgroup skipped ~12 of every 13 entries -- other images' RoIs, RoIs left or right of its 8 pixels -- at the price of two
// dependent scalar loads each, and that walk, not the gradient traffic, was most of the kernel (2048 RoIs: 273 -> see DESIGN.md).
// Ordered block compaction (ballot + prefix), so the summation order of the gather stays the fixed ascending-RoI order.
__global__ __launch_bounds__(256) void roi_tile_lists_kernel(const RoiRect* __restrict__ rect, int K, int H, int n_xt, int32_t* __restrict__ lists,
                                                             int32_t* __restrict__ counts) {
    __shared__ int wave_cnt[4];
    const int xt = blockIdx.x, y = blockIdx.y, b = blockIdx.z;
    const int x0 = xt * kXT;
    const int lid = (b * H + y) * n_xt + xt;
    int32_t* out = lists + (size_t)lid * K;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int base = 0;
    for (int r0 = 0; r0 < K; r0 += 256) {
        const int r = r0 + threadIdx.x;
        bool hit = false;
        if (r < K) {
            const RoiRect rc = rect[r];
            hit = rc.b == b && y >= rc.ymin && y <= rc.ymax && rc.xmax >= rc.xmin && rc.xmax >= x0 && rc.xmin < x0 + kXT;
        }
        const unsigned long long m = __ballot(hit);
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; w++) off += wave_cnt[w];
        if (hit) out[off + __popcll(m & ((1ull << lane) - 1ull))] = r;
        base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) counts[lid] = base;
}

// NS > 1 (round 4): the workgroup is NS waves on the SAME 256 channels, wave s walking list entries s, s + NS, ... -- a tile's list (~100 RoIs at
// 512 RoIs per image) is a chain of dependent L2 round trips per entry (weights, then gradient rows), and the kernel's time was the longest chain;
// the partial sums meet in LDS and are added in wave order (fixed order: still deterministic).  blockDim.x = 64 NS.
template <int PO, int NS = 1>  // PO = compile-time bound on PHo and PWo (4 for bin_step=2 on 7x7, 8 otherwise)
__global__ __launch_bounds__(256) void roi_align_bwd_gather_kernel(const float* __restrict__ grad, int K, int C, int H, int W, int Wp,
                                                                    int PHo, int PWo, const float* __restrict__ Wy,
                                                                    const float* __restrict__ Wx, const RoiRect* __restrict__ rect,
                                                                    const int32_t* __restrict__ lists, const int32_t* __restrict__ counts,
                                                                    int cchunks, int accumulate, float* __restrict__ gfeat, int n_xt, int B,
                                                                    int xcd_map) {
    // (a workgroup of 8 x-tiles x 128 channels with the channel chunk chosen by XCD, like the forward kernel's slices, measured
    // slower: 1.29 vs 1.10 ms on 2048 large RoIs -- the x-tiles' re-reads of a RoI's gradient already hit in L2)
    int xt, chunk, y, b;
    if (xcd_map) {
        // round 6: 1-D grid of 8 x ceil(S / P) workgroups (S spatial tiles, P = 8 / cchunks XCDs per channel chunk).  Workgroup h runs on XCD
        // h % 8: that XCD owns ONE channel chunk and a contiguous 1/P of the (image, row, x-tile) domain, walked x-tile first -- the ~15 tiles
        // that read a RoI's gradient rows (neighbours in x and y) sit on one XCD and find them in ITS L2.  With the 3-D grid the x-tiles of a
        // row alternated between two XCDs (n_xt cchunks = 32 workgroups per row): every gradient row was fetched from HBM twice (PMC 1.4-2.0x).
        const unsigned h = blockIdx.x, xcd = h & 7u, i = h >> 3;
        const unsigned P = 8u / (unsigned)cchunks, S = (unsigned)n_xt * (unsigned)H * (unsigned)B, Sp = (S + P - 1) / P;
        chunk = (int)(xcd % (unsigned)cchunks);
        const unsigned t = (xcd / (unsigned)cchunks) * Sp + i;
        if (i >= Sp || t >= S) return;
        xt = (int)(t % (unsigned)n_xt);
        y = (int)((t / (unsigned)n_xt) % (unsigned)H);
        b = (int)(t / ((unsigned)n_xt * (unsigned)H));
    } else {
        xt = blockIdx.x / cchunks; chunk = blockIdx.x % cchunks;
        y = blockIdx.y; b = blockIdx.z;
    }
    const int x0 = xt * kXT;
    const int ws = NS > 1 ? (int)(threadIdx.x >> 6) : 0;                       // which share of the list this wave walks
    const int cv = NS > 1 ? chunk * 64 + (int)(threadIdx.x & 63) : chunk * (int)blockDim.x + (int)threadIdx.x;
    const bool c_ok = cv < C / 4;
    float4 acc[kXT];
#pragma unroll
    for (int i = 0; i < kXT; i++) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int lid = (b * H + y) * n_xt + xt;
    const int32_t* row_list = lists + (size_t)lid * K;
    const int n_row = counts[lid];
    // Per list entry: wave-uniform data (the RoI index, its <= PO row weights for this feature row, its PWo x 8 column weights for these
    // pixels) comes through SCALAR loads; the gradient rows come as 16 B per lane.  Everything an entry needs is REQUESTED before anything
    // is used -- all row weights, then the gradient vectors of every active bin row, then the column weights -- so the loads of an entry
    // are in flight together (round 2's form tested `j < PWo` / `wy != 0` around every single load: each 16 B load was followed by its own
    // full wait, ~8 dependent L2 round trips per entry), and the NEXT entry's RoI index is fetched while this one is being accumulated.
    if constexpr (PO <= 4) {
    constexpr int PC = PO, JB = PO;   // (PO == 4 here: 64 registers of gradient vectors, 32 scalar column weights)
    int r_next = n_row > ws ? row_list[ws] : 0;
    for (int e = ws; e < n_row; e += NS) {
        const int r = __builtin_amdgcn_readfirstlane(r_next);   // every entry is a RoI of this image that touches this row and these pixels
        if (e + NS < n_row) r_next = row_list[e + NS];
        const float* wyr = Wy + (size_t)r * PHo * H + y;
        const float* g = grad + (size_t)r * PHo * PWo * C + cv * 4;
        const float* wxr = Wx + (size_t)r * PWo * Wp + x0;
        float wy[PO];
#pragma unroll
        for (int pi = 0; pi < PO; pi++) wy[pi] = pi < PHo ? wyr[(pi < PHo ? pi : 0) * H] : 0.f;
        float4 T[PO];
#pragma unroll
        for (int j = 0; j < PO; j++) T[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int p0 = 0; p0 < PO; p0 += PC) {
            float4 gv[PC][PO];
#pragma unroll
            for (int q = 0; q < PC; q++) {
                if (wy[p0 + q] != 0.f && c_ok) {   // wave-uniform: a scalar branch around PO loads, none of them waited for here
#pragma unroll
                    for (int j = 0; j < PO; j++) gv[q][j] = *reinterpret_cast<const float4*>(g + ((size_t)(p0 + q) * PWo + (j < PWo ? j : PWo - 1)) * C);
                }
            }
#pragma unroll
            for (int q = 0; q < PC; q++) {
                const float w = wy[p0 + q];
                if (w != 0.f && c_ok) {
#pragma unroll
                    for (int j = 0; j < PO; j++) { T[j].x += w * gv[q][j].x; T[j].y += w * gv[q][j].y; T[j].z += w * gv[q][j].z; T[j].w += w * gv[q][j].w; }
                }
            }
        }
        bool any = false;
#pragma unroll
        for (int pi = 0; pi < PO; pi++) any |= wy[pi] != 0.f;
        if (!any) continue;
#pragma unroll
        for (int j0 = 0; j0 < PO; j0 += JB) {   // a few bin columns at a time: their uniform weights live in scalar registers
            float wv[JB][kXT];
#pragma unroll
            for (int q = 0; q < JB; q++) {
                const int j = j0 + q;
                const float* src = wxr + (j < PWo ? j : PWo - 1) * Wp;          // 8 consecutive pixels, 32 B aligned
                const float4 w0 = *reinterpret_cast<const float4*>(src), w1 = *reinterpret_cast<const float4*>(src + 4);
                const float keep = j < PWo ? 1.f : 0.f;                         // (bin columns past PWo do not exist: weight 0)
                wv[q][0] = keep * w0.x; wv[q][1] = keep * w0.y; wv[q][2] = keep * w0.z; wv[q][3] = keep * w0.w;
                wv[q][4] = keep * w1.x; wv[q][5] = keep * w1.y; wv[q][6] = keep * w1.z; wv[q][7] = keep * w1.w;
            }
#pragma unroll
            for (int q = 0; q < JB; q++) {
                if (j0 + q < PO) {
#pragma unroll
                    for (int i = 0; i < kXT; i++) {
                        acc[i].x += wv[q][i] * T[j0 + q].x; acc[i].y += wv[q][i] * T[j0 + q].y; acc[i].z += wv[q][i] * T[j0 + q].z; acc[i].w += wv[q][i] * T[j0 + q].w;
                    }
                }
            }
        }
    }
    } else {
    // 8 bins per axis (all-bin pooling of the 64-RoI distillation passes): the batched form above would need 8 x 8 gradient vectors and 64
    // uniform weights live at once and spills (1.08 vs 0.80 ms on 2048 RoIs); here ONE bin row's vectors are requested together (column
    // indices past PWo clamped: their weight is zero), then the next row's, and the column weights come two bin columns at a time
    int r_next = n_row > ws ? row_list[ws] : 0;
    for (int e = ws; e < n_row; e += NS) {
        const int r = __builtin_amdgcn_readfirstlane(r_next);
        if (e + NS < n_row) r_next = row_list[e + NS];
        const float* wyr = Wy + (size_t)r * PHo * H + y;
        const float* g = grad + (size_t)r * PHo * PWo * C + cv * 4;
        const float* wxr = Wx + (size_t)r * PWo * Wp + x0;
        float wy[PO];
#pragma unroll
        for (int pi = 0; pi < PO; pi++) wy[pi] = pi < PHo ? wyr[(pi < PHo ? pi : 0) * H] : 0.f;
        float4 T[PO];
#pragma unroll
        for (int j = 0; j < PO; j++) T[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        bool any = false;
#pragma unroll
        for (int pi = 0; pi < PO; pi++) {
            const float w = wy[pi];
            if (w != 0.f) {   // wave-uniform
                any = true;
                if (c_ok) {
                    float4 gv[PO];
#pragma unroll
                    for (int j = 0; j < PO; j++) gv[j] = *reinterpret_cast<const float4*>(g + ((size_t)pi * PWo + (j < PWo ? j : PWo - 1)) * C);
#pragma unroll
                    for (int j = 0; j < PO; j++) { T[j].x += w * gv[j].x; T[j].y += w * gv[j].y; T[j].z += w * gv[j].z; T[j].w += w * gv[j].w; }
                }
            }
        }
        if (!any) continue;
#pragma unroll
        for (int j0 = 0; j0 < PO; j0 += 2) {
            float wv[2][kXT];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int j = j0 + q;
                const float* src = wxr + (j < PWo ? j : PWo - 1) * Wp;
                const float4 w0 = *reinterpret_cast<const float4*>(src), w1 = *reinterpret_cast<const float4*>(src + 4);
                const float keep = j < PWo ? 1.f : 0.f;
                wv[q][0] = keep * w0.x; wv[q][1] = keep * w0.y; wv[q][2] = keep * w0.z; wv[q][3] = keep * w0.w;
                wv[q][4] = keep * w1.x; wv[q][5] = keep * w1.y; wv[q][6] = keep * w1.z; wv[q][7] = keep * w1.w;
            }
#pragma unroll
            for (int q = 0; q < 2; q++)
#pragma unroll
                for (int i = 0; i < kXT; i++) {
                    acc[i].x += wv[q][i] * T[j0 + q].x; acc[i].y += wv[q][i] * T[j0 + q].y; acc[i].z += wv[q][i] * T[j0 + q].z; acc[i].w += wv[q][i] * T[j0 + q].w;
                }
        }
    }
    }
    if constexpr (NS > 1) {   // partial sums of waves 1 .. NS-1 -> LDS; wave 0 adds them in wave order
        __shared__ float4 part[NS - 1][kXT][64];
        const int lane = threadIdx.x & 63;
        if (ws > 0) {
#pragma unroll
            for (int i = 0; i < kXT; i++) part[ws - 1][i][lane] = acc[i];
        }
        __syncthreads();
        if (ws > 0) return;
#pragma unroll
        for (int s2 = 0; s2 < NS - 1; s2++)
#pragma unroll
            for (int i = 0; i < kXT; i++) {
                const float4 v = part[s2][i][lane];
                acc[i].x += v.x; acc[i].y += v.y; acc[i].z += v.z; acc[i].w += v.w;
            }
    }
    if (!c_ok) return;
#pragma unroll
    for (int i = 0; i < kXT; i++) {
        const int x = x0 + i;
        if (x >= W) break;
        float4* d = reinterpret_cast<float4*>(gfeat + (((size_t)b * H + y) * W + x) * C) + cv;
        float4 v = acc[i];
        if (accumulate) { const float4 o = *d; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        *d = v;
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

// ---------------------------------------------------------------------------------------------------
// float64 instantiation of the NCHW pair: the reference dispatches float AND double (AT_DISPATCH_FLOATING_TYPES, ROIAlign_cuda.cu:283,329;
// ROIAlign_cpu.cpp:242).  The same arithmetic with every T = double, as the reference's templates give it: coordinates, weights and the
// accumulation in double, the (i + .5f) sample offset a float literal converted to T (ROIAlign_cuda.cu:104-109).
// ---------------------------------------------------------------------------------------------------
struct RoiGeomD {
    double y0, x0, bh, bw;
    int gh, gw, b;
};
#pragma clang fp contract(off)
__device__ __forceinline__ RoiGeomD roi_geom_d(const double* __restrict__ r, double scale, int PH, int PW, int sr) {
    RoiGeomD g;
    g.b = (int)r[0];
    const double sw = r[1] * scale, sh = r[2] * scale, ew = r[3] * scale, eh = r[4] * scale;
    const double rw = fmax(ew - sw, 1.), rh = fmax(eh - sh, 1.);
    g.x0 = sw; g.y0 = sh;
    g.bh = rh / (double)PH;
    g.bw = rw / (double)PW;
    g.gh = sr > 0 ? sr : (int)ceil(rh / (double)PH);
    g.gw = sr > 0 ? sr : (int)ceil(rw / (double)PW);
    return g;
}
struct TapD {
    int p0, p1, p2, p3;
    double w0, w1, w2, w3;
};
#pragma clang fp contract(off)
__device__ __forceinline__ TapD make_tap_d(const RoiGeomD& g, int H, int W, int ph, int pw, int iy, int ix) {
    double y = g.y0 + ph * g.bh + (double)(iy + .5f) * g.bh / (double)g.gh;
    double x = g.x0 + pw * g.bw + (double)(ix + .5f) * g.bw / (double)g.gw;
    TapD t;
    if (y < -1.0 || y > (double)H || x < -1.0 || x > (double)W) {
        t.p0 = t.p1 = t.p2 = t.p3 = -1;
        t.w0 = t.w1 = t.w2 = t.w3 = 0.;
        return t;
    }
    if (y <= 0) y = 0;
    if (x <= 0) x = 0;
    int yl = (int)y, xl = (int)x, yh, xh;
    if (yl >= H - 1) { yh = yl = H - 1; y = (double)yl; } else yh = yl + 1;
    if (xl >= W - 1) { xh = xl = W - 1; x = (double)xl; } else xh = xl + 1;
    const double ly = y - yl, lx = x - xl;
    const double hy = 1. - ly, hx = 1. - lx;
    t.w0 = hy * hx; t.w1 = hy * lx; t.w2 = ly * hx; t.w3 = ly * lx;
    t.p0 = yl * W + xl; t.p1 = yl * W + xh; t.p2 = yh * W + xl; t.p3 = yh * W + xh;
    return t;
}
#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void roi_align_fwd_nchw_f64(const double* __restrict__ feat, const double* __restrict__ rois, int64_t total, int C,
                                                               int H, int W, double scale, int PH, int PW, int sr, double* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int pw = i % PW, ph = (i / PW) % PH, c = (i / PW / PH) % C, n = i / PW / PH / C;
        const RoiGeomD g = roi_geom_d(rois + 5 * (size_t)n, scale, PH, PW, sr);
        const double* plane = feat + ((size_t)g.b * C + c) * H * W;
        double acc = 0.;
        for (int iy = 0; iy < g.gh; iy++)
            for (int ix = 0; ix < g.gw; ix++) {
                const TapD t = make_tap_d(g, H, W, ph, pw, iy, ix);
                if (t.p0 < 0) continue;
                acc += t.w0 * plane[t.p0] + t.w1 * plane[t.p1] + t.w2 * plane[t.p2] + t.w3 * plane[t.p3];
            }
        out[i] = acc / (double)(g.gh * g.gw);
    }
}
#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void roi_align_bwd_nchw_f64(const double* __restrict__ grad, const double* __restrict__ rois, int64_t total, int C,
                                                               int H, int W, double scale, int PH, int PW, int sr, double* __restrict__ gfeat) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int pw = i % PW, ph = (i / PW) % PH, c = (i / PW / PH) % C, n = i / PW / PH / C;
        const RoiGeomD g = roi_geom_d(rois + 5 * (size_t)n, scale, PH, PW, sr);
        double* plane = gfeat + ((size_t)g.b * C + c) * H * W;
        const double gv = grad[i];
        const double count = (double)(g.gh * g.gw);
        for (int iy = 0; iy < g.gh; iy++)
            for (int ix = 0; ix < g.gw; ix++) {
                const TapD t = make_tap_d(g, H, W, ph, pw, iy, ix);
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
        static const bool slice_on = !(getenv("ABR_ROIALIGN_CSLICES") && atoi(getenv("ABR_ROIALIGN_CSLICES")) == 0);
        if (C % 4 == 0) {
            // one channel slice per XCD when a slice still fills 32 lanes of 16 B (C >= 1024) and the map is too big for one L2
            const int cslices = (slice_on && (C / 4) % 8 == 0 && C / 4 / 8 >= 32 && (int64_t)H * W * C * 4 > (2 << 20)) ? 8 : 1;
            pick_shape(C / 4 / cslices, nbins, &tx, &bpb);
            const int bpr = (nbins + bpb - 1) / bpb;
            roi_align_fwd_nhwc<4><<<(unsigned)(K * bpr * cslices), 256, 0, st>>>(feat, rois, K, C, H, W, scale, PH, PW, sr,
                                                                                  bin_step, PHo, PWo, bpb, tx, bpr, cslices, out);
        } else {
            pick_shape(C, nbins, &tx, &bpb);
            const int bpr = (nbins + bpb - 1) / bpb;
            roi_align_fwd_nhwc<1><<<(unsigned)(K * bpr), 256, 0, st>>>(feat, rois, K, C, H, W, scale, PH, PW, sr,
                                                                        bin_step, PHo, PWo, bpb, tx, bpr, 1, out);
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

extern "C" int abr_roi_align_forward_f64(const double* feat, const double* rois, int K, int B, int C, int H, int W, double scale, int PH, int PW,
                                         int sr, double* out, void* stream) {
    ABR_REQUIRE(K >= 0 && B > 0 && C > 0 && H > 0 && W > 0 && PH > 0 && PW > 0, "roi_align_forward_f64: bad shape");
    if (K == 0) return ABR_OK;
    ABR_REQUIRE(feat && rois && out, "roi_align_forward_f64: null pointer");
    const int64_t total = (int64_t)K * C * PH * PW;
    const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 32);
    roi_align_fwd_nchw_f64<<<grid, 256, 0, abr::as_stream(stream)>>>(feat, rois, total, C, H, W, scale, PH, PW, sr, out);
    ABR_CHECK_LAUNCH("roi_align_forward_f64");
    return ABR_OK;
}

extern "C" int abr_roi_align_backward_f64(const double* grad, const double* rois, int K, int B, int C, int H, int W, double scale, int PH, int PW,
                                          int sr, double* gfeat, void* stream) {
    ABR_REQUIRE(K >= 0 && B > 0 && C > 0 && H > 0 && W > 0 && PH > 0 && PW > 0, "roi_align_backward_f64: bad shape");
    ABR_REQUIRE(gfeat, "roi_align_backward_f64: null output");
    hipStream_t st = abr::as_stream(stream);
    if (hipMemsetAsync(gfeat, 0, sizeof(double) * (size_t)B * C * H * W, st) != hipSuccess) {   // at::zeros, ROIAlign_cuda.cu:316
        abr::set_error("roi_align_backward_f64: memset failed");
        return ABR_E_LAUNCH;
    }
    if (K == 0) return ABR_OK;
    ABR_REQUIRE(grad && rois, "roi_align_backward_f64: null pointer");
    const int64_t total = (int64_t)K * C * PH * PW;
    const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 32);
    roi_align_bwd_nchw_f64<<<grid, 256, 0, st>>>(grad, rois, total, C, H, W, scale, PH, PW, sr, gfeat);
    ABR_CHECK_LAUNCH("roi_align_backward_f64");
    return ABR_OK;
}

static inline int round8(int w) { return (w + 7) / 8 * 8; }

extern "C" int64_t abr_roi_align_backward_ws_bytes(int K, int B, int H, int W, int PH, int PW, int bin_step) {
    const int PHo = (PH + bin_step - 1) / bin_step, PWo = (PW + bin_step - 1) / bin_step;
    const int64_t n_lists = (int64_t)B * H * ((W + kXT - 1) / kXT);
    return (int64_t)K * ((int64_t)PHo * H + (int64_t)PWo * round8(W)) * 4 + (int64_t)K * sizeof(RoiRect) + 256 +
           (n_lists * K + n_lists) * 4 + 64;   // + one RoI list (worst case K entries) and its length per gather workgroup
}

extern "C" int abr_roi_align_backward_gather(const float* grad, const float* rois, int K, int B, int C, int H, int W, float scale,
                                             int PH, int PW, int sr, int bin_step, int accumulate, float* gfeat, void* workspace,
                                             int64_t ws_bytes, void* stream) {
    ABR_REQUIRE(K >= 0 && B > 0 && C > 0 && H > 0 && W > 0 && PH > 0 && PW > 0 && bin_step >= 1, "roi_align_backward_gather: bad shape");
    ABR_REQUIRE(C % 4 == 0, "roi_align_backward_gather: C must be a multiple of 4");
    ABR_REQUIRE(gfeat, "roi_align_backward_gather: null output");
    const int PHo = (PH + bin_step - 1) / bin_step, PWo = (PW + bin_step - 1) / bin_step;
    ABR_REQUIRE(PHo <= 8 && PWo <= 8, "roi_align_backward_gather: at most 8 bins per axis");
    hipStream_t st = abr::as_stream(stream);
    if (K == 0) {
        if (!accumulate && hipMemsetAsync(gfeat, 0, sizeof(float) * (size_t)B * C * H * W, st) != hipSuccess) return ABR_E_LAUNCH;
        return ABR_OK;
    }
    ABR_REQUIRE(grad && rois && workspace, "roi_align_backward_gather: null pointer");
    ABR_REQUIRE(ws_bytes >= abr_roi_align_backward_ws_bytes(K, B, H, W, PH, PW, bin_step), "roi_align_backward_gather: workspace too small");
    const int Wp = round8(W);
    float* Wy = (float*)workspace;
    float* Wx = Wy + (size_t)K * PHo * H;
    RoiRect* rect = (RoiRect*)(((uintptr_t)(Wx + (size_t)K * PWo * Wp) + 63) & ~(uintptr_t)63);
    const size_t lds = sizeof(float) * ((size_t)PHo * H + (size_t)PWo * Wp) + 16;
    ABR_REQUIRE(lds <= 60 * 1024, "roi_align_backward_gather: feature map too large for the table builder");
    roi_bwd_tables_kernel<<<K, 64, lds, st>>>(rois, K, H, W, Wp, scale, PH, PW, sr, bin_step, PHo, PWo, Wy, Wx, rect);
    int32_t* lists = (int32_t*)(((uintptr_t)(rect + K) + 63) & ~(uintptr_t)63);
    const int n_xt = (W + kXT - 1) / kXT;
    int32_t* counts = lists + (size_t)B * H * n_xt * K;
    roi_tile_lists_kernel<<<dim3((unsigned)n_xt, (unsigned)H, (unsigned)B), 256, 0, st>>>(rect, K, H, n_xt, lists, counts);
    // one wave per workgroup (64 lanes x 16 B = 256 channels): four times the workgroups of a 256-thread block, so that the dependent
    // chain of each (list entry -> weights -> gradient rows) has more neighbours to hide behind (ABR_ROIALIGN_BWD_TB=256: round 2's blocks)
    static const int tb = getenv("ABR_ROIALIGN_BWD_TB") ? atoi(getenv("ABR_ROIALIGN_BWD_TB")) : 64;
    // ABR_ROIALIGN_BWD_SPLIT (default 4; 1 = round 3's one wave per list): waves per workgroup sharing a tile's RoI list (see the kernel)
    static const int ns = getenv("ABR_ROIALIGN_BWD_SPLIT") ? atoi(getenv("ABR_ROIALIGN_BWD_SPLIT")) : 4;
    const int rec = abr::prof_start(st, abr::PROF_ROIALIGN_BWD, 0.0);
    // ABR_ROIALIGN_BWD_XCD (default 1; 0 = rounds 2-5's 3-D grid): channel chunk and spatial range of a workgroup chosen by the XCD it runs on
    static const int xcd_on = getenv("ABR_ROIALIGN_BWD_XCD") ? atoi(getenv("ABR_ROIALIGN_BWD_XCD")) : 1;
    auto grid_of = [&](int cchunks, int* xcd_map) {
        *xcd_map = xcd_on && (cchunks == 1 || cchunks == 2 || cchunks == 4 || cchunks == 8) ? 1 : 0;
        if (!*xcd_map) return dim3((unsigned)(n_xt * cchunks), (unsigned)H, (unsigned)B);
        const int64_t P = 8 / cchunks, S = (int64_t)n_xt * H * B;
        return dim3((unsigned)(8 * ((S + P - 1) / P)), 1u, 1u);
    };
    if (ns == 4) {
        const int cchunks = (C / 4 + 63) / 64;
        int xm = 0;
        const dim3 grid = grid_of(cchunks, &xm);
        if (PHo <= 4 && PWo <= 4)
            roi_align_bwd_gather_kernel<4, 4><<<grid, 256, 0, st>>>(grad, K, C, H, W, Wp, PHo, PWo, Wy, Wx, rect, lists, counts, cchunks, accumulate, gfeat, n_xt, B, xm);
        else
            roi_align_bwd_gather_kernel<8, 4><<<grid, 256, 0, st>>>(grad, K, C, H, W, Wp, PHo, PWo, Wy, Wx, rect, lists, counts, cchunks, accumulate, gfeat, n_xt, B, xm);
    } else {
        const int TB = (tb == 64 || tb == 128 || tb == 256) ? tb : 64;
        const int cchunks = (C / 4 + TB - 1) / TB;
        int xm = 0;
        const dim3 grid = grid_of(cchunks, &xm);
        if (PHo <= 4 && PWo <= 4)
            roi_align_bwd_gather_kernel<4><<<grid, TB, 0, st>>>(grad, K, C, H, W, Wp, PHo, PWo, Wy, Wx, rect, lists, counts, cchunks, accumulate, gfeat, n_xt, B, xm);
        else
            roi_align_bwd_gather_kernel<8><<<grid, TB, 0, st>>>(grad, K, C, H, W, Wp, PHo, PWo, Wy, Wx, rect, lists, counts, cchunks, accumulate, gfeat, n_xt, B, xm);
    }
    abr::prof_stop(st, rec);
    ABR_CHECK_LAUNCH("roi_align_backward_gather");
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
