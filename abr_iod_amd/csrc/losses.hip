// Detector + RoI-distillation losses: forward value and gradient in one launch each.
// In the reference each of these is a chain of 5-15 small ATen kernels plus autograd; here the scalar and
// the gradient (for a known upstream scale) come out of a single pass over the logits.
#include <float.h>

#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// Sigmoid focal loss -- csrc/cuda/SigmoidFocalLoss_cuda.cu:20-101 (sub-expressions with `1.` literals are
// evaluated in double there; kept so, it is elementwise and fp64 is cheap on CDNA4).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void focal_fwd(const float* __restrict__ logits, const int32_t* __restrict__ targets,
                                                  int64_t total, int C, float gamma, float alpha,
                                                  float* __restrict__ losses) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int n = i / C, d = i % C, t = targets[n];
        const float c1 = (t == d + 1), c2 = (t >= 0) & (t != d + 1);
        const float zn = (float)(1.0 - alpha), zp = alpha, x = logits[i];
        const float p = (float)(1. / (1. + expf(-x)));
        const float term1 = (float)(powf((float)(1. - p), gamma) * logf(fmaxf(p, FLT_MIN)));
        const float term2 =
            (float)(powf(p, gamma) * (-1. * x * (x >= 0) - logf((float)(1. + expf((float)(x - 2. * x * (x >= 0)))))));
        float l = 0.f;
        l += -c1 * term1 * zp;
        l += -c2 * term2 * zn;
        losses[i] = l;
    }
}

__global__ __launch_bounds__(256) void focal_bwd(const float* __restrict__ logits, const int32_t* __restrict__ targets,
                                                  const float* __restrict__ d_losses, int64_t total, int C, float gamma,
                                                  float alpha, float* __restrict__ d_logits) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int n = i / C, d = i % C, t = targets[n];
        const float c1 = (t == d + 1), c2 = (t >= 0) & (t != d + 1);
        const float zn = (float)(1.0 - alpha), zp = alpha, x = logits[i];
        const float p = (float)(1. / (1. + expf(-x)));
        const float term1 = (float)(powf((float)(1. - p), gamma) * (1. - p - (p * gamma * logf(fmaxf(p, FLT_MIN)))));
        const float term2 = (float)(powf(p, gamma) *
                                    ((-1. * x * (x >= 0) - logf((float)(1. + expf((float)(x - 2. * x * (x >= 0)))))) *
                                         (1. - p) * gamma - p));
        float g = 0.f;
        g += -c1 * term1 * zp;
        g += -c2 * term2 * zn;
        d_logits[i] = g * d_losses[i];
    }
}

// float64 instantiation (AT_DISPATCH_FLOATING_TYPES, SigmoidFocalLoss_cuda.cu:128,172): the reference's template with T = double keeps its
// expf / powf / logf calls, i.e. float transcendentals inside double arithmetic; the same expressions here.
__global__ __launch_bounds__(256) void focal_fwd_f64(const double* __restrict__ logits, const int32_t* __restrict__ targets, int64_t total, int C,
                                                      float gamma, float alpha, double* __restrict__ losses) {   // (gamma / alpha are `const float` in the reference's template too)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int n = i / C, d = i % C, t = targets[n];
        const double c1 = (t == d + 1), c2 = (t >= 0) & (t != d + 1);
        const double zn = (1.0 - alpha), zp = alpha, x = logits[i];
        const double p = 1. / (1. + expf((float)-x));
        const double term1 = powf((float)(1. - p), gamma) * logf((float)fmax(p, (double)FLT_MIN));
        const double term2 = powf((float)p, gamma) * (-1. * x * (x >= 0) - logf((float)(1. + expf((float)(x - 2. * x * (x >= 0))))));
        double l = 0.0;
        l += -c1 * term1 * zp;
        l += -c2 * term2 * zn;
        losses[i] = l;
    }
}
__global__ __launch_bounds__(256) void focal_bwd_f64(const double* __restrict__ logits, const int32_t* __restrict__ targets, const double* __restrict__ d_losses,
                                                      int64_t total, int C, float gamma, float alpha, double* __restrict__ d_logits) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int n = i / C, d = i % C, t = targets[n];
        const double c1 = (t == d + 1), c2 = (t >= 0) & (t != d + 1);
        const double zn = (1.0 - alpha), zp = alpha, x = logits[i];
        const double p = 1. / (1. + expf((float)-x));
        const double term1 = powf((float)(1. - p), gamma) * (1. - p - (p * gamma * logf((float)fmax(p, (double)FLT_MIN))));
        const double term2 = powf((float)p, gamma) *
                             ((-1. * x * (x >= 0) - logf((float)(1. + expf((float)(x - 2. * x * (x >= 0)))))) * (1. - p) * gamma - p);
        double g = 0.0;
        g += -c1 * term1 * zp;
        g += -c2 * term2 * zn;
        d_logits[i] = g * d_losses[i];
    }
}

// Add this workgroup's share `v` (valid in thread 0) to *loss_out: through the deterministic last-arrival sum when there is scratch (the
// result is then STORED: loss_out need not be zero), else with an atomic (loss_out zeroed by the host).
__device__ __forceinline__ void loss_add(float v, float* loss_out, const abr::DetWs ws) {
    if (ws.part) {
        const float in[1] = {v};
        float tot[1];
        if (abr::det_sum_last<1>(in, ws, tot) && threadIdx.x == 0) *loss_out = tot[0];
    } else if (threadIdx.x == 0) {
        atomicAdd(loss_out, v);
    }
}

// ------------------------------------------------------------------------------------------------
// smooth L1 -- layers/smooth_l1_loss.py:6-17
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sl1(float d, float beta, float* g) {
    const float a = fabsf(d);
    if (a < beta) { *g = d / beta; return 0.5f * a * a / beta; }
    *g = d > 0.f ? 1.f : -1.f;
    return a - 0.5f * beta;
}

__global__ __launch_bounds__(256) void smooth_l1_kernel(const float* __restrict__ x, const float* __restrict__ t, int64_t n,
                                                         float beta, float scale, float* __restrict__ loss_out,
                                                         float gscale, float* __restrict__ grad, const abr::DetWs ws) {
    __shared__ float sm[4];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float g;
        acc += sl1(x[i] - t[i], beta, &g);
        if (grad) grad[i] = g * gscale * scale;
    }
    acc = abr::block_sum<4>(acc, sm);
    loss_add(acc * scale, loss_out, ws);
}

// x row rows[i], columns col0[i]..col0[i]+3  vs  t row trows[i] (t is [*,4]; trows==NULL -> same row as x)
__global__ __launch_bounds__(256) void smooth_l1_rows_kernel(const float* __restrict__ x, int x_cols,
                                                              const float* __restrict__ t, const int64_t* __restrict__ rows,
                                                              const int64_t* __restrict__ col0, const int64_t* __restrict__ trows,
                                                              int n_rows, float beta,
                                                              float scale, const float* __restrict__ denom_dev,
                                                              float* __restrict__ loss_out, float gscale,
                                                              float* __restrict__ grad, const abr::DetWs ws) {
    __shared__ float sm[4];
    float acc = 0.f;
    if (denom_dev) scale = scale / fmaxf(*denom_dev, 1.f);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_rows * 4; i += gridDim.x * blockDim.x) {
        const int64_t r = rows[i >> 2];
        if (r < 0) continue;  // -1 padding of a fixed-size index list
        const int64_t c = (col0 ? col0[i >> 2] : 0) + (i & 3);
        float g;
        const int64_t tr = trows ? trows[i >> 2] : r;
        acc += sl1(x[r * x_cols + c] - t[tr * 4 + (i & 3)], beta, &g);
        if (grad) grad[r * x_cols + c] = g * gscale * scale;
    }
    acc = abr::block_sum<4>(acc, sm);
    loss_add(acc * scale, loss_out, ws);
}

// ------------------------------------------------------------------------------------------------
// Classification loss of the box head -- modeling/roi_heads/box_head/loss.py:151-162
// one thread per RoI row (K <= 128 classes, rows are 84 B at K=21)
// ------------------------------------------------------------------------------------------------
constexpr int kMaxK = 128;

// 8 lanes per row (a 256-thread workgroup takes 32 rows): a lane owns the classes c = lane, lane + 8, ...; row maximum and sums are
// xor-shuffle reductions inside the 8-lane group.
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                          int n, int K, int ldz, int ldg, int inclusive, int n_old, const int* __restrict__ n_valid,
                                                          float* __restrict__ loss_out, float gscale, float* __restrict__ dz, const abr::DetWs ws) {
    __shared__ float sm[4];
    constexpr int G = 8;
    const int i = blockIdx.x * (256 / G) + threadIdx.x / G, l = threadIdx.x % G;
    const bool live = i < n;
    const int ii = live ? i : 0;           // (idle groups run the shuffles on row 0 and contribute nothing)
    float li = 0.f;
    {
        const float* z = logits + (size_t)ii * ldz;
        const int64_t lab = labels[ii];
        float mx = -INFINITY;
        for (int c = l; c < K; c += G) mx = fmaxf(mx, z[c]);
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, G));
        float se = 0.f, se_old = 0.f;
        for (int c = l; c < K; c += G) {
            const float e = expf(z[c] - mx);
            se += e;
            if (c <= n_old) se_old += e;
        }
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) { se += __shfl_xor(se, o, G); se_old += __shfl_xor(se_old, o, G); }
        const float lse = logf(se) + mx;
        const float inv_n = 1.f / (float)(*n_valid);
        if (lab >= 0 && l == 0) {
            if (!inclusive) {
                li = -(z[lab] - lse);
            } else if (lab == 0) {
                li = -((logf(se_old) + mx) - lse);      // :155 outputs[:,0] = lse(z[0..n_old]) - den
            } else if (lab > n_old) {
                li = -(z[lab] - lse);                   // :156
            } else {
                li = 0.f;                               // columns 1..n_old stay 0 (quirk 4)
            }
        }
        if (dz && live) {
            float* g = dz + (size_t)i * ldg;
            const float s = gscale * inv_n;
            for (int c = l; c < K; c += G) {
                float v = 0.f;
                if (lab >= 0) {
                    const float p = expf(z[c] - lse);
                    if (!inclusive) v = p - (c == lab);
                    else if (lab == 0) v = p - (c <= n_old ? expf(z[c] - mx) / se_old : 0.f);
                    else if (lab > n_old) v = p - (c == lab);
                }
                g[c] = v * s;
            }
        }
        li *= inv_n;
    }
    if (!live) li = 0.f;
    li = abr::block_sum<4>(li, sm);
    loss_add(li, loss_out, ws);
}

__global__ void count_valid_kernel(const int64_t* __restrict__ labels, int n, int* __restrict__ out) {
    __shared__ float sm[4];
    float c = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) c += labels[i] >= 0;
    c = abr::block_sum<4>(c, sm);
    if (threadIdx.x == 0) *out = max(1, (int)c);
}

// ------------------------------------------------------------------------------------------------
// RoI distillation -- distillation/distillation.py:164-240
// ------------------------------------------------------------------------------------------------
// 16 lanes per RoI (a 256-thread workgroup takes 16 RoIs): a lane owns the classes / box coordinates c = lane, lane + 16, ... and the
// per-RoI maxima and sums are xor-shuffle reductions inside the 16-lane group -- the kernel sits between the last forward GEMM and the
// first backward one, where one thread per RoI walking ~190 strided floats cost 130 us.
__global__ __launch_bounds__(256) void roi_distill_kernel(const float* __restrict__ z_s, const float* __restrict__ b_s,
                                                           const float* __restrict__ z_t, const float* __restrict__ b_t, int n,
                                                           int ld_zs, int ld_bs, int ld_zt, int ld_bt, int ld_dzt, int ld_dbt,
                                                           int K_old, int K_all, int dist_id, float* __restrict__ loss_out,
                                                           float gscale, float* __restrict__ d_zt, float* __restrict__ d_bt, const abr::DetWs ws) {
    __shared__ float sm[4];
    constexpr int G = 16;
    const int i = blockIdx.x * (256 / G) + threadIdx.x / G, l = threadIdx.x % G;
    auto gsum = [](float v) {
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, G);
        return v;
    };
    auto gmax = [](float v) {
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, G));
        return v;
    };
    float li = 0.f;
    const bool live = i < n;
    const int ii = live ? i : 0;           // (idle groups run the shuffles on RoI 0 and contribute nothing)
    {
        const float* zs = z_s + (size_t)ii * ld_zs;
        const float* zt = z_t + (size_t)ii * ld_zt;
        const float inv_n = 1.f / (float)n;
        if (dist_id) {
            float mt = -INFINITY, ms = -INFINITY;
            for (int c = l; c < K_all; c += G) mt = fmaxf(mt, zt[c]);
            for (int c = l; c < K_old; c += G) ms = fmaxf(ms, zs[c]);
            mt = gmax(mt);
            ms = gmax(ms);
            float set = 0.f, sebg = 0.f, ses = 0.f;
            for (int c = l; c < K_all; c += G) {
                const float e = expf(zt[c] - mt);
                set += e;
                if (c == 0 || c >= K_old) sebg += e;
            }
            for (int c = l; c < K_old; c += G) ses += expf(zs[c] - ms);
            set = gsum(set); sebg = gsum(sebg); ses = gsum(ses);
            const float den = logf(set) + mt;
            const float out_bg = (logf(sebg) + mt) - den;                       // :196
            const float lab0 = expf(zs[0] - ms) / ses;
            float acc = l == 0 ? lab0 * out_bg : 0.f;
            for (int c = l; c < K_old; c += G)
                if (c >= 1) acc += (expf(zs[c] - ms) / ses) * (zt[c] - den);    // :195,:198
            acc = gsum(acc);
            if (l == 0) li = -(acc / (float)K_old) * inv_n;                     // :198-199
            if (d_zt && live) {
                float* g = d_zt + (size_t)i * ld_dzt;
                const float s = -gscale * inv_n / (float)K_old;
                for (int c = l; c < K_all; c += G) {
                    const float e = expf(zt[c] - mt);
                    float v = -e / set;
                    if (c == 0 || c >= K_old) v += lab0 * e / sebg;
                    if (c >= 1 && c < K_old) v += expf(zs[c] - ms) / ses;
                    g[c] = v * s;
                }
            }
        } else {
            float mean_s = 0.f, mean_t = 0.f;
            for (int c = l; c < K_old; c += G) mean_s += zs[c];
            for (int c = l; c < K_all; c += G) mean_t += zt[c];
            mean_s = gsum(mean_s) / (float)K_old;
            mean_t = gsum(mean_t) / (float)K_all;
            float acc = 0.f, dsum = 0.f;
            for (int c = l; c < K_old; c += G) {
                const float d = (zt[c] - mean_t) - (zs[c] - mean_s);
                acc += d * d;
                dsum += d;
            }
            acc = gsum(acc);
            dsum = gsum(dsum);
            if (l == 0) li = acc / (float)K_old * inv_n;                        // :185-188
            if (d_zt && live) {
                float* g = d_zt + (size_t)i * ld_dzt;
                const float s = gscale * inv_n / (float)K_old;
                for (int c = l; c < K_all; c += G) {
                    float v = -2.f * dsum / (float)K_all;
                    if (c < K_old) v += 2.f * ((zt[c] - mean_t) - (zs[c] - mean_s));
                    g[c] = v * s;
                }
            }
        }
        // boxes: mean_n mean_k sum_4 (b_t[:,1:K_old] - b_s[:,1:])^2       :204-209
        const float* bs = b_s + (size_t)ii * ld_bs;
        const float* bt = b_t + (size_t)ii * ld_bt;
        const int kk = K_old - 1;
        float bacc = 0.f;
        const float bsc = kk > 0 ? inv_n / (float)kk : 0.f;
        for (int c = 4 + l; c < K_old * 4; c += G) {
            const float d = bt[c] - bs[c];
            bacc += d * d;
        }
        bacc = gsum(bacc);
        if (l == 0) li += bacc * bsc;
        if (d_bt && live) {
            float* g = d_bt + (size_t)i * ld_dbt;
            for (int c = l; c < K_all * 4; c += G)
                g[c] = (c >= 4 && c < K_old * 4) ? 2.f * (bt[c] - bs[c]) * bsc * gscale : 0.f;
        }
    }
    if (!live) li = 0.f;
    li = abr::block_sum<4>(li, sm);
    loss_add(li, loss_out, ws);
}

// ------------------------------------------------------------------------------------------------
// BCE-with-logits over sampled anchors -- modeling/rpn/loss.py:145-146
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bce_gather_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                          const int64_t* __restrict__ idx, const int64_t* __restrict__ yidx, int n_idx,
                                                          const float* __restrict__ denom_dev, float* __restrict__ loss_out,
                                                          float gscale, float* __restrict__ grad, const abr::DetWs ws) {
    __shared__ float sm[4];
    float acc = 0.f;
    const float inv = 1.f / (denom_dev ? fmaxf(*denom_dev, 1.f) : (float)n_idx);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_idx; i += gridDim.x * blockDim.x) {
        const int64_t j = idx[i];
        if (j < 0) continue;  // -1 padding of a fixed-size index list
        const float xv = x[j], yv = y[yidx ? yidx[i] : j];
        acc += fmaxf(xv, 0.f) - xv * yv + log1pf(expf(-fabsf(xv)));
        if (grad) grad[j] = (1.f / (1.f + expf(-xv)) - yv) * inv * gscale;
    }
    acc = abr::block_sum<4>(acc, sm);
    loss_add(acc * inv, loss_out, ws);
}

// ------------------------------------------------------------------------------------------------
// Ablation distillation losses -- distillation/distillation.py:18-84 (RPN) and :133-161 (backbone features)
// ------------------------------------------------------------------------------------------------
// pass 1 of the feature loss: sums of both maps (their means normalise the difference)
__global__ __launch_bounds__(256) void feat_distill_sums_kernel(const float* __restrict__ s, const float* __restrict__ t, int64_t n,
                                                                 float* __restrict__ stats, const abr::DetWs ws) {
    __shared__ float sm[4];
    float a = 0.f, b = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) { a += s[i]; b += t[i]; }
    a = abr::block_sum<4>(a, sm);
    b = abr::block_sum<4>(b, sm);
    if (ws.part) {
        const float in[2] = {a, b};
        float tot[2];
        if (abr::det_sum_last<2>(in, ws, tot) && threadIdx.x == 0) { stats[0] = tot[0]; stats[1] = tot[1]; }
    } else if (threadIdx.x == 0) { atomicAdd(stats, a); atomicAdd(stats + 1, b); }
}

// pass 2: loss = mean(max((s - mean s) - (t - mean t), 0)), and the number of positive differences (the mean's share of the gradient)
__global__ __launch_bounds__(256) void feat_distill_loss_kernel(const float* __restrict__ s, const float* __restrict__ t, int64_t n,
                                                                 float* __restrict__ stats, float* __restrict__ loss_out, const abr::DetWs ws) {
    __shared__ float sm[4];
    const float inv = 1.f / (float)n;
    const float shift = stats[1] * inv - stats[0] * inv;  // d_i = s_i - t_i + (mean t - mean s)
    float l = 0.f, c = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = s[i] - t[i] + shift;
        if (d > 0.f) { l += d; c += 1.f; }
    }
    l = abr::block_sum<4>(l, sm);
    c = abr::block_sum<4>(c, sm);
    if (ws.part) {
        const float in[2] = {l * inv, c};
        float tot[2];
        if (abr::det_sum_last<2>(in, ws, tot) && threadIdx.x == 0) { *loss_out = tot[0]; stats[2] = tot[1]; }
    } else if (threadIdx.x == 0) { atomicAdd(loss_out, l * inv); atomicAdd(stats + 2, c); }
}

// pass 3: d loss / d t_j = (1/n) * (mean(mask) - mask_j)
__global__ __launch_bounds__(256) void feat_distill_grad_kernel(const float* __restrict__ s, const float* __restrict__ t, int64_t n,
                                                                 const float* __restrict__ stats, float gscale, float* __restrict__ g) {
    const float inv = 1.f / (float)n;
    const float shift = stats[1] * inv - stats[0] * inv;
    const float mmean = stats[2] * inv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = s[i] - t[i] + shift;
        g[i] = gscale * inv * (mmean - (d > 0.f ? 1.f : 0.f));
    }
}

// RPN distillation on the NHWC head outputs: row r = (image, y, x), anchor a: objectness at obj[r*ld_o + a], deltas at reg[r*ld_r + 4a..]
//   cls  = mean_anchors max(o_s - o_t, 0)^2        ("filtered_l2")
//   bbox = mean_anchors [o_s - o_t > thr] * sum_4 (d_s - d_t)^2   ("l2" on the masked deltas; the mask carries no gradient)
__global__ __launch_bounds__(256) void rpn_distill_kernel(const float* __restrict__ obj_s, const float* __restrict__ reg_s, int ld_os, int ld_rs,
                                                           const float* __restrict__ obj_t, const float* __restrict__ reg_t, int ld_ot, int ld_rt,
                                                           int64_t rows, int A, float thr, int use_bbox, float* __restrict__ loss_out,
                                                           float gscale, float* __restrict__ g_obj, float* __restrict__ g_reg, int ld_go,
                                                           int ld_gr, const abr::DetWs ws) {
    __shared__ float sm[4];
    const int64_t n = rows * A;
    const float inv = 1.f / (float)n;
    float l = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / A;
        const int a = (int)(i % A);
        const float diff = obj_s[r * ld_os + a] - obj_t[r * ld_ot + a];
        const float pos = fmaxf(diff, 0.f);
        l += pos * pos;
        if (g_obj) g_obj[r * ld_go + a] = -2.f * pos * inv * gscale;
        const bool m = use_bbox && diff > thr;
        const float* ds = reg_s + r * ld_rs + 4 * a;
        const float* dt = reg_t + r * ld_rt + 4 * a;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float dd = m ? ds[e] - dt[e] : 0.f;
            l += dd * dd;
            if (g_reg) g_reg[r * ld_gr + 4 * a + e] = -2.f * dd * inv * gscale;
        }
    }
    l = abr::block_sum<4>(l, sm);
    loss_add(l * inv, loss_out, ws);
}

}  // namespace

extern "C" int abr_sigmoid_focal_forward(const float* logits, const int32_t* targets, int N, int C, float gamma,
                                         float alpha, float* losses, void* stream) {
    ABR_REQUIRE(N >= 0 && C > 0, "sigmoid_focal_forward: bad shape");
    if (N == 0) return ABR_OK;
    ABR_REQUIRE(logits && targets && losses, "sigmoid_focal_forward: null pointer");
    const int64_t total = (int64_t)N * C;
    focal_fwd<<<(unsigned)std::min<int64_t>((total + 255) / 256, 4096), 256, 0, abr::as_stream(stream)>>>(
        logits, targets, total, C, gamma, alpha, losses);
    ABR_CHECK_LAUNCH("sigmoid_focal_forward");
    return ABR_OK;
}

extern "C" int abr_sigmoid_focal_backward(const float* logits, const int32_t* targets, const float* d_losses, int N,
                                          int C, float gamma, float alpha, float* d_logits, void* stream) {
    ABR_REQUIRE(N >= 0 && C > 0, "sigmoid_focal_backward: bad shape");
    if (N == 0) return ABR_OK;
    ABR_REQUIRE(logits && targets && d_losses && d_logits, "sigmoid_focal_backward: null pointer");
    const int64_t total = (int64_t)N * C;
    focal_bwd<<<(unsigned)std::min<int64_t>((total + 255) / 256, 4096), 256, 0, abr::as_stream(stream)>>>(
        logits, targets, d_losses, total, C, gamma, alpha, d_logits);
    ABR_CHECK_LAUNCH("sigmoid_focal_backward");
    return ABR_OK;
}

extern "C" int abr_sigmoid_focal_forward_f64(const double* logits, const int32_t* targets, int N, int C, float gamma, float alpha, double* losses,
                                             void* stream) {
    ABR_REQUIRE(N >= 0 && C > 0, "sigmoid_focal_forward_f64: bad shape");
    if (N == 0) return ABR_OK;
    ABR_REQUIRE(logits && targets && losses, "sigmoid_focal_forward_f64: null pointer");
    const int64_t total = (int64_t)N * C;
    focal_fwd_f64<<<(unsigned)std::min<int64_t>((total + 255) / 256, 4096), 256, 0, abr::as_stream(stream)>>>(logits, targets, total, C, gamma, alpha, losses);
    ABR_CHECK_LAUNCH("sigmoid_focal_forward_f64");
    return ABR_OK;
}

extern "C" int abr_sigmoid_focal_backward_f64(const double* logits, const int32_t* targets, const double* d_losses, int N, int C, float gamma,
                                              float alpha, double* d_logits, void* stream) {
    ABR_REQUIRE(N >= 0 && C > 0, "sigmoid_focal_backward_f64: bad shape");
    if (N == 0) return ABR_OK;
    ABR_REQUIRE(logits && targets && d_losses && d_logits, "sigmoid_focal_backward_f64: null pointer");
    const int64_t total = (int64_t)N * C;
    focal_bwd_f64<<<(unsigned)std::min<int64_t>((total + 255) / 256, 4096), 256, 0, abr::as_stream(stream)>>>(logits, targets, d_losses, total, C, gamma, alpha,
                                                                                                          d_logits);
    ABR_CHECK_LAUNCH("sigmoid_focal_backward_f64");
    return ABR_OK;
}

static int zero_loss(float* loss_out, int nfloat, hipStream_t st, const char* who) {
    if (hipMemsetAsync(loss_out, 0, sizeof(float) * nfloat, st) != hipSuccess) {
        abr::set_error("%s: memset failed", who);
        return ABR_E_LAUNCH;
    }
    return ABR_OK;
}

extern "C" int abr_smooth_l1(const float* x, const float* t, int64_t n, float beta, float scale, float* loss_out,
                             float gscale, float* grad, void* stream) {
    ABR_REQUIRE(n >= 0 && loss_out, "smooth_l1: bad args");
    hipStream_t st = abr::as_stream(stream);
    if (int e = zero_loss(loss_out, 1, st, "smooth_l1")) return e;
    if (n == 0) return ABR_OK;
    ABR_REQUIRE(x && t, "smooth_l1: null pointer");
    const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, 1024);
    smooth_l1_kernel<<<grid, 256, 0, st>>>(x, t, n, beta, scale, loss_out, gscale, grad, abr::det_ws(st, grid));
    ABR_CHECK_LAUNCH("smooth_l1");
    return ABR_OK;
}

extern "C" int abr_smooth_l1_rows(const float* x, int x_cols, const float* t, const int64_t* rows, const int64_t* col0,
                                  const int64_t* trows, int n_rows, float beta, float scale, const float* denom_dev, float* loss_out,
                                  float gscale, float* grad, void* stream) {
    ABR_REQUIRE(n_rows >= 0 && loss_out && x_cols >= 4, "smooth_l1_rows: bad args");
    hipStream_t st = abr::as_stream(stream);
    if (int e = zero_loss(loss_out, 1, st, "smooth_l1_rows")) return e;
    if (n_rows == 0) return ABR_OK;
    ABR_REQUIRE(x && t && rows, "smooth_l1_rows: null pointer");
    const unsigned grid = abr::cdiv((int64_t)n_rows * 4, 256);
    smooth_l1_rows_kernel<<<grid, 256, 0, st>>>(x, x_cols, t, rows, col0, trows, n_rows, beta, scale, denom_dev, loss_out, gscale, grad, abr::det_ws(st, grid));
    ABR_CHECK_LAUNCH("smooth_l1_rows");
    return ABR_OK;
}

extern "C" int abr_softmax_ce(const float* logits, int ld_logits, const int64_t* labels, int n, int K, int inclusive, int n_old,
                              float* loss_out, float gscale, float* d_logits, int ld_dlogits, void* stream) {
    if (ld_logits <= 0) ld_logits = K;
    if (ld_dlogits <= 0) ld_dlogits = K;
    ABR_REQUIRE(n >= 0 && K > 0 && K <= kMaxK && loss_out, "softmax_ce: bad args (K<=128)");
    ABR_REQUIRE(!inclusive || (n_old >= 0 && n_old < K), "softmax_ce: n_old out of range");
    hipStream_t st = abr::as_stream(stream);
    // loss_out[0] = loss ; loss_out[1] reinterpret as int scratch for the valid-row count
    if (int e = zero_loss(loss_out, 2, st, "softmax_ce")) return e;
    if (n == 0) return ABR_OK;
    ABR_REQUIRE(logits && labels, "softmax_ce: null pointer");
    int* cnt = reinterpret_cast<int*>(loss_out + 1);
    count_valid_kernel<<<1, 256, 0, st>>>(labels, n, cnt);
    softmax_ce_kernel<<<abr::cdiv(n, 32), 256, 0, st>>>(logits, labels, n, K, ld_logits, ld_dlogits, inclusive, n_old, cnt, loss_out, gscale,
                                                          d_logits, abr::det_ws(st, abr::cdiv(n, 32)));
    ABR_CHECK_LAUNCH("softmax_ce");
    return ABR_OK;
}

extern "C" int abr_roi_distill(const float* z_s, const float* b_s, const float* z_t, const float* b_t, int n, int K_old,
                               int K_all, const int32_t* ld_host, int dist_id, float* loss_out, float gscale, float* d_zt,
                               float* d_bt, void* stream) {
    int ld[6] = {K_old, K_old * 4, K_all, K_all * 4, K_all, K_all * 4};
    if (ld_host)
        for (int i = 0; i < 6; i++)
            if (ld_host[i] > 0) ld[i] = ld_host[i];
    ABR_REQUIRE(n >= 0 && K_old > 0 && K_all >= K_old && K_all <= kMaxK && loss_out, "roi_distill: bad args");
    ABR_REQUIRE(!dist_id || K_all > K_old, "roi_distill: dist='id' needs K_all > K_old (empty slice in the reference)");
    hipStream_t st = abr::as_stream(stream);
    if (int e = zero_loss(loss_out, 1, st, "roi_distill")) return e;
    if (n == 0) return ABR_OK;
    ABR_REQUIRE(z_s && b_s && z_t && b_t, "roi_distill: null pointer");
    roi_distill_kernel<<<abr::cdiv(n, 16), 256, 0, st>>>(z_s, b_s, z_t, b_t, n, ld[0], ld[1], ld[2], ld[3], ld[4], ld[5], K_old, K_all, dist_id, loss_out,
                                                           gscale,
                                                           d_zt, d_bt, abr::det_ws(st, abr::cdiv(n, 16)));
    ABR_CHECK_LAUNCH("roi_distill");
    return ABR_OK;
}

extern "C" int abr_bce_logits_gather(const float* x, const float* y, const int64_t* idx, const int64_t* yidx, int n_idx,
                                     const float* denom_dev, float* loss_out,
                                     float gscale, float* grad, void* stream) {
    ABR_REQUIRE(n_idx >= 0 && loss_out, "bce_logits_gather: bad args");
    hipStream_t st = abr::as_stream(stream);
    if (int e = zero_loss(loss_out, 1, st, "bce_logits_gather")) return e;
    if (n_idx == 0) return ABR_OK;
    ABR_REQUIRE(x && y && idx, "bce_logits_gather: null pointer");
    const unsigned grid = std::min(abr::cdiv(n_idx, 256), 256u);
    bce_gather_kernel<<<grid, 256, 0, st>>>(x, y, idx, yidx, n_idx, denom_dev, loss_out, gscale, grad, abr::det_ws(st, grid));
    ABR_CHECK_LAUNCH("bce_logits_gather");
    return ABR_OK;
}

extern "C" int abr_feat_distill(const float* src, const float* tgt, int64_t n, float* loss_out, float* stats3, float gscale, float* d_tgt,
                                void* stream) {
    ABR_REQUIRE(n >= 0 && loss_out && stats3, "feat_distill: bad args");
    hipStream_t st = abr::as_stream(stream);
    if (int e = zero_loss(loss_out, 1, st, "feat_distill")) return e;
    if (hipMemsetAsync(stats3, 0, 3 * sizeof(float), st) != hipSuccess) return ABR_E_LAUNCH;
    if (n == 0) return ABR_OK;
    ABR_REQUIRE(src && tgt, "feat_distill: null pointer");
    const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, 2048);
    feat_distill_sums_kernel<<<grid, 256, 0, st>>>(src, tgt, n, stats3, abr::det_ws(st, 2 * (size_t)grid));
    feat_distill_loss_kernel<<<grid, 256, 0, st>>>(src, tgt, n, stats3, loss_out, abr::det_ws(st, 2 * (size_t)grid));
    if (d_tgt) feat_distill_grad_kernel<<<grid, 256, 0, st>>>(src, tgt, n, stats3, gscale, d_tgt);
    ABR_CHECK_LAUNCH("feat_distill");
    return ABR_OK;
}

extern "C" int abr_rpn_distill(const float* obj_s, const float* reg_s, int ld_obj_s, int ld_reg_s, const float* obj_t, const float* reg_t,
                               int ld_obj_t, int ld_reg_t, int64_t rows, int A, float bbox_threshold, int use_bbox, float* loss_out,
                               float gscale, float* d_obj_t, float* d_reg_t, int ld_d_obj, int ld_d_reg, void* stream) {
    ABR_REQUIRE(rows >= 0 && A > 0 && loss_out, "rpn_distill: bad args");
    hipStream_t st = abr::as_stream(stream);
    if (int e = zero_loss(loss_out, 1, st, "rpn_distill")) return e;
    if (rows == 0) return ABR_OK;
    ABR_REQUIRE(obj_s && reg_s && obj_t && reg_t, "rpn_distill: null pointer");
    ABR_REQUIRE((d_obj_t == nullptr) == (d_reg_t == nullptr), "rpn_distill: gradients come together");
    const unsigned grid = (unsigned)std::min<int64_t>((rows * A + 255) / 256, 2048);
    rpn_distill_kernel<<<grid, 256, 0, st>>>(
        obj_s, reg_s, ld_obj_s, ld_reg_s, obj_t, reg_t, ld_obj_t, ld_reg_t, rows, A, bbox_threshold, use_bbox, loss_out, gscale, d_obj_t,
        d_reg_t, ld_d_obj, ld_d_reg, abr::det_ws(st, grid));
    ABR_CHECK_LAUNCH("rpn_distill");
    return ABR_OK;
}
