// RPN / RoI-head glue kernels: anchors, proposal decode+clip, IoU+Matcher+labels+BoxCoder.encode.
// In the reference these are dozens of small ATen index kernels per image (and several nonzero() host syncs);
// here each is one launch and nothing leaves the device.
#include "common.h"

namespace {

// anchor_generator.py:84-110 : (shifts[:,None] + cell[None]).reshape(-1,4), location-major / anchor-minor
__global__ void grid_anchors_kernel(const float* __restrict__ cell, int A, int H, int W, int stride, int img_h, int img_w,
                                    int straddle, float* __restrict__ out, uint8_t* __restrict__ vis) {
    const int total = H * W * A;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int a = i % A, x = (i / A) % W, y = i / (A * W);
        const float sx = (float)(x * stride), sy = (float)(y * stride);
        const float x1 = sx + cell[4 * a], y1 = sy + cell[4 * a + 1], x2 = sx + cell[4 * a + 2], y2 = sy + cell[4 * a + 3];
        reinterpret_cast<float4*>(out)[i] = make_float4(x1, y1, x2, y2);
        if (vis)
            vis[i] = straddle >= 0 ? (x1 >= -straddle && y1 >= -straddle && x2 < img_w + straddle && y2 < img_h + straddle) : 1;
    }
}

#pragma clang fp contract(off)
__global__ void decode_clip_kernel(const float* __restrict__ reg, int reg_stride, int reg_col0, int A,
                                   const float* __restrict__ anchors, const int64_t* __restrict__ idx, int N, int n_anchor,
                                   int k, const int32_t* __restrict__ img_hw, float wx, float wy, float ww, float wh,
                                   float* __restrict__ out) {
    const float clip = 4.135166556742356f;  // log(1000/16), box_coder.py:20
    const int total = N * k;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int i = t / k;
        const int64_t a = idx[t];
        // anchor a = location*A + a': its 4 deltas sit in row `location` of reg at columns reg_col0 + 4a' (NHWC head output)
        const float* d = reg + ((size_t)i * (n_anchor / A) + a / A) * reg_stride + reg_col0 + 4 * (a % A);
        const float4 b = reinterpret_cast<const float4*>(anchors)[a];
        const float w = b.z - b.x + 1, h = b.w - b.y + 1;               // box_coder.py:66-69
        const float cx = b.x + 0.5f * w, cy = b.y + 0.5f * h;
        const float dx = d[0] / wx, dy = d[1] / wy;
        const float dw = fminf(d[2] / ww, clip), dh = fminf(d[3] / wh, clip);
        const float pcx = dx * w + cx, pcy = dy * h + cy;
        const float pw = expf(dw) * w, ph = expf(dh) * h;
        float4 o = make_float4(pcx - 0.5f * pw, pcy - 0.5f * ph, pcx + 0.5f * pw - 1, pcy + 0.5f * ph - 1);
        if (img_hw) {                                                   // bounding_box.py:214-225 clip_to_image
            const float W1 = (float)(img_hw[2 * i + 1] - 1), H1 = (float)(img_hw[2 * i] - 1);
            o.x = fminf(fmaxf(o.x, 0.f), W1);
            o.y = fminf(fmaxf(o.y, 0.f), H1);
            o.z = fminf(fmaxf(o.z, 0.f), W1);
            o.w = fminf(fmaxf(o.w, 0.f), H1);
        }
        reinterpret_cast<float4*>(out)[t] = o;
    }
}

#pragma clang fp contract(off)
__device__ __forceinline__ float box_iou(const float4 g, const float4 b) {
    // structures/boxlist_ops.py:53-88, TO_REMOVE = 1
    const float area1 = (g.z - g.x + 1) * (g.w - g.y + 1);
    const float area2 = (b.z - b.x + 1) * (b.w - b.y + 1);
    const float lx = fmaxf(g.x, b.x), ly = fmaxf(g.y, b.y), rx = fminf(g.z, b.z), ry = fminf(g.w, b.w);
    const float w = fmaxf(rx - lx + 1, 0.f), h = fmaxf(ry - ly + 1, 0.f);
    const float inter = w * h;
    return inter / (area1 + area2 - inter);
}

// pass 1: per-gt maximum IoU over all boxes (needed only for Matcher.set_low_quality_matches_, matcher.py:83-112)
// maximum over a 256-thread workgroup (all threads call; s_m: 4 floats).  ONE atomic per workgroup and ground-truth box then goes to the
// shared word: with one per WAVE, the ~560 waves of the RPN call queued on a handful of addresses (80 us for 35 910 anchors x 5 boxes)
__device__ __forceinline__ float block_max_256(float m, float* s_m) {
    m = abr::wave_max(m);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    return fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
}

__global__ __launch_bounds__(256) void gt_max_iou_kernel(const float* __restrict__ boxes, int n, const float* __restrict__ gt, int G,
                                  unsigned* __restrict__ rowmax) {
    __shared__ float s_m[4];
    for (int g = 0; g < G; g++) {
        const float4 gb = reinterpret_cast<const float4*>(gt)[g];
        float m = 0.f;
        for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
            m = fmaxf(m, box_iou(gb, reinterpret_cast<const float4*>(boxes)[j]));
        m = block_max_256(m, s_m);
        if (threadIdx.x == 0) atomicMax(rowmax + g, __float_as_uint(m));  // IoU >= 0: bit pattern is monotone
    }
}

#pragma clang fp contract(off)
__global__ void match_encode_kernel(const float* __restrict__ boxes, int n, const float* __restrict__ gt,
                                    const int64_t* __restrict__ gt_labels, int G, const uint8_t* __restrict__ vis, float hi,
                                    float lo, int allow_lq, const unsigned* __restrict__ rowmax, float wx, float wy, float ww,
                                    float wh, int64_t* __restrict__ matched, float* __restrict__ labels_f32,
                                    int64_t* __restrict__ labels_i64, float* __restrict__ reg_targets) {
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const float4 b = reinterpret_cast<const float4*>(boxes)[j];
        float best = -1.f;
        int bi = 0;
        bool lq = false;
        for (int g = 0; g < G; g++) {
            const float v = box_iou(reinterpret_cast<const float4*>(gt)[g], b);
            if (v > best) { best = v; bi = g; }                                    // first max wins (torch.max)
            if (allow_lq && v == __uint_as_float(rowmax[g])) lq = true;
        }
        int64_t m = best < lo ? -1 : (best < hi ? -2 : bi);                         // matcher.py:68-75
        if (lq) m = bi;                                                             // :108-112
        if (matched) matched[j] = m;
        const int gi = m < 0 ? 0 : (int)m;                                          // clamp(min=0)
        if (labels_f32) {                                                           // rpn/loss.py:78-92
            float l = m >= 0 ? 1.f : 0.f;
            if (vis && !vis[j]) l = -1.f;
            if (m == -2) l = -1.f;
            labels_f32[j] = l;
        }
        if (labels_i64) {                                                           // box_head/loss.py:66-75
            int64_t l = gt_labels ? gt_labels[gi] : 1;
            if (m == -1) l = 0;
            if (m == -2) l = -1;
            labels_i64[j] = l;
        }
        if (reg_targets) {                                                          // box_coder.py:22-50
            const float4 r = reinterpret_cast<const float4*>(gt)[gi];
            const float ew = b.z - b.x + 1, eh = b.w - b.y + 1;
            const float ecx = b.x + 0.5f * ew, ecy = b.y + 0.5f * eh;
            const float gw = r.z - r.x + 1, gh = r.w - r.y + 1;
            const float gcx = r.x + 0.5f * gw, gcy = r.y + 0.5f * gh;
            float4 o;
            o.x = wx * (gcx - ecx) / ew;
            o.y = wy * (gcy - ecy) / eh;
            o.z = ww * logf(gw / ew);
            o.w = wh * logf(gh / eh);
            reinterpret_cast<float4*>(reg_targets)[j] = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Box-head training targets for the whole batch without leaving the device (abr_roi_head_targets):
//   candidate list of image i = its post-NMS proposals (rows keep[i, :n_keep[i]] of the decoded, score-sorted boxes) followed by
//   its ground-truth boxes (rpn/inference.py:53-74 add_gt_proposals); every candidate is matched against the image's GT
//   (Matcher 0.5 / 0.5, no low-quality rule), labelled and encoded (box_head/loss.py:56-84) -- the same arithmetic, in the same
//   order, as match_encode_kernel above.
#pragma clang fp contract(off)
__global__ void cand_match_kernel(const float* __restrict__ props, const float* __restrict__ scores, const int32_t* __restrict__ keep,
                                  const int32_t* __restrict__ n_keep, int k_pre, int post, const float* const* __restrict__ gt_ptrs,
                                  const int64_t* const* __restrict__ gt_label_ptrs, const int32_t* __restrict__ n_gt, int Pmax, float hi,
                                  float lo, float wx, float wy, float ww, float wh, float* __restrict__ cand,
                                  int64_t* __restrict__ labels_all, float* __restrict__ regt_all, float* __restrict__ obj_all,
                                  int32_t* __restrict__ n_cand, float* __restrict__ n_valid) {
    const int i = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int nk = min(n_keep[i], post), G = n_gt[i];
    const int nc = nk + G;
    if (j == 0) {
        n_cand[i] = nc;
        if (i == 0) *n_valid = 0.f;   // (accumulated by roi_merge_gather_kernel, a later launch on the same stream)
    }
    if (j >= Pmax) return;
    const float* gt = gt_ptrs[i];
    const int64_t* gt_labels = gt_label_ptrs[i];
    const int64_t o = (int64_t)i * Pmax + j;
    if (j >= nc) {
        labels_all[o] = -1;   // never sampled
        obj_all[o] = 0.f;
        reinterpret_cast<float4*>(cand)[o] = make_float4(0.f, 0.f, 0.f, 0.f);
        reinterpret_cast<float4*>(regt_all)[o] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const int src = j < nk ? keep[(int64_t)i * post + j] : 0;
    const float4 b = j < nk ? reinterpret_cast<const float4*>(props)[(int64_t)i * k_pre + src] : reinterpret_cast<const float4*>(gt)[j - nk];
    obj_all[o] = j < nk ? scores[(int64_t)i * k_pre + src] : 1.f;               // GT boxes join with objectness 1 (inference.py:66-68)
    float best = -1.f;
    int bi = 0;
    for (int g = 0; g < G; g++) {
        const float v = box_iou(reinterpret_cast<const float4*>(gt)[g], b);
        if (v > best) { best = v; bi = g; }                                    // first max wins (torch.max)
    }
    const int64_t m = best < lo ? -1 : (best < hi ? -2 : bi);                   // matcher.py:68-75
    const int gi = m < 0 ? 0 : (int)m;                                          // clamp(min=0)
    int64_t l = gt_labels[gi];                                                  // box_head/loss.py:66-75
    if (m == -1) l = 0;
    if (m == -2) l = -1;
    labels_all[o] = l;
    const float4 r = reinterpret_cast<const float4*>(gt)[gi];                   // box_coder.py:22-50
    const float ew = b.z - b.x + 1, eh = b.w - b.y + 1;
    const float ecx = b.x + 0.5f * ew, ecy = b.y + 0.5f * eh;
    const float gw = r.z - r.x + 1, gh = r.w - r.y + 1;
    const float gcx = r.x + 0.5f * gw, gcy = r.y + 0.5f * gh;
    float4 t;
    t.x = wx * (gcx - ecx) / ew;
    t.y = wy * (gcy - ecy) / eh;
    t.z = ww * logf(gw / ew);
    t.w = wh * logf(gh / eh);
    reinterpret_cast<float4*>(cand)[o] = b;
    reinterpret_cast<float4*>(regt_all)[o] = t;
}

__device__ __forceinline__ int lower_bound_i64(const int64_t* a, int n, int64_t v) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// The sampler's two ascending index lists of image i merged into ONE ascending list (box_head/loss.py:108-114 takes
// nonzero(pos_mask | neg_mask), i.e. ascending candidate order) and the sampled rows gathered: rois (batch index, box), labels,
// regression targets.  Rows past the number drawn are padding: label -1 (ignored by the loss kernels), a zero box of image i.
__global__ __launch_bounds__(256) void roi_merge_gather_kernel(const float* __restrict__ cand, const int64_t* __restrict__ labels_all,
                                                               const float* __restrict__ regt_all, const float* __restrict__ obj_all, int Pmax,
                                                               const int64_t* __restrict__ pos_idx, int max_pos,
                                                               const int64_t* __restrict__ neg_idx, int batch,
                                                               const int32_t* __restrict__ counts, float* __restrict__ rois,
                                                               int64_t* __restrict__ labels, float* __restrict__ reg_targets,
                                                               int64_t* __restrict__ sampled_idx, float* __restrict__ obj,
                                                               int64_t* __restrict__ pos_rows, int64_t* __restrict__ col0, int num_classes,
                                                               int cls_agnostic, float* __restrict__ n_valid) {
    const int i = blockIdx.x;
    const int cp = counts[2 * i], cn = counts[2 * i + 1];
    const int64_t* pos = pos_idx + (int64_t)i * max_pos;
    const int64_t* neg = neg_idx + (int64_t)i * batch;
    for (int e = threadIdx.x; e < batch; e += blockDim.x) {
        int64_t v = -1;
        int rank;
        if (e < cp) {
            v = pos[e];
            rank = e + lower_bound_i64(neg, cn, v);
        } else if (e < cp + cn) {
            v = neg[e - cp];
            rank = (e - cp) + lower_bound_i64(pos, cp, v);
        } else {
            rank = e;   // padding rows keep their place behind the drawn ones
        }
        const int64_t row = (int64_t)i * batch + rank;
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f), t = b;
        int64_t l = -1;
        float ob = 0.f;
        if (v >= 0) {
            const int64_t o = (int64_t)i * Pmax + v;
            b = reinterpret_cast<const float4*>(cand)[o];
            t = reinterpret_cast<const float4*>(regt_all)[o];
            l = labels_all[o];
            ob = obj_all[o];
        }
        obj[row] = ob;
        pos_rows[row] = l > 0 ? row : -1;                                      // rows entering the box-regression loss (loss.py:166)
        col0[row] = num_classes + (cls_agnostic ? 4 : 4 * (l > 0 ? l : 0));    // their 4 columns of the fused predictor output (:168-171)
        float* r = rois + row * 5;
        r[0] = (float)i; r[1] = b.x; r[2] = b.y; r[3] = b.z; r[4] = b.w;
        labels[row] = l;
        reinterpret_cast<float4*>(reg_targets)[row] = t;
        sampled_idx[row] = v;
    }
    if (threadIdx.x == 0) atomicAdd(n_valid, (float)(cp + cn));
}

// rows picks[i*P + j] of image i's post-NMS list -> rois [N*P, 5] = (i, box), obj [N*P] (the 64 distillation proposals of the source
// model: generalized_rcnn.py:140-158; the post-NMS list is already in descending objectness order)
__global__ void gather_proposals_kernel(const float* __restrict__ props, const float* __restrict__ scores, const int32_t* __restrict__ keep,
                                        int k_pre, int post, const int64_t* __restrict__ picks, int N, int P, float* __restrict__ rois,
                                        float* __restrict__ obj) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * P) return;
    const int i = t / P;
    const int src = keep[(int64_t)i * post + picks[t]];
    const float4 b = reinterpret_cast<const float4*>(props)[(int64_t)i * k_pre + src];
    float* r = rois + (int64_t)t * 5;
    r[0] = (float)i; r[1] = b.x; r[2] = b.y; r[3] = b.z; r[4] = b.w;
    if (obj) obj[t] = scores[(int64_t)i * k_pre + src];
}

// ---------------------------------------------------------------------------------------------------------------------------------
// RPN training targets for the whole batch (abr_rpn_targets_batched): the anchors are shared by the images (same feature-map size), the
// GT boxes and the visibility masks are per image.  Same arithmetic as gt_max_iou_kernel / match_encode_kernel, blockIdx.y = image.
__global__ __launch_bounds__(256) void gt_max_iou_batched_kernel(const float* __restrict__ boxes, int n, const float* const* __restrict__ gt_ptrs,
                                          const int32_t* __restrict__ n_gt, int g_max, unsigned* __restrict__ rowmax) {
    __shared__ float s_m[4];
    const int i = blockIdx.y;
    const float* gt = gt_ptrs[i];
    const int G = n_gt[i];
    for (int g = 0; g < G; g++) {
        const float4 gb = reinterpret_cast<const float4*>(gt)[g];
        float m = 0.f;
        for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
            m = fmaxf(m, box_iou(gb, reinterpret_cast<const float4*>(boxes)[j]));
        m = block_max_256(m, s_m);
        if (threadIdx.x == 0) atomicMax(rowmax + (size_t)i * g_max + g, __float_as_uint(m));
    }
}

#pragma clang fp contract(off)
__global__ void rpn_match_batched_kernel(const float* __restrict__ boxes, int n, const float* const* __restrict__ gt_ptrs,
                                         const int32_t* __restrict__ n_gt, int g_max, const uint8_t* const* __restrict__ vis_ptrs, float hi,
                                         float lo, const unsigned* __restrict__ rowmax, float wx, float wy, float ww, float wh,
                                         float* __restrict__ labels, float* __restrict__ reg_targets) {
    const int i = blockIdx.y;
    const float* gt = gt_ptrs[i];
    const uint8_t* vis = vis_ptrs[i];
    const int G = n_gt[i];
    const unsigned* rm = rowmax + (size_t)i * g_max;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const float4 b = reinterpret_cast<const float4*>(boxes)[j];
        float best = -1.f;
        int bi = 0;
        bool lq = false;
        for (int g = 0; g < G; g++) {
            const float v = box_iou(reinterpret_cast<const float4*>(gt)[g], b);
            if (v > best) { best = v; bi = g; }
            if (v == __uint_as_float(rm[g])) lq = true;                            // allow_low_quality_matches (matcher.py:83-112)
        }
        int64_t m = best < lo ? -1 : (best < hi ? -2 : bi);
        if (lq) m = bi;
        const int gi = m < 0 ? 0 : (int)m;
        float l = m >= 0 ? 1.f : 0.f;                                               // rpn/loss.py:78-92
        if (vis && !vis[j]) l = -1.f;
        if (m == -2) l = -1.f;
        labels[(size_t)i * n + j] = l;
        const float4 r = reinterpret_cast<const float4*>(gt)[gi];
        const float ew = b.z - b.x + 1, eh = b.w - b.y + 1;
        const float ecx = b.x + 0.5f * ew, ecy = b.y + 0.5f * eh;
        const float gw = r.z - r.x + 1, gh = r.w - r.y + 1;
        const float gcx = r.x + 0.5f * gw, gcy = r.y + 0.5f * gh;
        float4 o;
        o.x = wx * (gcx - ecx) / ew;
        o.y = wy * (gcy - ecy) / eh;
        o.z = ww * logf(gw / ew);
        o.w = wh * logf(gh / eh);
        reinterpret_cast<float4*>(reg_targets)[(size_t)i * n + j] = o;
    }
}

// RPN loss bookkeeping in one launch: the sampler's two padded lists -> the concatenated "all sampled" list, the flat positions of their
// objectness logits and box deltas inside the fused NHWC head output (anchor j of the flattened batch: row j / A, columns j % A and
// A + 4 (j % A) of a row of Cf columns), and the number of sampled anchors as a float (the losses' normaliser, rpn/loss.py:136,146).
__global__ void rpn_loss_indices_kernel(const int64_t* __restrict__ pos, int n_pos, const int64_t* __restrict__ neg, int n_neg,
                                        const int32_t* __restrict__ counts, int n_img, int A, int Cf, int64_t* __restrict__ samp,
                                        int64_t* __restrict__ obj_flat, int64_t* __restrict__ pos_row, int64_t* __restrict__ pos_col,
                                        float* __restrict__ denom) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_pos + n_neg) {
        const int64_t v = t < n_pos ? pos[t] : neg[t - n_pos];
        samp[t] = v;
        const int64_t row = v >= 0 ? v / A : -1;
        obj_flat[t] = v >= 0 ? row * Cf + (v - row * A) : -1;
        if (t < n_pos) {
            pos_row[t] = row;
            pos_col[t] = v >= 0 ? A + 4 * (v - row * A) : 0;
        }
    }
    if (t == 0) {
        int c = 0;
        for (int i = 0; i < 2 * n_img; i++) c += counts[i];
        *denom = (float)c;
    }
}

// BoxCoder.encode row-wise (box_coder.py:22-50): out[i] = encode(gt[i], ex[i])
#pragma clang fp contract(off)
__global__ void box_encode_kernel(const float* __restrict__ gt, const float* __restrict__ ex, int n, float wx, float wy,
                                  float ww, float wh, float* __restrict__ out) {
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const float4 r = reinterpret_cast<const float4*>(gt)[j], b = reinterpret_cast<const float4*>(ex)[j];
        const float ew = b.z - b.x + 1, eh = b.w - b.y + 1;
        const float ecx = b.x + 0.5f * ew, ecy = b.y + 0.5f * eh;
        const float gw = r.z - r.x + 1, gh = r.w - r.y + 1;
        const float gcx = r.x + 0.5f * gw, gcy = r.y + 0.5f * gh;
        reinterpret_cast<float4*>(out)[j] = make_float4(wx * (gcx - ecx) / ew, wy * (gcy - ecy) / eh, ww * logf(gw / ew), wh * logf(gh / eh));
    }
}

}  // namespace

extern "C" int abr_box_encode(const float* gt, const float* ex, int n, float wx, float wy, float ww, float wh, float* out,
                              void* stream) {
    ABR_REQUIRE(n >= 0, "box_encode: bad n");
    if (n == 0) return ABR_OK;
    ABR_REQUIRE(gt && ex && out, "box_encode: null pointer");
    box_encode_kernel<<<abr::cdiv(n, 256), 256, 0, abr::as_stream(stream)>>>(gt, ex, n, wx, wy, ww, wh, out);
    ABR_CHECK_LAUNCH("box_encode");
    return ABR_OK;
}

extern "C" int abr_grid_anchors(const float* cell, int A, int H, int W, int stride, int img_h, int img_w, int straddle,
                                float* out, uint8_t* vis, void* stream) {
    ABR_REQUIRE(cell && out && A > 0 && H > 0 && W > 0, "grid_anchors: bad args");
    grid_anchors_kernel<<<abr::cdiv((int64_t)H * W * A, 256), 256, 0, abr::as_stream(stream)>>>(cell, A, H, W, stride, img_h,
                                                                                                  img_w, straddle, out, vis);
    ABR_CHECK_LAUNCH("grid_anchors");
    return ABR_OK;
}

extern "C" int abr_rpn_decode_clip(const float* reg, int reg_stride, int reg_col0, int A, const float* anchors, const int64_t* idx,
                                   int N, int n_anchor, int k, const int32_t* img_hw, float wx, float wy, float ww, float wh,
                                   float* out, void* stream) {
    ABR_REQUIRE(N >= 0 && k >= 0 && reg_stride >= 4 && A >= 1 && n_anchor % A == 0, "rpn_decode_clip: bad args");
    if (N == 0 || k == 0) return ABR_OK;
    ABR_REQUIRE(reg && anchors && idx && out, "rpn_decode_clip: null pointer");
    decode_clip_kernel<<<abr::cdiv((int64_t)N * k, 256), 256, 0, abr::as_stream(stream)>>>(
        reg, reg_stride, reg_col0, A, anchors, idx, N, n_anchor, k, img_hw, wx, wy, ww, wh, out);
    ABR_CHECK_LAUNCH("rpn_decode_clip");
    return ABR_OK;
}

extern "C" int64_t abr_match_workspace_bytes(int n, int G) { return (int64_t)(G > 0 ? G : 1) * 4; }

extern "C" int abr_match_encode(const float* boxes, int n, const float* gt, const int64_t* gt_labels, int G,
                                const uint8_t* vis, float hi, float lo, int allow_low_quality, float wx, float wy, float ww,
                                float wh, int64_t* matched, float* labels_f32, int64_t* labels_i64, float* reg_targets,
                                void* workspace, int64_t workspace_bytes, void* stream) {
    ABR_REQUIRE(n >= 0, "match_encode: bad n");
    // Matcher raises on empty GT (matcher.py:53-62 "No ground-truth boxes available for one of the images")
    ABR_REQUIRE(G > 0, "match_encode: no ground-truth boxes available for one of the images during training");
    if (n == 0) return ABR_OK;
    ABR_REQUIRE(boxes && gt, "match_encode: null pointer");
    hipStream_t st = abr::as_stream(stream);
    unsigned* rowmax = nullptr;
    if (allow_low_quality) {
        ABR_REQUIRE(workspace && workspace_bytes >= abr_match_workspace_bytes(n, G), "match_encode: workspace too small");
        rowmax = (unsigned*)workspace;
        if (hipMemsetAsync(rowmax, 0, 4 * (size_t)G, st) != hipSuccess) return ABR_E_LAUNCH;
        gt_max_iou_kernel<<<std::min(abr::cdiv(n, 1024), 64u), 256, 0, st>>>(boxes, n, gt, G, rowmax);
    }
    match_encode_kernel<<<abr::cdiv(n, 256), 256, 0, st>>>(boxes, n, gt, gt_labels, G, vis, hi, lo, allow_low_quality, rowmax,
                                                           wx, wy, ww, wh, matched, labels_f32, labels_i64, reg_targets);
    ABR_CHECK_LAUNCH("match_encode");
    return ABR_OK;
}

extern "C" int abr_sample_pos_neg(const void* labels, int labels_are_int64, int N, int n, int64_t stride, int batch_size, int max_pos,
                                  uint64_t seed, int first_image, int64_t index_offset_per_image, int64_t* pos_idx,
                                  int64_t* neg_idx, int32_t* counts, void* stream);

extern "C" int abr_roi_head_targets(const float* props, const int32_t* keep, const int32_t* n_keep, int N, int k_pre, int post,
                                    const float* const* gt_ptrs, const int64_t* const* gt_label_ptrs, const int32_t* n_gt, int g_max,
                                    float hi, float lo, float wx, float wy, float ww, float wh, int batch_size, int max_pos,
                                    uint64_t seed, float* cand, int64_t* labels_all, float* regt_all, int32_t* n_cand, int64_t* pos_idx,
                                    int64_t* neg_idx, int32_t* counts, float* rois, int64_t* labels, float* reg_targets,
                                    int64_t* sampled_idx, float* n_valid, const float* scores, float* obj_all, float* obj,
                                    int64_t* pos_rows, int64_t* col0, int num_classes, int cls_agnostic, void* stream) {
    ABR_REQUIRE(N > 0 && k_pre > 0 && post > 0 && g_max > 0 && batch_size > 0 && max_pos >= 0 && max_pos <= batch_size,
                "roi_head_targets: bad sizes");
    ABR_REQUIRE(props && keep && n_keep && gt_ptrs && gt_label_ptrs && n_gt && cand && labels_all && regt_all && n_cand && pos_idx && neg_idx &&
                counts && rois && labels && reg_targets && sampled_idx && n_valid && scores && obj_all && obj && pos_rows && col0,
                "roi_head_targets: null pointer");
    hipStream_t st = abr::as_stream(stream);
    const int Pmax = post + g_max;
    cand_match_kernel<<<dim3(abr::cdiv(Pmax, 256), N), 256, 0, st>>>(props, scores, keep, n_keep, k_pre, post, gt_ptrs, gt_label_ptrs, n_gt, Pmax,
                                                                     hi, lo, wx, wy, ww, wh, cand, labels_all, regt_all, obj_all, n_cand, n_valid);
    ABR_CHECK_LAUNCH("roi_head_targets (match)");
    const int rc = abr_sample_pos_neg(labels_all, 1, N, Pmax, Pmax, batch_size, max_pos, seed, 0, 0, pos_idx, neg_idx, counts, stream);
    if (rc != ABR_OK) return rc;
    roi_merge_gather_kernel<<<N, 256, 0, st>>>(cand, labels_all, regt_all, obj_all, Pmax, pos_idx, max_pos, neg_idx, batch_size, counts, rois, labels,
                                               reg_targets, sampled_idx, obj, pos_rows, col0, num_classes, cls_agnostic, n_valid);
    ABR_CHECK_LAUNCH("roi_head_targets (gather)");
    return ABR_OK;
}

extern "C" int abr_gather_proposals(const float* props, const float* scores, const int32_t* keep, int N, int k_pre, int post,
                                    const int64_t* picks, int P, float* rois, float* obj, void* stream) {
    ABR_REQUIRE(N >= 0 && P >= 0 && k_pre > 0 && post > 0, "gather_proposals: bad sizes");
    if (N * P == 0) return ABR_OK;
    ABR_REQUIRE(props && keep && picks && rois && (!obj || scores), "gather_proposals: null pointer");
    gather_proposals_kernel<<<abr::cdiv((int64_t)N * P, 256), 256, 0, abr::as_stream(stream)>>>(props, scores, keep, k_pre, post, picks, N, P, rois, obj);
    ABR_CHECK_LAUNCH("gather_proposals");
    return ABR_OK;
}

extern "C" int abr_rpn_targets_batched(const float* anchors, int n, int N, const float* const* gt_ptrs, const int32_t* n_gt, int g_max,
                                       const uint8_t* const* vis_ptrs, float hi, float lo, float wx, float wy, float ww, float wh,
                                       float* labels, float* reg_targets, void* workspace, int64_t workspace_bytes, void* stream) {
    ABR_REQUIRE(n >= 0 && N > 0 && g_max > 0, "rpn_targets_batched: bad sizes");
    if (n == 0) return ABR_OK;
    ABR_REQUIRE(anchors && gt_ptrs && n_gt && vis_ptrs && labels && reg_targets && workspace, "rpn_targets_batched: null pointer");
    ABR_REQUIRE(workspace_bytes >= (int64_t)N * g_max * 4, "rpn_targets_batched: workspace too small");
    hipStream_t st = abr::as_stream(stream);
    unsigned* rowmax = (unsigned*)workspace;
    if (hipMemsetAsync(rowmax, 0, 4 * (size_t)N * g_max, st) != hipSuccess) return ABR_E_LAUNCH;
    gt_max_iou_batched_kernel<<<dim3(std::min(abr::cdiv(n, 1024), 64u), N), 256, 0, st>>>(anchors, n, gt_ptrs, n_gt, g_max, rowmax);
    rpn_match_batched_kernel<<<dim3(abr::cdiv(n, 256), N), 256, 0, st>>>(anchors, n, gt_ptrs, n_gt, g_max, vis_ptrs, hi, lo, rowmax, wx, wy, ww, wh,
                                                                          labels, reg_targets);
    ABR_CHECK_LAUNCH("rpn_targets_batched");
    return ABR_OK;
}

extern "C" int abr_rpn_loss_indices(const int64_t* pos, int n_pos, const int64_t* neg, int n_neg, const int32_t* counts, int n_img, int A,
                                    int Cf, int64_t* samp, int64_t* obj_flat, int64_t* pos_row, int64_t* pos_col, float* denom, void* stream) {
    ABR_REQUIRE(n_pos >= 0 && n_neg >= 0 && n_img > 0 && A > 0 && Cf >= 5 * A, "rpn_loss_indices: bad sizes");
    ABR_REQUIRE(pos && neg && counts && samp && obj_flat && pos_row && pos_col && denom, "rpn_loss_indices: null pointer");
    rpn_loss_indices_kernel<<<abr::cdiv(std::max(n_pos + n_neg, 1), 256), 256, 0, abr::as_stream(stream)>>>(pos, n_pos, neg, n_neg, counts, n_img, A, Cf,
                                                                                                             samp, obj_flat, pos_row, pos_col, denom);
    ABR_CHECK_LAUNCH("rpn_loss_indices");
    return ABR_OK;
}
