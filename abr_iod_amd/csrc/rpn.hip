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
__global__ void gt_max_iou_kernel(const float* __restrict__ boxes, int n, const float* __restrict__ gt, int G,
                                  unsigned* __restrict__ rowmax) {
    for (int g = 0; g < G; g++) {
        const float4 gb = reinterpret_cast<const float4*>(gt)[g];
        float m = 0.f;
        for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
            m = fmaxf(m, box_iou(gb, reinterpret_cast<const float4*>(boxes)[j]));
        m = abr::wave_max(m);
        if ((threadIdx.x & 63) == 0) atomicMax(rowmax + g, __float_as_uint(m));  // IoU >= 0: bit pattern is monotone
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
        gt_max_iou_kernel<<<std::min(abr::cdiv(n, 256), 512u), 256, 0, st>>>(boxes, n, gt, G, rowmax);
    }
    match_encode_kernel<<<abr::cdiv(n, 256), 256, 0, st>>>(boxes, n, gt, gt_labels, G, vis, hi, lo, allow_low_quality, rowmax,
                                                           wx, wy, ww, wh, matched, labels_f32, labels_i64, reg_targets);
    ABR_CHECK_LAUNCH("match_encode");
    return ABR_OK;
}
