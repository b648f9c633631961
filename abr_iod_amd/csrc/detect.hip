// Test-time detection post-processing.
//
// Reference: maskrcnn_benchmark/modeling/roi_heads/box_head/inference.py:43-151 (PostProcessor.forward / filter_results):
//   softmax over classes, BoxCoder.decode per class, clip_to_image, then PER IMAGE and PER CLASS: nonzero(score > thresh),
//   gather, _C.nms (sort + mask + host sweep), torch.full labels; cat; kthvalue on the HOST (.cpu()) for the top-100 cut.
//   With 21 classes and 4 images that is ~170 nonzero()/nms host syncs per batch.
//
// Here: 4 launches per BATCH, nothing leaves the device until the caller reads the detection counts.
//   det_softmax_decode_kernel   one thread per (proposal, class): prob + decoded + clipped box
//   det_sort_kernel             one workgroup per (class, image): threshold + in-LDS bitonic sort by (score desc, index asc),
//                               boxes gathered into sorted order
//   abr_nms_sorted_batched      the training path's NMS (nms.hip) over the N*C sorted lists at once
//   det_final_kernel            one workgroup per image: radix-select the detections_per_img-th largest kept score, then an
//                               ORDER-PRESERVING compaction (class-major, proposal order inside a class) of everything >= it.
#include "common.h"

extern "C" int64_t abr_nms_workspace_bytes(int N, int n_max);
extern "C" int abr_nms_sorted_batched(const float* boxes, const int32_t* counts, int N, int n_max, float thr, int strict_gt,
                                      int max_keep, int32_t* keep, int32_t* n_keep, void* workspace, int64_t workspace_bytes,
                                      void* stream);

namespace {

constexpr int TT = 1024;

#pragma clang fp contract(off)
__global__ void det_softmax_decode_kernel(const float* __restrict__ logits, int ld_logits, const float* __restrict__ deltas,
                                          int ld_deltas, int delta_col0, int agnostic_col, const float* __restrict__ rois, int K,
                                          int C, const int32_t* __restrict__ img_hw, float wx, float wy, float ww, float wh,
                                          float* __restrict__ prob, float* __restrict__ boxes) {
    const float clip = 4.135166556742356f;  // log(1000/16), box_coder.py:20
    const int total = K * C;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int r = t / C, j = t % C;
        // F.softmax(class_logits, -1)  (inference.py:56): exp(x - max) / sum
        const float* lr = logits + (size_t)r * ld_logits;
        float mx = lr[0];
        for (int c = 1; c < C; c++) mx = fmaxf(mx, lr[c]);
        float sum = 0.f;
        for (int c = 0; c < C; c++) sum += expf(lr[c] - mx);
        prob[t] = expf(lr[j] - mx) / sum;
        // BoxCoder.decode (box_coder.py:52-95) of class j's deltas against the proposal, then clip_to_image(remove_empty=False)
        const float* d = deltas + (size_t)r * ld_deltas + delta_col0 + (agnostic_col >= 0 ? agnostic_col : 4 * j);
        const float* roi = rois + (size_t)r * 5;
        const int img = (int)roi[0];
        const float w = roi[3] - roi[1] + 1, h = roi[4] - roi[2] + 1;
        const float cx = roi[1] + 0.5f * w, cy = roi[2] + 0.5f * h;
        const float dx = d[0] / wx, dy = d[1] / wy;
        const float dw = fminf(d[2] / ww, clip), dh = fminf(d[3] / wh, clip);
        const float pcx = dx * w + cx, pcy = dy * h + cy;
        const float pw = expf(dw) * w, ph = expf(dh) * h;
        const float W1 = (float)(img_hw[2 * img + 1] - 1), H1 = (float)(img_hw[2 * img] - 1);
        float4 o;
        o.x = fminf(fmaxf(pcx - 0.5f * pw, 0.f), W1);
        o.y = fminf(fmaxf(pcy - 0.5f * ph, 0.f), H1);
        o.z = fminf(fmaxf(pcx + 0.5f * pw - 1, 0.f), W1);
        o.w = fminf(fmaxf(pcy + 0.5f * ph - 1, 0.f), H1);
        reinterpret_cast<float4*>(boxes)[t] = o;
    }
}

// grid = (C, N).  prob [K,C], boxes [K,C,4]; rows of image i are row_off[i] .. row_off[i+1]-1.
// -> s_boxes/s_scores/s_idx [(i*C+j)*r_max + t] for t < counts[i*C+j]: class-j candidates with prob > thresh, descending score
//    (equal scores: ascending proposal index -- the stable order); s_idx = proposal index inside the image.
__global__ __launch_bounds__(TT) void det_sort_kernel(const float* __restrict__ prob, const float* __restrict__ boxes,
                                                      const int32_t* __restrict__ row_off, int C, int r_max, float thresh,
                                                      float* __restrict__ s_boxes, float* __restrict__ s_scores,
                                                      int32_t* __restrict__ s_idx, int32_t* __restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* buf = reinterpret_cast<unsigned long long*>(smem);
    __shared__ int s_cnt;
    const int j = blockIdx.x, i = blockIdx.y;
    const int r0 = row_off[i], n = min(row_off[i + 1] - r0, r_max);
    int m = 1;
    while (m < n) m <<= 1;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    int mine = 0;
    for (int t = threadIdx.x; t < m; t += TT) {
        unsigned long long key = 0ull;
        if (t < n) {
            const float p = prob[(size_t)(r0 + t) * C + j];
            if (p > thresh) {  // inference.py:116 `scores > self.score_thresh`; p > thresh >= 0 so the bit pattern is non-zero
                key = ((unsigned long long)__float_as_uint(p) << 32) | (unsigned)(~(unsigned)t);
                mine++;
            }
        }
        buf[t] = key;
    }
    if (mine) atomicAdd(&s_cnt, mine);
    __syncthreads();
    for (int size = 2; size <= m; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < (m >> 1); t += TT) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const unsigned long long a = buf[lo], b = buf[hi];
                if ((a < b) == desc) { buf[lo] = b; buf[hi] = a; }
            }
            __syncthreads();
        }
    }
    const int cnt = s_cnt;
    const size_t base = ((size_t)i * C + j) * r_max;
    for (int t = threadIdx.x; t < cnt; t += TT) {
        const unsigned long long v = buf[t];
        const int r = (int)(~(unsigned)(v & 0xFFFFFFFFu));
        s_scores[base + t] = __uint_as_float((unsigned)(v >> 32));
        s_idx[base + t] = r;
        reinterpret_cast<float4*>(s_boxes)[base + t] = reinterpret_cast<const float4*>(boxes)[(size_t)(r0 + r) * C + j];
    }
    if (threadIdx.x == 0) counts[i * C + j] = cnt;
}

// exclusive position of this thread's flag among the workgroup's flags, and the workgroup total
__device__ __forceinline__ int block_excl_scan(bool flag, int* wave_tot, int& total) {
    const unsigned long long b = __ballot(flag);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int in_wave = __popcll(b & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wave] = __popcll(b);
    __syncthreads();
    int before = 0, tot = 0;
    for (int w = 0; w < TT / 64; w++) {
        const int v = wave_tot[w];
        before += w < wave ? v : 0;
        tot += v;
    }
    __syncthreads();
    total = tot;
    return before + in_wave;
}

// grid = N.  filter_results' tail (inference.py:136-151): cat over classes 1..C-1, then if more than D detections keep those
// with score >= the D-th largest (kthvalue(total - D + 1)), ties included, order preserved.  Inside a class the reference's
// order is ASCENDING PROPOSAL INDEX, not score: both _C.nms variants return the kept indices sorted ascending
// (csrc/cpu/nms_cpu.cpp:64 nonzero(suppressed == 0); csrc/cuda/nms.cu:127-131 `.sort(0, false)`) and filter_results feeds them
// boxes in proposal order.  So each class scatters its survivors into an LDS slot table indexed by proposal and compacts that.
// Class 0's NMS survivors go out as the "background" list (the reference returns it next to the detections).
__global__ __launch_bounds__(TT) void det_final_kernel(const float* __restrict__ s_boxes, const float* __restrict__ s_scores,
                                                       const int32_t* __restrict__ s_idx, const int32_t* __restrict__ keep,
                                                       const int32_t* __restrict__ n_keep, int C, int r_max, int D, int cap,
                                                       float* __restrict__ out_boxes, float* __restrict__ out_scores,
                                                       int64_t* __restrict__ out_labels, int32_t* __restrict__ out_count,
                                                       float* __restrict__ bg_boxes, float* __restrict__ bg_scores,
                                                       int32_t* __restrict__ bg_count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* slot = reinterpret_cast<int*>(smem);  // [r_max]: 1 + sorted position of the survivor that came from proposal r, or 0
    __shared__ int hist[256];
    __shared__ int wave_tot[TT / 64];
    __shared__ unsigned s_prefix;
    __shared__ int s_krem;
    const int i = blockIdx.x;
    int total = 0;
    for (int j = 1; j < C; j++) total += n_keep[i * C + j];
    unsigned thr = 0;  // bit pattern of the cut score; 0 keeps everything (scores are > 0)
    if (D > 0 && total > D) {
        if (threadIdx.x == 0) { s_prefix = 0; s_krem = D; }
        __syncthreads();
        for (int pass = 3; pass >= 0; pass--) {
            for (int t = threadIdx.x; t < 256; t += TT) hist[t] = 0;
            __syncthreads();
            const unsigned prefix = s_prefix;
            const int shift = pass * 8;
            const unsigned hi_mask = pass == 3 ? 0u : (0xFFFFFFFFu << (shift + 8));
            for (int j = 1; j < C; j++) {
                const size_t base = ((size_t)i * C + j) * r_max;
                const int nk = n_keep[i * C + j];
                for (int t = threadIdx.x; t < nk; t += TT) {
                    const unsigned key = __float_as_uint(s_scores[base + keep[base + t]]);
                    if ((key & hi_mask) == (prefix & hi_mask)) atomicAdd(&hist[(key >> shift) & 255], 1);
                }
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                int krem = s_krem, b = 255, cum = 0;
                for (; b >= 0; b--) {
                    if (cum + hist[b] >= krem) break;
                    cum += hist[b];
                }
                s_prefix = prefix | ((unsigned)b << shift);
                s_krem = krem - cum;
            }
            __syncthreads();
        }
        thr = s_prefix;
    }
    int running = 0;
    for (int j = bg_count ? 0 : 1; j < C; j++) {
        const size_t base = ((size_t)i * C + j) * r_max;
        const int nk = n_keep[i * C + j];
        const unsigned cut = j == 0 ? 0u : thr;
        for (int r = threadIdx.x; r < r_max; r += TT) slot[r] = 0;
        __syncthreads();
        for (int t = threadIdx.x; t < nk; t += TT) {
            const int src = keep[base + t];
            if (__float_as_uint(s_scores[base + src]) >= cut) slot[s_idx[base + src]] = src + 1;
        }
        __syncthreads();
        int pos0 = j == 0 ? 0 : running;
        for (int r0 = 0; r0 < r_max; r0 += TT) {
            const int r = r0 + threadIdx.x;
            const int src1 = r < r_max ? slot[r] : 0;
            int tot;
            const int pos = pos0 + block_excl_scan(src1 != 0, wave_tot, tot);
            if (src1) {
                const float4 bx = reinterpret_cast<const float4*>(s_boxes)[base + src1 - 1];
                const float sc = s_scores[base + src1 - 1];
                if (j == 0) {
                    reinterpret_cast<float4*>(bg_boxes)[(size_t)i * r_max + pos] = bx;
                    bg_scores[(size_t)i * r_max + pos] = sc;
                } else if (pos < cap) {
                    const size_t o = (size_t)i * cap + pos;
                    reinterpret_cast<float4*>(out_boxes)[o] = bx;
                    out_scores[o] = sc;
                    out_labels[o] = j;
                }
            }
            pos0 += tot;
        }
        if (j == 0) {
            if (threadIdx.x == 0) bg_count[i] = pos0;
        } else {
            running = pos0;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out_count[i] = min(running, cap);
}

inline int64_t align256(int64_t v) { return (v + 255) & ~(int64_t)255; }

}  // namespace

extern "C" int abr_det_softmax_decode(const float* logits, int ld_logits, const float* deltas, int ld_deltas, int delta_col0,
                                      int agnostic_col, const float* rois, int K, int C, const int32_t* img_hw, float wx, float wy,
                                      float ww, float wh, float* prob, float* boxes, void* stream) {
    ABR_REQUIRE(K >= 0 && C >= 1 && ld_logits >= C && ld_deltas >= 4, "det_softmax_decode: bad shape");
    if (K == 0) return ABR_OK;
    ABR_REQUIRE(logits && deltas && rois && img_hw && prob && boxes, "det_softmax_decode: null pointer");
    det_softmax_decode_kernel<<<abr::cdiv((int64_t)K * C, 256), 256, 0, abr::as_stream(stream)>>>(
        logits, ld_logits, deltas, ld_deltas, delta_col0, agnostic_col, rois, K, C, img_hw, wx, wy, ww, wh, prob, boxes);
    ABR_CHECK_LAUNCH("det_softmax_decode");
    return ABR_OK;
}

extern "C" int64_t abr_det_select_workspace_bytes(int N, int C, int r_max) {
    const int64_t L = (int64_t)N * C;
    return align256(L * r_max * 16) + 3 * align256(L * r_max * 4) + 2 * align256(L * 4) +
           align256(abr_nms_workspace_bytes((int)L, r_max));
}

extern "C" int abr_det_select(const float* prob, const float* boxes, const int32_t* row_offsets, int N, int C, int r_max,
                              float score_thresh, float nms_thresh, int detections_per_img, int cap, float* out_boxes,
                              float* out_scores, int64_t* out_labels, int32_t* out_count, float* bg_boxes, float* bg_scores,
                              int32_t* bg_count, void* workspace, int64_t workspace_bytes, void* stream) {
    ABR_REQUIRE(N >= 0 && C >= 1 && r_max >= 0 && cap >= 0, "det_select: bad shape");
    ABR_REQUIRE(score_thresh >= 0.f, "det_select: score_thresh must be >= 0");
    ABR_REQUIRE(r_max <= 16384, "det_select: more than 16384 proposals per image do not fit the in-LDS sort");
    if (N == 0) return ABR_OK;
    ABR_REQUIRE(out_count, "det_select: null out_count");
    hipStream_t st = abr::as_stream(stream);
    if (r_max == 0) {  // no proposals at all: empty detections, empty background
        if (hipMemsetAsync(out_count, 0, 4 * (size_t)N, st) != hipSuccess) return ABR_E_LAUNCH;
        if (bg_count && hipMemsetAsync(bg_count, 0, 4 * (size_t)N, st) != hipSuccess) return ABR_E_LAUNCH;
        return ABR_OK;
    }
    ABR_REQUIRE(prob && boxes && row_offsets && workspace, "det_select: null pointer");
    ABR_REQUIRE(cap == 0 || (out_boxes && out_scores && out_labels), "det_select: null output");
    ABR_REQUIRE(!bg_count || (bg_boxes && bg_scores), "det_select: background outputs must come together");
    ABR_REQUIRE(workspace_bytes >= abr_det_select_workspace_bytes(N, C, r_max), "det_select: workspace too small");
    const int64_t L = (int64_t)N * C;
    unsigned char* w = (unsigned char*)workspace;
    float* s_boxes = (float*)w;      w += align256(L * r_max * 16);
    float* s_scores = (float*)w;     w += align256(L * r_max * 4);
    int32_t* s_idx = (int32_t*)w;    w += align256(L * r_max * 4);
    int32_t* counts = (int32_t*)w;   w += align256(L * 4);
    int32_t* keep = (int32_t*)w;     w += align256(L * r_max * 4);
    int32_t* n_keep = (int32_t*)w;   w += align256(L * 4);
    void* nms_ws = w;
    int m = 1;
    while (m < r_max) m <<= 1;
    const size_t lds = (size_t)m * 8;
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(det_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    det_sort_kernel<<<dim3(C, N), TT, lds, st>>>(prob, boxes, row_offsets, C, r_max, score_thresh, s_boxes, s_scores, s_idx, counts);
    ABR_CHECK_LAUNCH("det_sort");
    const int rc = abr_nms_sorted_batched(s_boxes, counts, (int)L, r_max, nms_thresh, 0, r_max, keep, n_keep, nms_ws,
                                          abr_nms_workspace_bytes((int)L, r_max), stream);
    if (rc != ABR_OK) return rc;
    const size_t lds_f = (size_t)r_max * 4;
    if (lds_f > 32 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(det_final_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f);
    det_final_kernel<<<N, TT, lds_f, st>>>(s_boxes, s_scores, s_idx, keep, n_keep, C, r_max, detections_per_img, cap, out_boxes, out_scores,
                                       out_labels, out_count, bg_boxes, bg_scores, bg_count);
    ABR_CHECK_LAUNCH("det_final");
    return ABR_OK;
}
