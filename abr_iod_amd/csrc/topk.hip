// RPN proposal ranking: sigmoid + sorted top-k per image in ONE kernel.
//
// Reference: maskrcnn_benchmark/modeling/rpn/inference.py:87-96
//     objectness = permute_and_flatten(objectness, N, A, 1, H, W).view(N, -1).sigmoid()
//     objectness, topk_idx = objectness.topk(pre_nms_top_n, dim=1, sorted=True)
// (ATen: permute copy, sigmoid, radix-select + segmented sort = ~8 kernels).  Here the objectness logits are read in place
// from the fused NHWC head output (anchor j = row j/A, column j%A: already the flattened order):
//   0. (chip-wide kernel) sigmoid keys once into a scratch + 12-bit histogram;   then one 1024-thread workgroup per image:
//   1. radix select (12 + 12 + 8 bits, LDS histograms over the key scratch) of the k-th largest sigmoid value;
//   2. the survivors (all values >= threshold, ties included) are packed as 64-bit (score bits << 32 | ~index) words into LDS;
//   3. a bitonic sort of that LDS array (<= 16384 words = 128 KB of the CU's 160 KB) orders them by descending score,
//      equal scores by ascending index;
//   4. the first k are written out as fp32 scores + int64 indices.
// Scores are non-negative floats, so their bit patterns order like the values.
#include <algorithm>
#include <map>
#include <mutex>

#include "common.h"

namespace {

constexpr int TT = 1024;
constexpr int CAP = 16384;  // LDS sort capacity (64-bit words)

__device__ __forceinline__ unsigned score_bits(const float* __restrict__ base, int j, int A, int ld) {
    const float x = base[(size_t)(j / A) * ld + (j % A)];
    return __float_as_uint(1.f / (1.f + expf(-x)));
}

// Phase 1, chip-wide (grid = G slices x N images): the sigmoid key of every anchor, once, into a scratch + a histogram of the keys' top 12 bits
// (LDS per workgroup, non-empty bins flushed with global atomics).  One workgroup per image used to make FIVE passes over its 143 640 logits with an
// expf each (175 of the kernel's 290 us on one CU), and the step waits for about 60 % of this kernel's time (ABR_TOPK_TWICE probe, MEASUREMENTS.md).
constexpr int KT = 256;      // threads of the key kernel
constexpr int HB = 4096;     // bins of a 12-bit level

__global__ __launch_bounds__(KT) void topk_keys_kernel(const float* __restrict__ logits, int64_t img_stride, int n, int A, int ld, int per,
                                                       unsigned* __restrict__ keys, int* __restrict__ hist) {
    __shared__ int lh[HB];
    const int img = blockIdx.y;
    for (int i = threadIdx.x; i < HB; i += KT) lh[i] = 0;
    __syncthreads();
    const float* base = logits + (size_t)img * img_stride;
    const int j0 = blockIdx.x * per, j1 = min(j0 + per, n);
    for (int j = j0 + threadIdx.x; j < j1; j += KT) {
        const unsigned key = score_bits(base, j, A, ld);
        keys[(size_t)img * n + j] = key;
        atomicAdd(&lh[key >> 20], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < HB; i += KT)
        if (lh[i]) atomicAdd(&hist[img * HB + i], lh[i]);
}

// wave 0: the bin holding the s_krem-th largest element of histogram h (nbins a multiple of 64), counted from the TOP bin -> s_bin, s_krem -= what lies above it
__device__ __forceinline__ void find_bin_from_top(const int* h, int nbins, int* s_bin, int* s_krem) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x, per = nbins / 64;
        const int base_bin = (63 - lane) * per;          // lane 0 owns the top bins: a prefix scan over lanes is a suffix sum over bins
        int s = 0;
        for (int i = 0; i < per; i++) s += h[base_bin + i];
        int inc = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(inc, d, 64);
            if (lane >= d) inc += v;
        }
        const int exc = inc - s, krem = *s_krem;
        if (exc < krem && krem <= inc) {
            int cum = exc, b = base_bin + per - 1;
            for (; b > base_bin; b--) {
                if (cum + h[b] >= krem) break;
                cum += h[b];
            }
            *s_bin = b;
            *s_krem = krem - cum;
        }
    }
}

// Phase 2, one 1024-thread workgroup per image: exact k-th largest key by two more histogram levels over the key scratch (12 + 12 + 8 bits; plain
// coalesced dword reads, no expf), survivors packed into LDS, bitonic sort, emit.  Cleans the level-1 histogram for the next call.
__global__ __launch_bounds__(TT) void topk_select_sort_kernel(const unsigned* __restrict__ keys_all, int* __restrict__ hist_all, int n, int k,
                                                              float* __restrict__ scores, int64_t* __restrict__ idx) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* buf = reinterpret_cast<unsigned long long*>(smem);  // [CAP]
    __shared__ int h[HB];
    __shared__ int s_bin, s_krem, s_fill;
    const unsigned* keys = keys_all + (size_t)blockIdx.x * n;
    int* hist = hist_all + blockIdx.x * HB;
    for (int i = threadIdx.x; i < HB; i += TT) { h[i] = hist[i]; hist[i] = 0; }
    if (threadIdx.x == 0) { s_krem = k; s_fill = 0; s_bin = 0; }
    __syncthreads();
    unsigned thr = 0;
    if (k < n) {
        find_bin_from_top(h, HB, &s_bin, &s_krem);
        __syncthreads();
        const unsigned b1 = (unsigned)s_bin;
        for (int i = threadIdx.x; i < HB; i += TT) h[i] = 0;
        __syncthreads();
        for (int j = threadIdx.x; j < n; j += TT) {
            const unsigned key = keys[j];
            if ((key >> 20) == b1) atomicAdd(&h[(key >> 8) & (HB - 1)], 1);
        }
        __syncthreads();
        find_bin_from_top(h, HB, &s_bin, &s_krem);
        __syncthreads();
        const unsigned p24 = (b1 << 12) | (unsigned)s_bin;   // top 24 bits of the threshold
        for (int i = threadIdx.x; i < 256; i += TT) h[i] = 0;
        __syncthreads();
        for (int j = threadIdx.x; j < n; j += TT) {
            const unsigned key = keys[j];
            if ((key >> 8) == p24) atomicAdd(&h[key & 255], 1);
        }
        __syncthreads();
        find_bin_from_top(h, 256, &s_bin, &s_krem);
        __syncthreads();
        thr = (p24 << 8) | (unsigned)s_bin;
    }
    // pack every element with key >= thr (ties included; dropped only if the LDS capacity overflows)
    for (int j = threadIdx.x; j < n; j += TT) {
        const unsigned key = keys[j];
        if (key >= thr) {
            const int pos = atomicAdd(&s_fill, 1);
            if (pos < CAP) buf[pos] = ((unsigned long long)key << 32) | (unsigned)(~(unsigned)j);
        }
    }
    __syncthreads();
    const int filled = min(s_fill, CAP);
    int m = 1;
    while (m < filled) m <<= 1;
    for (int j = filled + threadIdx.x; j < m; j += TT) buf[j] = 0ull;  // padding sorts last
    __syncthreads();
    // bitonic sort, descending
    for (int size = 2; size <= m; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < (m >> 1); t += TT) {
                const int lo = 2 * t - (t & (stride - 1));  // index of the lower element of the pair
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const unsigned long long a = buf[lo], b = buf[hi];
                if ((a < b) == desc) { buf[lo] = b; buf[hi] = a; }
            }
            __syncthreads();
        }
    }
    for (int j = threadIdx.x; j < k; j += TT) {
        const unsigned long long v = j < filled ? buf[j] : 0ull;
        scores[(size_t)blockIdx.x * k + j] = __uint_as_float((unsigned)(v >> 32));
        idx[(size_t)blockIdx.x * k + j] = (int64_t)(~(unsigned)(v & 0xFFFFFFFFu));
    }
}

}  // namespace

extern "C" int abr_topk_sigmoid(const float* logits, int64_t img_stride, int N, int n, int A, int ld, int k, float* scores,
                                int64_t* idx, void* stream) {
    ABR_REQUIRE(N >= 0 && n > 0 && A >= 1 && ld >= A && k >= 0 && k <= n, "topk_sigmoid: bad args");
    ABR_REQUIRE(k <= CAP - 1024, "topk_sigmoid: k too large for the in-LDS sort (max 15360)");
    if (N == 0 || k == 0) return ABR_OK;
    ABR_REQUIRE(logits && scores && idx, "topk_sigmoid: null pointer");
    const size_t lds = (size_t)CAP * 8;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(topk_select_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    // scratch per stream (launches on a stream are ordered): level-1 histograms [cap_imgs][4096] at the FRONT, keys [N][n] behind them.  The histogram
    // region is zero on entry (allocation zero-fills it, phase 2 cleans what it used); its position does not depend on n or N, so a later call with a
    // smaller feature map / another batch size never finds stale key bits where it expects zeros.  Grow-only: keyed on capacity, not on N ==.
    struct Ws { void* p = nullptr; int cap_imgs = 0; size_t cap_keys = 0; };
    static std::mutex mu;
    static std::map<hipStream_t, Ws> pool;
    hipStream_t st = abr::as_stream(stream);
    unsigned* keys;
    int* hist;
    {
        std::lock_guard<std::mutex> g(mu);
        Ws& w = pool[st];
        const size_t need_keys = (size_t)N * n;
        if (w.cap_imgs < N || w.cap_keys < need_keys) {
            const int imgs = std::max(std::max(N, w.cap_imgs), 8);
            const size_t nkeys = std::max(need_keys, w.cap_keys);
            if (w.p) { (void)hipStreamSynchronize(st); (void)hipFree(w.p); w.p = nullptr; w.cap_imgs = 0; w.cap_keys = 0; }
            ABR_REQUIRE(hipMalloc(&w.p, (size_t)imgs * HB * 4 + nkeys * 4) == hipSuccess, "topk_sigmoid: scratch allocation failed");
            w.cap_imgs = imgs;
            w.cap_keys = nkeys;
            (void)hipMemsetAsync(w.p, 0, (size_t)imgs * HB * 4, st);
        }
        hist = static_cast<int*>(w.p);
        keys = reinterpret_cast<unsigned*>(static_cast<char*>(w.p) + (size_t)w.cap_imgs * HB * 4);
    }
    const int G = 64;
    const int per = ((n + G - 1) / G + KT - 1) / KT * KT;
    topk_keys_kernel<<<dim3((unsigned)((n + per - 1) / per), (unsigned)N), KT, 0, st>>>(logits, img_stride, n, A, ld, per, keys, hist);
    topk_select_sort_kernel<<<N, TT, lds, st>>>(keys, hist, n, k, scores, idx);
    ABR_CHECK_LAUNCH("topk_sigmoid");
    return ABR_OK;
}
