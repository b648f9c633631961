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

// raw = 0: the key is the bit pattern of sigmoid(x) (non-negative: orders like the value).  raw = 1 (abr_sort_scores_desc: the score sort of
// _C.nms): the key orders like x itself for ANY finite float -- sign bit flipped for x >= 0, all bits for x < 0 (-0 sorts below +0, NaNs at the ends)
__device__ __forceinline__ unsigned score_bits(const float* __restrict__ base, int j, int A, int ld, int raw) {
    const float x = base[(size_t)(j / A) * ld + (j % A)];
    if (raw) {
        const unsigned b = __float_as_uint(x);
        return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    }
    return __float_as_uint(1.f / (1.f + expf(-x)));
}

// Phase 1, chip-wide (grid = G slices x N images): the sigmoid key of every anchor, once, into a scratch + a histogram of the keys' top 12 bits
// (LDS per workgroup, non-empty bins flushed with global atomics).  One workgroup per image used to make FIVE passes over its 143 640 logits with an
// expf each (175 of the kernel's 290 us on one CU), and the step waits for about 60 % of this kernel's time (ABR_TOPK_TWICE probe, MEASUREMENTS.md).
constexpr int KT = 256;      // threads of the key kernel
constexpr int HB = 4096;     // bins of a 12-bit level

__global__ __launch_bounds__(KT) void topk_keys_kernel(const float* __restrict__ logits, int64_t img_stride, int n, int A, int ld, int per,
                                                       unsigned* __restrict__ keys, int* __restrict__ hist, int raw) {
    __shared__ int lh[HB];
    const int img = blockIdx.y;
    for (int i = threadIdx.x; i < HB; i += KT) lh[i] = 0;
    __syncthreads();
    const float* base = logits + (size_t)img * img_stride;
    const int j0 = blockIdx.x * per, j1 = min(j0 + per, n);
    for (int j = j0 + threadIdx.x; j < j1; j += KT) {
        const unsigned key = score_bits(base, j, A, ld, raw);
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
        if (scores) scores[(size_t)blockIdx.x * k + j] = __uint_as_float((unsigned)(v >> 32));
        idx[(size_t)blockIdx.x * k + j] = (int64_t)(~(unsigned)(v & 0xFFFFFFFFu));
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Round 4: phases 2-5 spread over the chip (the one-workgroup-per-image form below spent 130 of its 178 us in a 16384-word bitonic sort on ONE CU
// per image, and the step waits for ~60 % of this chain).
//   2. partition (chip-wide, one workgroup per key slice): the level-1 bin b1 that holds the k-th key is read off the histogram by every workgroup;
//      keys above the bin are survivors for certain, keys inside it are CANDIDATES; both are appended to per-image arrays as 64-bit
//      (key << 32 | ~index) words (positions reserved per workgroup: two global atomics per slice);
//   3. select (one workgroup per image): two more radix levels over the CANDIDATES only (a few thousand words, not the 143 640 keys) give the exact
//      k-th key; candidates >= it join the survivors (ties included); the tail of the last 1024-word chunk is zero-filled; cleans the histogram;
//   4. chunk sort (one 256-thread workgroup per 1024 survivors): bitonic sort in LDS, descending;
//   5. merge by rank (one workgroup per chunk): all sorted chunks of the image in LDS; an element's final position = its position in its own chunk
//      + the number of larger elements in every other chunk (binary searches); words are unique (they carry the index), so positions are too;
//      the first k positions are written out as fp32 scores + int64 indices.
// Outputs are identical to the one-workgroup form's (descending score, equal scores by ascending index).
struct TopkCnt { int surv, cand, filled, pad; };
constexpr int CH = 1024;      // words per sorted chunk
constexpr int MAXCH = CAP / CH;

__global__ __launch_bounds__(KT) void topk_partition_kernel(const unsigned* __restrict__ keys_all, const int* __restrict__ hist_all, int n, int k, int per,
                                                            unsigned long long* __restrict__ surv_all, unsigned long long* __restrict__ cand_all,
                                                            TopkCnt* __restrict__ cnt_all) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* st_s = reinterpret_cast<unsigned long long*>(smem);   // [per] staged survivors of this slice
    unsigned long long* st_c = st_s + per;                                     // [per] staged candidates
    __shared__ int h[HB];
    __shared__ int s_bin, s_krem, n_s, n_c, base_s, base_c;
    const int img = blockIdx.y;
    const unsigned* keys = keys_all + (size_t)img * n;
    const int* hist = hist_all + img * HB;
    for (int i = threadIdx.x; i < HB; i += KT) h[i] = hist[i];
    if (threadIdx.x == 0) { s_krem = k; s_bin = 0; n_s = 0; n_c = 0; }
    __syncthreads();
    int b1 = -1;                       // k == n: every key survives
    if (k < n) {
        find_bin_from_top(h, HB, &s_bin, &s_krem);
        __syncthreads();
        b1 = s_bin;
    }
    const int j0 = blockIdx.x * per, j1 = min(j0 + per, n);
    for (int j = j0 + threadIdx.x; j < j1; j += KT) {
        const unsigned key = keys[j];
        const int bin = (int)(key >> 20);
        const unsigned long long w = ((unsigned long long)key << 32) | (unsigned)(~(unsigned)j);
        if (bin > b1) st_s[atomicAdd(&n_s, 1)] = w;
        else if (bin == b1) st_c[atomicAdd(&n_c, 1)] = w;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        base_s = n_s ? atomicAdd(&cnt_all[img].surv, n_s) : 0;
        base_c = n_c ? atomicAdd(&cnt_all[img].cand, n_c) : 0;
    }
    __syncthreads();
    unsigned long long* surv = surv_all + (size_t)img * CAP;
    unsigned long long* cand = cand_all + (size_t)img * n;
    for (int i = threadIdx.x; i < n_s; i += KT)
        if (base_s + i < CAP) surv[base_s + i] = st_s[i];
    for (int i = threadIdx.x; i < n_c; i += KT) cand[base_c + i] = st_c[i];
}

__global__ __launch_bounds__(TT) void topk_select_kernel(int* __restrict__ hist_all, int n, int k, unsigned long long* __restrict__ surv_all,
                                                         const unsigned long long* __restrict__ cand_all, TopkCnt* __restrict__ cnt_all) {
    __shared__ int h[HB];
    __shared__ int s_bin, s_krem, s_fill;
    const int img = blockIdx.x;
    int* hist = hist_all + img * HB;
    for (int i = threadIdx.x; i < HB; i += TT) { h[i] = hist[i]; hist[i] = 0; }   // the histogram is clean again for the next call
    if (threadIdx.x == 0) { s_krem = k; s_fill = 0; s_bin = 0; }
    __syncthreads();
    const int nsurv = min(cnt_all[img].surv, CAP), ncand = cnt_all[img].cand;
    unsigned long long* surv = surv_all + (size_t)img * CAP;
    const unsigned long long* cand = cand_all + (size_t)img * n;
    if (k < n) {
        find_bin_from_top(h, HB, &s_bin, &s_krem);       // b1 again (the partition's choice), s_krem = how many of the candidates are wanted
        __syncthreads();
        const unsigned b1 = (unsigned)s_bin;
        for (int i = threadIdx.x; i < HB; i += TT) h[i] = 0;
        __syncthreads();
        for (int j = threadIdx.x; j < ncand; j += TT) atomicAdd(&h[(unsigned)(cand[j] >> 40) & (HB - 1)], 1);   // key bits 19..8
        __syncthreads();
        find_bin_from_top(h, HB, &s_bin, &s_krem);
        __syncthreads();
        const unsigned p24 = (b1 << 12) | (unsigned)s_bin;
        for (int i = threadIdx.x; i < 256; i += TT) h[i] = 0;
        __syncthreads();
        for (int j = threadIdx.x; j < ncand; j += TT) {
            const unsigned key = (unsigned)(cand[j] >> 32);
            if ((key >> 8) == p24) atomicAdd(&h[key & 255], 1);
        }
        __syncthreads();
        find_bin_from_top(h, 256, &s_bin, &s_krem);
        __syncthreads();
        const unsigned thr = (p24 << 8) | (unsigned)s_bin;
        for (int j = threadIdx.x; j < ncand; j += TT) {   // ties included; dropped only when the sort capacity overflows
            const unsigned long long w = cand[j];
            if ((unsigned)(w >> 32) >= thr) {
                const int pos = nsurv + atomicAdd(&s_fill, 1);
                if (pos < CAP) surv[pos] = w;
            }
        }
        __syncthreads();
    }
    __syncthreads();   // (k == n: nothing above separates the reads of the counters from their reset below)
    const int filled = min(nsurv + s_fill, CAP);
    for (int j = filled + threadIdx.x; j < (filled + CH - 1) / CH * CH; j += TT) surv[j] = 0ull;   // padding sorts last
    if (threadIdx.x == 0) { cnt_all[img].filled = filled; cnt_all[img].surv = 0; cnt_all[img].cand = 0; }
}

__global__ __launch_bounds__(256) void topk_chunk_sort_kernel(unsigned long long* __restrict__ surv_all, const TopkCnt* __restrict__ cnt_all) {
    __shared__ unsigned long long buf[CH];
    const int img = blockIdx.y, c = blockIdx.x;
    if (c * CH >= cnt_all[img].filled) return;
    unsigned long long* a = surv_all + (size_t)img * CAP + (size_t)c * CH;
    for (int i = threadIdx.x; i < CH; i += 256) buf[i] = a[i];
    __syncthreads();
    for (int size = 2; size <= CH; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
#pragma unroll
            for (int t = threadIdx.x; t < CH / 2; t += 256) {
                const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const unsigned long long x = buf[lo], y = buf[hi];
                if ((x < y) == desc) { buf[lo] = y; buf[hi] = x; }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < CH; i += 256) a[i] = buf[i];
}

__global__ __launch_bounds__(TT) void topk_merge_emit_kernel(const unsigned long long* __restrict__ surv_all, const TopkCnt* __restrict__ cnt_all, int k,
                                                             float* __restrict__ scores, int64_t* __restrict__ idx) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* buf = reinterpret_cast<unsigned long long*>(smem);   // [nch][CH]: every sorted chunk of the image
    const int img = blockIdx.y, c = blockIdx.x;
    const int filled = cnt_all[img].filled;
    if (c * CH >= filled) {
        // (filled < k cannot happen unless the capacity overflowed: then the tail reads as zeros, like the one-workgroup form)
        for (int j = max(filled, c * CH) + threadIdx.x; j < min(k, (c + 1) * CH); j += TT) {
            if (scores) scores[(size_t)img * k + j] = 0.f;
            idx[(size_t)img * k + j] = (int64_t)(~0u);
        }
        return;
    }
    const int nch = (filled + CH - 1) / CH;
    const uint4* src = reinterpret_cast<const uint4*>(surv_all + (size_t)img * CAP);
    uint4* dst = reinterpret_cast<uint4*>(buf);
    for (int i = threadIdx.x; i < nch * (CH / 2); i += TT) dst[i] = src[i];
    __syncthreads();
    const int t = threadIdx.x;
    const unsigned long long x = buf[c * CH + t];
    if (x == 0ull) return;                     // padding
    int rank = t;
    for (int o = 0; o < nch; o++) {
        if (o == c) continue;
        const unsigned long long* a = buf + o * CH;   // descending: count the elements larger than x
        int lo = 0, hi = CH;
#pragma unroll
        for (int it = 0; it < 11; it++) {          // 1025 possible answers (0 .. CH): 11 halvings; once lo == hi the steps are no-ops
            if (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (a[mid] > x) lo = mid + 1; else hi = mid;
            }
        }
        rank += lo;
    }
    if (rank < k) {
        if (scores) scores[(size_t)img * k + rank] = __uint_as_float((unsigned)(x >> 32));
        idx[(size_t)img * k + rank] = (int64_t)(~(unsigned)(x & 0xFFFFFFFFu));
    }
}

}  // namespace

static int topk_run(const float* logits, int64_t img_stride, int N, int n, int A, int ld, int k, float* scores, int64_t* idx, int raw, void* stream);

extern "C" int abr_topk_sigmoid(const float* logits, int64_t img_stride, int N, int n, int A, int ld, int k, float* scores,
                                int64_t* idx, void* stream) {
    ABR_REQUIRE(N >= 0 && n > 0 && A >= 1 && ld >= A && k >= 0 && k <= n, "topk_sigmoid: bad args");
    ABR_REQUIRE(k <= CAP - 1024, "topk_sigmoid: k too large for the in-LDS sort (max 15360)");
    if (N == 0 || k == 0) return ABR_OK;
    ABR_REQUIRE(logits && scores && idx, "topk_sigmoid: null pointer");
    return topk_run(logits, img_stride, N, n, A, ld, k, scores, idx, 0, stream);
}

extern "C" int64_t abr_sort_scores_max_n(void) { return CAP - 1024; }

extern "C" int abr_sort_scores_desc(const float* scores, int n, int64_t* order, void* stream) {
    ABR_REQUIRE(n >= 0 && n <= CAP - 1024, "sort_scores_desc: at most 15360 scores");
    if (n == 0) return ABR_OK;
    ABR_REQUIRE(scores && order, "sort_scores_desc: null pointer");
    return topk_run(scores, n, 1, n, 1, 1, n, nullptr, order, 1, stream);
}

static int topk_run(const float* logits, int64_t img_stride, int N, int n, int A, int ld, int k, float* scores, int64_t* idx, int raw, void* stream) {
    static const bool one_wg = getenv("ABR_TOPK_ONE_WG") && atoi(getenv("ABR_TOPK_ONE_WG")) != 0;   // round 3's one-workgroup-per-image phase 2 (A/B)
    const int G = 64;
    const int per = ((n + G - 1) / G + KT - 1) / KT * KT;
    const size_t lds_sort = (size_t)CAP * 8, lds_part = (size_t)per * 16;
    // the partition kernel keeps its slice's keys in LDS beside 16 KB of histogram: beyond 64 KB (n > ~196 k keys per image; the C4 geometries of
    // this path stay below 63 k) it cannot be launched -- say so instead of returning a launch error
    ABR_REQUIRE(one_wg || lds_part + 16384 <= 65536, "topk_sigmoid: %d keys per image exceed the partition kernel's LDS (at most 196608; ABR_TOPK_ONE_WG=1 has no such limit)", n);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(topk_select_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sort);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(topk_merge_emit_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sort);
        attr = true;
    }
    // scratch per stream (launches on a stream are ordered), all of it at offsets that depend on CAPACITIES only:
    //   [level-1 histograms: cap_imgs x 4096][counters: cap_imgs][survivors: cap_imgs x CAP words][keys: cap_keys][candidates: cap_keys words]
    // The histogram and counter regions are zero on entry (allocation zero-fills them, the select phase cleans what it used); their position does not
    // depend on n or N, so a later call with a smaller feature map / another batch size never finds stale key bits where it expects zeros.
    // Grow-only: keyed on capacity, not on N ==.
    struct Ws { void* p = nullptr; int cap_imgs = 0; size_t cap_keys = 0; };
    static std::mutex mu;
    static std::map<hipStream_t, Ws> pool;
    hipStream_t st = abr::as_stream(stream);
    unsigned* keys;
    int* hist;
    TopkCnt* cnt;
    unsigned long long *surv, *cand;
    {
        std::lock_guard<std::mutex> g(mu);
        Ws& w = pool[st];
        const size_t need_keys = (size_t)N * n;
        if (w.cap_imgs < N || w.cap_keys < need_keys) {
            const int imgs = std::max(std::max(N, w.cap_imgs), 8);
            const size_t nkeys = std::max(need_keys, w.cap_keys);
            if (w.p) { (void)hipStreamSynchronize(st); (void)hipFree(w.p); w.p = nullptr; w.cap_imgs = 0; w.cap_keys = 0; }
            const size_t head = (size_t)imgs * HB * 4 + (size_t)imgs * sizeof(TopkCnt);
            ABR_REQUIRE(hipMalloc(&w.p, head + (size_t)imgs * CAP * 8 + nkeys * 4 + nkeys * 8 + 16) == hipSuccess, "topk_sigmoid: scratch allocation failed");
            w.cap_imgs = imgs;
            w.cap_keys = nkeys;
            (void)hipMemsetAsync(w.p, 0, head, st);
        }
        char* b = static_cast<char*>(w.p);
        hist = reinterpret_cast<int*>(b);                          b += (size_t)w.cap_imgs * HB * 4;
        cnt = reinterpret_cast<TopkCnt*>(b);                       b += (size_t)w.cap_imgs * sizeof(TopkCnt);
        surv = reinterpret_cast<unsigned long long*>(b);           b += (size_t)w.cap_imgs * CAP * 8;
        cand = reinterpret_cast<unsigned long long*>(b);           b += w.cap_keys * 8;
        keys = reinterpret_cast<unsigned*>(b);
    }
    const unsigned slices = (unsigned)((n + per - 1) / per);
    topk_keys_kernel<<<dim3(slices, (unsigned)N), KT, 0, st>>>(logits, img_stride, n, A, ld, per, keys, hist, raw);
    if (one_wg) {
        topk_select_sort_kernel<<<N, TT, lds_sort, st>>>(keys, hist, n, k, scores, idx);
    } else {
        topk_partition_kernel<<<dim3(slices, (unsigned)N), KT, lds_part, st>>>(keys, hist, n, k, per, surv, cand, cnt);
        topk_select_kernel<<<N, TT, 0, st>>>(hist, n, k, surv, cand, cnt);
        topk_chunk_sort_kernel<<<dim3(MAXCH, (unsigned)N), 256, 0, st>>>(surv, cnt);
        topk_merge_emit_kernel<<<dim3(MAXCH, (unsigned)N), TT, lds_sort, st>>>(surv, cnt, k, scores, idx);
    }
    ABR_CHECK_LAUNCH("topk_sigmoid");
    return ABR_OK;
}
