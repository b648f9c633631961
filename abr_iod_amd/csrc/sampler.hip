// BalancedPositiveNegativeSampler as ONE kernel per image (no nonzero()/randperm()/host syncs).
//
// Reference: maskrcnn_benchmark/modeling/balanced_positive_negative_sampler.py:19-77: per image
//   positive = nonzero(labels >= 1), negative = nonzero(labels == 0)
//   num_pos = min(#positive, int(batch*fraction)), num_neg = min(#negative, batch - num_pos)
//   choose randperm(#positive)[:num_pos] and randperm(#negative)[:num_neg]            (uniform random subsets)
// and the callers immediately turn the masks back into ASCENDING index lists with nonzero() (rpn/loss.py:119-123,
// box_head/loss.py:114).  In the reference that is ~12 tiny ATen kernels and 2 blocking nonzero() per image.
//
// Here a 1024-thread workgroup per image draws the same distribution: every candidate gets a 32-bit random key
// (counter-based hash of seed, image, index); the num-smallest keys are the subset (= a random permutation's prefix).
// The k-th smallest key is found by a 4-pass radix select on LDS histograms, ties are broken by index, and the chosen
// indices are written in ascending order with ballot-based block scans.  The draw itself is device RNG and, exactly as
// in the reference, not reproducible across devices: parity tests inject the sampled indices.
#include "common.h"

namespace {

constexpr int NT_BIG = 1024;   // RPN anchors (143 640 labels per image)
constexpr int NT_SMALL = 256;  // box-head candidates (~2000 per image): this launch sits on the step's critical path between NMS and ROIAlign, and
                               // its ~40 workgroup barriers cost a quarter as much with four waves as with sixteen (66 -> ~25 us)

__device__ __forceinline__ unsigned rkey(uint64_t seed, unsigned img, unsigned idx) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * ((uint64_t)img << 32 | idx) + 0x632BE59BD9B4E019ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (unsigned)(z >> 16);
}

template <typename T>
__device__ __forceinline__ int cls_of(T v) { return v >= (T)1 ? 1 : (v == (T)0 ? 0 : -1); }  // 1 pos, 0 neg, -1 ignored

// exclusive prefix of a 1-bit flag over the block + block total.  sm: NWV ints.  All threads must call.
template <int NWV>
__device__ __forceinline__ int block_scan_flag(bool f, int* sm, int* total) {
    const unsigned long long b = __ballot(f);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int in_wave = __builtin_popcountll(b & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) sm[w] = __builtin_popcountll(b);
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NWV; i++) {
        const int c = sm[i];
        if (i < w) base += c;
        tot += c;
    }
    *total = tot;
    return base + in_wave;
}

template <typename T, int NT>
__global__ __launch_bounds__(NT) void sample_kernel(const T* __restrict__ labels_all, int n, int64_t stride, int batch, int max_pos,
                                                     uint64_t seed, int img0, int64_t idx_off, int64_t* __restrict__ pos_idx,
                                                     int64_t* __restrict__ neg_idx, int32_t* __restrict__ counts) {
    constexpr int NWV = NT / 64;
    __shared__ int hist[256];
    __shared__ int sm[NWV];
    __shared__ unsigned s_prefix;
    __shared__ int s_krem;
    const int img = blockIdx.x;
    const T* labels = labels_all + (size_t)img * stride;
    const unsigned uimg = (unsigned)(img0 + img);
    int64_t* outs[2] = {neg_idx + (size_t)img * batch, pos_idx + (size_t)img * max_pos};
    const int64_t off = idx_off * img;

    // 1. counts
    int cp = 0, cn = 0;
    for (int i = threadIdx.x; i < n; i += NT) {
        const int c = cls_of(labels[i]);
        cp += c == 1;
        cn += c == 0;
    }
    int tot;
    // reduce via the scan helper's LDS (two rounds)
    __shared__ int s_cnt[2];
    if (threadIdx.x == 0) { s_cnt[0] = 0; s_cnt[1] = 0; }
    __syncthreads();
    atomicAdd(&s_cnt[1], cp);
    atomicAdd(&s_cnt[0], cn);
    __syncthreads();
    const int n_pos = s_cnt[1], n_neg = s_cnt[0];
    const int k_pos = min(n_pos, max_pos);
    const int k_neg = min(n_neg, batch - k_pos);
    if (threadIdx.x == 0) { counts[2 * img] = k_pos; counts[2 * img + 1] = k_neg; }

    for (int cls = 1; cls >= 0; cls--) {
        const int k = cls ? k_pos : k_neg, avail = cls ? n_pos : n_neg;
        int64_t* out = outs[cls];
        const int cap = cls ? max_pos : batch;
        // 2. threshold key thr = k-th smallest key of this class (radix select, 8 bits per pass, MSB first)
        unsigned thr = 0xFFFFFFFFu;
        int ties_needed = 0;  // how many elements with key == thr are taken (lowest indices first)
        const bool all = k >= avail;
        if (!all && k > 0) {
            if (threadIdx.x == 0) { s_prefix = 0; s_krem = k; }
            __syncthreads();
            for (int pass = 3; pass >= 0; pass--) {
                for (int i = threadIdx.x; i < 256; i += NT) hist[i] = 0;
                __syncthreads();
                const unsigned prefix = s_prefix;
                const int shift = pass * 8;
                const unsigned hi_mask = pass == 3 ? 0u : (0xFFFFFFFFu << (shift + 8));
                for (int i = threadIdx.x; i < n; i += NT) {
                    if (cls_of(labels[i]) != cls) continue;
                    const unsigned key = rkey(seed, uimg, (unsigned)i);
                    if ((key & hi_mask) == (prefix & hi_mask)) atomicAdd(&hist[(key >> shift) & 255], 1);
                }
                __syncthreads();
                if (threadIdx.x < 64) {   // wave 0: the bin holding the krem-th smallest key (lane l owns bins 4 l .. 4 l + 3; prefix sum over the lanes)
                    const int lane = threadIdx.x, krem = s_krem;
                    const int h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
                    const int sum4 = h0 + h1 + h2 + h3;
                    int inc = sum4;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) {
                        const int v = __shfl_up(inc, d, 64);
                        if (lane >= d) inc += v;
                    }
                    const int exc = inc - sum4;
                    if (exc < krem && krem <= inc) {   // exactly one lane (krem >= 1 and the class holds >= krem keys with this prefix)
                        int b = 4 * lane, cum = exc;
                        if (cum + h0 < krem) { cum += h0; b++; if (cum + h1 < krem) { cum += h1; b++; if (cum + h2 < krem) { cum += h2; b++; } } }
                        s_prefix = prefix | ((unsigned)b << shift);
                        s_krem = krem - cum;  // rank of the target inside bin b
                    }
                }
                __syncthreads();
            }
            thr = s_prefix;
            ties_needed = s_krem;  // elements equal to thr to take
        }
        // 3. ordered compaction: (key < thr) or (key == thr and tie rank < ties_needed); `all` takes every member of the class
        int written = 0, ties_seen = 0;
        for (int base = 0; base < n; base += NT) {
            const int i = base + threadIdx.x;
            bool member = false, less = false, tie = false;
            if (i < n && k > 0) {
                member = cls_of(labels[i]) == cls;
                if (member && !all) {
                    const unsigned key = rkey(seed, uimg, (unsigned)i);
                    less = key < thr;
                    tie = key == thr;
                }
            }
            bool take = member && (all || less);
            if (!all) {
                int tt;
                const int trank = block_scan_flag<NWV>(member && tie, sm, &tt);
                if (member && tie && ties_seen + trank < ties_needed) take = true;
                ties_seen += tt;
            }
            int nt;
            const int pos = block_scan_flag<NWV>(take, sm, &nt);
            if (take && written + pos < cap) out[written + pos] = (int64_t)i + off;
            written += nt;
        }
        for (int j = written + threadIdx.x; j < cap; j += NT) out[j] = -1;  // padding
        __syncthreads();
    }
    (void)tot;
}

}  // namespace

extern "C" int abr_sample_pos_neg(const void* labels, int labels_are_int64, int N, int n, int64_t stride, int batch_size, int max_pos,
                                  uint64_t seed, int first_image, int64_t index_offset_per_image, int64_t* pos_idx,
                                  int64_t* neg_idx, int32_t* counts, void* stream) {
    ABR_REQUIRE(N >= 0 && n >= 0 && batch_size > 0 && max_pos >= 0 && max_pos <= batch_size, "sample_pos_neg: bad args");
    if (N == 0) return ABR_OK;
    ABR_REQUIRE(labels && pos_idx && neg_idx && counts, "sample_pos_neg: null pointer");
    hipStream_t st = abr::as_stream(stream);
    const bool small = n <= 8192;
#define ABR_SAMPLE_LAUNCH(T, NTHR) sample_kernel<T, NTHR><<<N, NTHR, 0, st>>>((const T*)labels, n, stride, batch_size, max_pos, seed, first_image, \
                                                                              index_offset_per_image, pos_idx, neg_idx, counts)
    if (labels_are_int64) { if (small) ABR_SAMPLE_LAUNCH(int64_t, NT_SMALL); else ABR_SAMPLE_LAUNCH(int64_t, NT_BIG); }
    else { if (small) ABR_SAMPLE_LAUNCH(float, NT_SMALL); else ABR_SAMPLE_LAUNCH(float, NT_BIG); }
#undef ABR_SAMPLE_LAUNCH
    ABR_CHECK_LAUNCH("sample_pos_neg");
    return ABR_OK;
}
