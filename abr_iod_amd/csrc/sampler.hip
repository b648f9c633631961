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
// indices are written in ascending order (every wave owns a contiguous index range; ballot prefixes inside it).  The draw itself is device RNG and, exactly as
// in the reference, not reproducible across devices: parity tests inject the sampled indices.
#include "common.h"

namespace {

constexpr int NT_BIG = 1024;   // RPN anchors (35 910 labels per image at 600x1000, 63 000 at 800x1333)
constexpr int NT_SMALL = 256;  // box-head candidates (~2000 per image): this launch sits on the step's chain between NMS and ROIAlign; four waves
                               // keep its ~20 workgroup barriers (radix select) cheap

__device__ __forceinline__ unsigned rkey(uint64_t seed, unsigned img, unsigned idx) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * ((uint64_t)img << 32 | idx) + 0x632BE59BD9B4E019ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (unsigned)(z >> 16);
}

template <typename T>
__device__ __forceinline__ int cls_of(T v) { return v >= (T)1 ? 1 : (v == (T)0 ? 0 : -1); }  // 1 pos, 0 neg, -1 ignored

template <typename T, int NT>
__global__ __launch_bounds__(NT) void sample_kernel(const T* __restrict__ labels_all, int n, int64_t stride, int batch, int max_pos,
                                                     uint64_t seed, int img0, int64_t idx_off, int64_t* __restrict__ pos_idx,
                                                     int64_t* __restrict__ neg_idx, int32_t* __restrict__ counts, int has_top) {
    constexpr int NWV = NT / 64;
    extern __shared__ unsigned char s_top[];   // [n] when has_top: the top byte of every candidate's key (see step 1)
    __shared__ int hist[256];
    __shared__ unsigned s_prefix;
    __shared__ int s_krem;
    const int img = blockIdx.x;
    const T* labels = labels_all + (size_t)img * stride;
    const unsigned uimg = (unsigned)(img0 + img);
    int64_t* outs[2] = {neg_idx + (size_t)img * batch, pos_idx + (size_t)img * max_pos};
    const int64_t off = idx_off * img;

    // Wave w owns the contiguous index range [i0, i1); lane l visits i0 + l + 64 it.  The labels are read from memory ONCE: the class of every
    // visited index (2 bits: 0 negative, 1 positive, 2 ignored) is kept in two 64-bit registers per lane (up to 64 visits = 65 536 candidates
    // with 16 waves) and every later pass -- the radix select's four, the compaction's two -- decodes it from there.  (Re-reading labels[i] in
    // each pass cost one exposed L2 round trip per 64 x NWV candidates and pass.)  The KEY of a candidate is a 64-bit mix (three 64-bit multiplies: ~200
    // cycles per wave instruction); six passes that each hashed all 35 910 anchors were ~85 of the RPN call's 120 us.  The keys are hashed once, here,
    // and only their TOP BYTE is kept (LDS, one byte per candidate): the first radix pass histograms those bytes, the later passes and the compaction
    // hash again only the 1 / 256 of the candidates whose top byte equals the threshold's.
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int chunk = ((n + NWV - 1) / NWV + 63) & ~63;
    const int i0 = wv * chunk, i1 = min(n, i0 + chunk);
    const int nit = chunk / 64;
    const bool cached = nit <= 64;                 // (workgroup-uniform)
    unsigned long long code0 = 0ull, code1 = 0ull;
    // 1. counts (+ the cache)
    int cp = 0, cn = 0;
    for (int it0 = 0; it0 < nit; it0 += 4) {       // four independent loads in flight per lane
        int c[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + lane + 64 * (it0 + u);
            c[u] = (it0 + u < nit && i < i1) ? cls_of(labels[i]) : -1;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int it = it0 + u;
            cp += c[u] == 1;
            cn += c[u] == 0;
            if (has_top && c[u] >= 0) s_top[i0 + lane + 64 * it] = (unsigned char)(rkey(seed, uimg, (unsigned)(i0 + lane + 64 * it)) >> 24);
            const unsigned long long two = c[u] < 0 ? 2ull : (unsigned long long)c[u];
            if (it < 32) code0 |= two << (2 * it);
            else if (it < 64) code1 |= two << (2 * (it - 32));
        }
    }
    // class of visit `it` of this lane (-1 ignored / out of range)
    auto cls_at = [&](int it) -> int {
        if (cached) {
            const unsigned two = (unsigned)(((it < 32 ? code0 : code1) >> (2 * (it & 31))) & 3ull);
            return two == 2u ? -1 : (int)two;
        }
        const int i = i0 + lane + 64 * it;
        return i < i1 ? cls_of(labels[i]) : -1;
    };
    // membership in class `cls` and the key's top byte for visits it0 .. it0 + 3 of this lane
    auto fetch4 = [&](int it0, int cls, bool (&mem)[4], unsigned (&top)[4]) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int it = it0 + u;
            mem[u] = it < nit && cls_at(it) == cls;
            top[u] = (has_top && mem[u]) ? (unsigned)s_top[i0 + lane + 64 * it] : 0u;
        }
    };
    __shared__ int s_cls[2][NWV];          // members of class 0 / 1 in every wave's range (step 1): pass A of a class that is taken whole
    __shared__ int s_wl[NWV], s_wt[NWV];   // per-wave counts (class totals here, takes / ties of a wave's chunk in step 3)
    {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { cp += __shfl_xor(cp, o, 64); cn += __shfl_xor(cn, o, 64); }
        if (lane == 0) { s_wl[wv] = cp; s_wt[wv] = cn; s_cls[1][wv] = cp; s_cls[0][wv] = cn; }
        __syncthreads();
    }
    int n_pos = 0, n_neg = 0;
#pragma unroll
    for (int i = 0; i < NWV; i++) { n_pos += s_wl[i]; n_neg += s_wt[i]; }
    __syncthreads();
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
                for (int it0 = 0; it0 < nit; it0 += 4) {       // (four visits per trip: their LDS reads are in flight together)
                    bool mem[4];
                    unsigned top[4];
                    fetch4(it0, cls, mem, top);
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        if (!mem[u]) continue;
                        if (has_top) {
                            if (pass == 3) { atomicAdd(&hist[top[u]], 1); continue; }
                            if (top[u] != (prefix >> 24)) continue;
                        }
                        const unsigned key = rkey(seed, uimg, (unsigned)(i0 + lane + 64 * (it0 + u)));
                        if ((key & hi_mask) == (prefix & hi_mask)) atomicAdd(&hist[(key >> shift) & 255], 1);
                    }
                }
                __syncthreads();
                if (threadIdx.x < 64) {   // wave 0: the bin holding the krem-th smallest key (lane l owns bins 4 l .. 4 l + 3; prefix sum over the lanes)
                    const int krem = s_krem;
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
        // 3. ordered compaction: (key < thr) or (key == thr and tie rank < ties_needed); `all` takes every member of the class.  Wave w owns the
        //    contiguous index range [w * chunk, (w + 1) * chunk): it counts its takes and ties (pass A), one barrier turns the counts into every
        //    wave's output offset and tie rank, and it writes its range in index order with ballot prefixes (pass B) -- two workgroup barriers per
        //    class instead of four per 64 x NWV candidates (the box head's call sits on the step's chain between NMS and ROIAlign)
        const unsigned long long lt = (1ull << lane) - 1ull;
        auto flags = [&](int it, bool member, unsigned top, bool& less, bool& tie) {
            less = tie = false;
            if (member && !all) {
                const unsigned t8 = has_top ? top : (thr >> 24);
                if (t8 < (thr >> 24)) less = true;
                else if (t8 == (thr >> 24)) {
                    const unsigned key = rkey(seed, uimg, (unsigned)(i0 + lane + 64 * it));
                    less = key < thr;
                    tie = key == thr;
                }
            }
        };
        int c_less = 0, c_tie = 0;
        if (all) c_less = k > 0 ? s_cls[cls][wv] : 0;       // the whole class: step 1 counted it per wave already
        for (int it0 = 0; it0 < nit && k > 0 && !all; it0 += 4) {
            bool mem[4];
            unsigned top[4];
            fetch4(it0, cls, mem, top);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                bool less, tie;
                flags(it0 + u, mem[u], top[u], less, tie);
                c_less += __builtin_popcountll(__ballot(mem[u] && less));
                c_tie += __builtin_popcountll(__ballot(mem[u] && tie));
            }
        }
        if (lane == 0) { s_wl[wv] = c_less; s_wt[wv] = c_tie; }
        __syncthreads();
        int ties_before = 0, base = 0, written = 0;
        {
            int tb = 0, acc = 0;
#pragma unroll
            for (int v = 0; v < NWV; v++) {
                const int take_t = all ? 0 : min(max(ties_needed - tb, 0), s_wt[v]);
                if (v == wv) { ties_before = tb; base = acc; }
                acc += s_wl[v] + take_t;
                tb += s_wt[v];
            }
            written = acc;      // everything the class contributes (before the cap)
        }
        int ties_seen = ties_before;
        for (int it0 = 0; it0 < nit && k > 0; it0 += 4) {
            bool mem[4];
            unsigned top[4];
            fetch4(it0, cls, mem, top);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                bool less, tie;
                flags(it0 + u, mem[u], top[u], less, tie);
                const int i = i0 + lane + 64 * (it0 + u);
                const unsigned long long bt = __ballot(mem[u] && tie);
                const int trank = ties_seen + __builtin_popcountll(bt & lt);
                const bool take = mem[u] && (all || less || (tie && trank < ties_needed));
                const unsigned long long bk = __ballot(take);
                const int pos = base + __builtin_popcountll(bk & lt);
                if (take && pos < cap) out[pos] = (int64_t)i + off;
                base += __builtin_popcountll(bk);
                ties_seen += __builtin_popcountll(bt);
            }
        }
        written = min(written, cap);
        __syncthreads();        // (s_wl / s_wt are rewritten by the next class)
        for (int j = written + threadIdx.x; j < cap; j += NT) out[j] = -1;  // padding
        __syncthreads();
    }
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
    // one LDS byte per candidate (160 KB per compute unit).  The large request has to be granted per device: where it is not (another part, a
    // failed call) candidate sets beyond the default 64 KB run without the key-byte cache (has_top = 0: the kernel re-hashes instead)
    static int big_ok[64] = {0};   // per device: 0 = not asked yet, 1 = granted, -1 = refused
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev = dev < 0 || dev >= 64 ? 0 : dev;
    if (big_ok[dev] == 0) {
        const bool a = hipFuncSetAttribute(reinterpret_cast<const void*>(sample_kernel<float, NT_BIG>), hipFuncAttributeMaxDynamicSharedMemorySize, 150016) == hipSuccess;
        const bool b = hipFuncSetAttribute(reinterpret_cast<const void*>(sample_kernel<int64_t, NT_BIG>), hipFuncAttributeMaxDynamicSharedMemorySize, 150016) == hipSuccess;
        big_ok[dev] = (a && b) ? 1 : -1;
        (void)hipGetLastError();
    }
    const int has_top = n <= (big_ok[dev] == 1 ? 150000 : 60000);
    const size_t lds = has_top ? (size_t)((n + 15) & ~15) : 0;
#define ABR_SAMPLE_LAUNCH(T, NTHR) sample_kernel<T, NTHR><<<N, NTHR, lds, st>>>((const T*)labels, n, stride, batch_size, max_pos, seed, first_image, \
                                                                                index_offset_per_image, pos_idx, neg_idx, counts, has_top)
    if (labels_are_int64) { if (small) ABR_SAMPLE_LAUNCH(int64_t, NT_SMALL); else ABR_SAMPLE_LAUNCH(int64_t, NT_BIG); }
    else { if (small) ABR_SAMPLE_LAUNCH(float, NT_SMALL); else ABR_SAMPLE_LAUNCH(float, NT_BIG); }
#undef ABR_SAMPLE_LAUNCH
    ABR_CHECK_LAUNCH("sample_pos_neg");
    return ABR_OK;
}
