// Greedy NMS entirely on device, batched over images.
//
// Reference semantics: csrc/cpu/nms_cpu.cpp:5-75 (`ovr >= thr` suppresses -- canonical here) and
// csrc/cuda/nms.cu:23-131 (`>`; 64x64 bit-mask tiles, then an 18 MB blocking D2H copy and a serial
// HOST sweep, nms.cu:99-123).  IoU arithmetic is kept in the reference's form inter/(a+b-inter) with
// contraction off so the compare against thr is bit-identical (index-exact keep lists).
//
// MI355X design:
//   pass 1  suppression bit-matrix: one wave64 per (row tile, column tile); a lane owns one row box and
//           builds one 64-bit word against the 64 column boxes staged in LDS -> a wave writes one
//           coalesced 512 B row of words.  Only tiles on/above the diagonal are computed.
//   pass 2  the greedy sweep stays ON DEVICE: one 256-thread workgroup per image.  Per 64-box block the word of
//           already-suppressed boxes is the OR of mask[i][block] over the boxes i kept so far (one independent
//           8 B load per kept box, spread over the workgroup), then wave 0 resolves the diagonal tile with lane
//           broadcasts (no memory in the serial chain).  Stops once max_keep boxes are kept (post_nms_top_n),
//           which on RPN proposals is long before the end of the list.
//   chunks  The two passes alternate over chunks of kChunk = 2048 boxes (mask chunk p, sweep chunk p, mask chunk p+1, ...): the mask
//           words of chunk p's columns are only ever read for the boxes KEPT before the chunk and for the chunk's own rows, so
//           that is all pass 1 computes -- (kept so far + 2048) x 2048 pairs instead of the whole upper triangle -- and every
//           launch after the sweep has its max_keep boxes returns at once.  2000 of 12000 proposals kept by box ~5500: 11 M pairs
//           per image instead of 72 M.
#include "common.h"

namespace {

#pragma clang fp contract(off)
__device__ __forceinline__ bool suppresses(const float4 a, float area_a, const float4 b, float thr, int strict_gt) {
    const float xx1 = fmaxf(a.x, b.x), yy1 = fmaxf(a.y, b.y);
    const float xx2 = fminf(a.z, b.z), yy2 = fminf(a.w, b.w);
    const float w = fmaxf(0.f, xx2 - xx1 + 1), h = fmaxf(0.f, yy2 - yy1 + 1);
    const float inter = w * h;
    const float area_b = (b.z - b.x + 1) * (b.w - b.y + 1);
    const float ovr = inter / (area_a + area_b - inter);
    return strict_gt ? (ovr > thr) : (ovr >= thr);
}

constexpr int kChunk = 2048;   // boxes per chunk (a multiple of the sweep's 256-box blocks)

// Mask words of the columns of chunk p.  grid = (kChunk/64 column tiles, ceil(max_keep/64) kept-row tiles + kChunk/64 chunk-row tiles, N).
// Row tiles below `kept_tiles` take their rows from the keep list (boxes kept before this chunk: every one of them precedes every
// column of the chunk, so the whole word is computed); the others are the chunk's own rows (on / above the diagonal only).
__global__ __launch_bounds__(64) void nms_mask_chunk_kernel(const float* __restrict__ boxes, const int32_t* __restrict__ counts,
                                                             int n_max, int words, float thr, int strict_gt, int chunk, int kept_tiles,
                                                             int max_keep, const int32_t* __restrict__ keep,
                                                             const int32_t* __restrict__ n_keep, uint64_t* __restrict__ mask) {
    const int img = blockIdx.z;
    const int n = counts[img];
    const int c0 = chunk * kChunk;
    const int cnt = n_keep[img];
    if (c0 >= n || cnt >= max_keep) return;          // the sweep is over for this image
    const int ct = c0 / 64 + blockIdx.x;             // global column tile
    if (ct * 64 >= n) return;
    const int lane = threadIdx.x;
    int ri, jstart = 0;
    if ((int)blockIdx.y < kept_tiles) {
        const int k = blockIdx.y * 64 + lane;
        if (blockIdx.y * 64 >= cnt) return;
        ri = k < cnt ? keep[(size_t)img * max_keep + k] : -1;
    } else {
        const int rt = c0 / 64 + (blockIdx.y - kept_tiles);
        if (ct < rt || rt * 64 >= n) return;
        ri = rt * 64 + lane;
        if (ri >= n) ri = -1;
        if (rt == ct) jstart = lane + 1;
    }
    __shared__ float4 cb[64];
    const float4* b = reinterpret_cast<const float4*>(boxes) + (size_t)img * n_max;
    const int cj = ct * 64 + lane;
    cb[lane] = cj < n ? b[cj] : make_float4(0, 0, -1, -1);
    __syncthreads();
    if (ri < 0) return;
    const float4 a = b[ri];
    const float area_a = (a.z - a.x + 1) * (a.w - a.y + 1);
    uint64_t bits = 0;
    const int jend = min(64, n - ct * 64);
    for (int j = jstart; j < jend; j++)
        if (suppresses(a, area_a, cb[j], thr, strict_gt)) bits |= 1ull << j;
    mask[((size_t)img * n_max + ri) * words + ct] = bits;
}

// grid = N, block = 256.  The kept list lives in LDS (max_keep ints).
// Column formulation: when the sweep reaches a block of boxes, the boxes of that block already suppressed are
//   removed(block) = OR over every box i kept so far of mask[i][block's words]
// -- one load per kept box, independent of each other, spread over the threads and OR-reduced.  (The earlier row formulation OR-ed
// each kept row into all later words: ~188x more loads.)
// The sweep is a chain of dependent steps; per block of SW = 4 mask words = 256 boxes (a kept box's four words are one 32-byte
// stretch of its mask row) the workgroup splits into two roles that run side by side:
//   * wave 0 runs the greedy chain of block sb on the scalar unit: count-trailing-zeros over the not-yet-suppressed bits of one
//     64-box word, one lane broadcast of the kept box's diagonal word per kept box (at most max_keep picks per image); what a word's
//     kept boxes suppress in the block's later words is OR-reduced across the wave once per word, outside the chain.  It touches no
//     global memory (its operands arrive through LDS), so no memory wait ever sits in the chain;
//   * waves 1-3 fetch everything block sb+1 will need: the gather over the boxes kept before sb, and -- unconditionally, selected by
//     the keep bits afterwards -- the next block's words of block sb's own rows and the diagonal words of block sb+1, into LDS.
constexpr int SW = 4;

__global__ __launch_bounds__(256) void nms_sweep_kernel(const uint64_t* __restrict__ mask, const int32_t* __restrict__ counts,
                                                         int n_max, int words, int max_keep, int chunk, int32_t* __restrict__ keep,
                                                         int32_t* __restrict__ n_keep) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* kept = reinterpret_cast<int*>(smem);                                  // [max_keep]
    uint64_t* s_red = reinterpret_cast<uint64_t*>(smem + (((size_t)max_keep * 4 + 15) & ~(size_t)15));  // [2][4 waves][SW] gathered words
    uint64_t* s_diag = s_red + 2 * 4 * SW;       // [2][SW q][SW w][64 lanes]: word w of the block for its row 64 q + lane (0 below the diagonal)
    uint64_t* s_nxt = s_diag + 2 * SW * SW * 64; // [2][SW q][SW w][64 lanes]: word w of the NEXT block for the same rows
    int* s_cnt = reinterpret_cast<int*>(s_nxt + 2 * SW * SW * 64);
    const int img = blockIdx.x;
    const int n = counts[img];
    const uint64_t* m = mask + (size_t)img * n_max * words;
    int32_t* kp = keep + (size_t)img * max_keep;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = (n + 63) / 64;                 // mask words that exist for this image (the mask kernel writes no others)
    // this launch sweeps the blocks of ONE chunk: [sb0, nsb); it picks up the keep list the earlier chunks left in `keep` / `n_keep`
    const int sb0 = chunk * (kChunk / (64 * SW));
    const int nsb = min((nw + SW - 1) / SW, sb0 + kChunk / (64 * SW));
    const int cnt0 = n_keep[img];
    if (sb0 >= nsb || cnt0 >= max_keep) return;
    for (int k = threadIdx.x; k < cnt0; k += 256) kept[k] = kp[k];
    if (threadIdx.x == 0) s_cnt[sb0 & 1] = cnt0;   // s_cnt[parity of the block]: written for block sb+1 while block sb's value is still being read
    __syncthreads();
    auto or_reduce = [&](uint64_t v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned lo = __shfl_xor((unsigned)(v & 0xffffffffu), o, 64);
            const unsigned hi = __shfl_xor((unsigned)(v >> 32), o, 64);
            v |= ((uint64_t)hi << 32) | lo;
        }
        return v;
    };
    auto uniform = [](uint64_t v) -> uint64_t {   // the value is wave-uniform by construction; the broadcast tells the compiler so
        return ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffu));
    };
    // word w (0..SW-1) of block sb for mask row i; 0 where the mask kernel wrote nothing (beyond the image's words, rows >= n)
    auto ld = [&](int i, int sb, int w) -> uint64_t {
        const int col = sb * SW + w;
        return (i < n && col < nw) ? m[(size_t)i * words + col] : 0ull;
    };
    // gather for block `sb` over kept[0..cnt) by the threads [t0, t0 + nt), reduced per wave into s_red[buf]
    auto gather = [&](int sb, int cnt, int t, int nt, int buf) {
        uint64_t acc[SW];
#pragma unroll
        for (int w = 0; w < SW; w++) acc[w] = 0ull;
        // FOUR kept rows per trip, all their words requested before any is used: with one row per trip every trip waited out an L2 round trip
        // (the sweep of a late chunk, ~1500 boxes kept, spent most of its 137 us here: 8 dependent round trips per block and thread)
        for (int k = t; k < cnt; k += 4 * nt) {
            uint64_t v[4][SW];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int kk = k + u * nt;
                const uint64_t* row = m + (size_t)kept[kk < cnt ? kk : k] * words + sb * SW;
#pragma unroll
                for (int w = 0; w < SW; w++) v[u][w] = (kk < cnt && sb * SW + w < nw) ? row[w] : 0ull;
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int w = 0; w < SW; w++) acc[w] |= v[u][w];
        }
#pragma unroll
        for (int w = 0; w < SW; w++) {
            const uint64_t r = or_reduce(acc[w]);
            if (lane == 0) s_red[buf * 4 * SW + wave * SW + w] = r;
        }
    };
    // the diagonal words of block `sb` (rows of block sb) and the words of block sb for the rows of block sb-1 -> LDS, by threads [t, t + nt)
    auto stage = [&](int sb, int t, int nt, int buf, bool with_nxt) {
        // every word of a thread's (up to six) entries is requested before the first is parked: one L2 round trip per block instead of six
        constexpr int NE = (SW * SW * 64 + 191) / 192;
        uint64_t vd[NE], vn[NE];
#pragma unroll
        for (int j = 0; j < NE; j++) {
            const int e = t + j * nt;
            const int l = e & 63, w = (e >> 6) % SW, q = e / (64 * SW);
            const bool in = e < SW * SW * 64;
            vd[j] = (in && w >= q) ? ld(sb * 64 * SW + 64 * q + l, sb, w) : 0ull;
            vn[j] = (in && with_nxt) ? ld((sb - 1) * 64 * SW + 64 * q + l, sb, w) : 0ull;
        }
#pragma unroll
        for (int j = 0; j < NE; j++) {
            const int e = t + j * nt;
            if (e < SW * SW * 64) {
                s_diag[buf * SW * SW * 64 + e] = vd[j];
                if (with_nxt) s_nxt[buf * SW * SW * 64 + e] = vn[j];
            }
        }
    };
    // ---- prologue: removed(sb0) from the boxes kept in the earlier chunks, and block sb0's diagonal words
    gather(sb0, cnt0, threadIdx.x, 256, sb0 & 1);
    stage(sb0, threadIdx.x, 256, sb0 & 1, false);
    uint64_t keepm[SW];                            // wave 0: keep bits of the block it has just swept
#pragma unroll
    for (int w = 0; w < SW; w++) keepm[w] = 0ull;
    bool have_prev = false;
    __syncthreads();
    int fin = sb0 & 1;
    for (int sb = sb0; sb < nsb; sb++) {
        const int buf = sb & 1;
        const int cnt = s_cnt[buf];                // boxes kept before block sb
        if (cnt >= max_keep) break;
        if (wave == 0) {
            // ---- greedy chain on block sb: LDS operands, scalar picks
            const uint64_t* red = s_red + buf * 4 * SW;
            uint64_t avail[SW], diag[SW][SW];
#pragma unroll
            for (int q = 0; q < SW; q++)
#pragma unroll
                for (int w = 0; w < SW; w++) diag[q][w] = w >= q ? s_diag[buf * SW * SW * 64 + (q * SW + w) * 64 + lane] : 0ull;
#pragma unroll
            for (int w = 0; w < SW; w++) {
                uint64_t cur = red[w] | red[SW + w] | red[2 * SW + w] | red[3 * SW + w];   // removed(sb), word w: kept before block sb-1 ...
                if (have_prev) {                                                          // ... and kept in block sb-1
                    uint64_t x = 0ull;
#pragma unroll
                    for (int q = 0; q < SW; q++) x |= ((keepm[q] >> lane) & 1ull) ? s_nxt[buf * SW * SW * 64 + (q * SW + w) * 64 + lane] : 0ull;
                    cur |= or_reduce(x);
                }
                const int base = (sb * SW + w) * 64;
                if (base >= n) cur = ~0ull;
                else if (n - base < 64) cur |= ~0ull << (n - base);
                avail[w] = uniform(~cur);
            }
            // Picks: the lowest not-yet-suppressed box is kept and its diagonal word knocks out later boxes of the word.  The loop carries only
            // `avail` (on the scalar unit: find-first-one, two lane broadcasts, three 64-bit bit operations); the keep bits fall out afterwards
            // as the boxes that were available and that no kept box's row suppressed (a kept box is never suppressed: rows only hold later
            // boxes).  max_keep is applied after the block -- later picks never change earlier ones.
            int c = cnt;
#pragma unroll
            for (int w = 0; w < SW; w++) {
                const uint64_t avail0 = avail[w];
                uint64_t supp = 0ull;
                while (avail[w] != 0ull) {
                    const int b = __builtin_ctzll(avail[w]);
                    const unsigned lo = __builtin_amdgcn_readlane((unsigned)(diag[w][w] & 0xffffffffu), b);
                    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(diag[w][w] >> 32), b);
                    const uint64_t row = ((uint64_t)hi << 32) | lo;
                    supp |= row;
                    avail[w] &= (avail[w] - 1ull) & ~row;
                }
                keepm[w] = avail0 & ~supp;
                if (c + __popcll(keepm[w]) > max_keep) {   // over the limit inside this word: keep its first max_keep - c picks
                    uint64_t km = keepm[w], first = 0ull;
                    for (int t = c; t < max_keep; t++) { first |= km & (0ull - km); km &= km - 1ull; }
                    keepm[w] = first;
                }
                c += __popcll(keepm[w]);
                // what the boxes kept in word w suppress in the block's later words: all at once, outside the chain
#pragma unroll
                for (int w2 = w + 1; w2 < SW; w2++) avail[w2] &= ~uniform(or_reduce(((keepm[w] >> lane) & 1ull) ? diag[w][w2] : 0ull));
                if (c >= max_keep) {
#pragma unroll
                    for (int w2 = w + 1; w2 < SW; w2++) avail[w2] = 0ull;
                }
            }
            int before = cnt;
#pragma unroll
            for (int w = 0; w < SW; w++) {          // kept boxes of this block, in index order
                if ((keepm[w] >> lane) & 1ull) kept[before + __popcll(keepm[w] & ((1ull << lane) - 1ull))] = (sb * SW + w) * 64 + lane;
                before += __popcll(keepm[w]);
            }
            if (lane == 0) s_cnt[buf ^ 1] = c;
            have_prev = true;
        } else if (sb + 1 < nsb) {
            // ---- waves 1-3: everything block sb+1 needs that does not depend on block sb's chain
            gather(sb + 1, cnt, threadIdx.x - 64, 192, buf ^ 1);
            stage(sb + 1, threadIdx.x - 64, 192, buf ^ 1, true);
        }
        if (wave == 0 && sb + 1 < nsb && lane < SW) s_red[(buf ^ 1) * 4 * SW + lane] = 0ull;   // wave 0 takes no part in that gather
        fin = buf ^ 1;
        __syncthreads();
    }
    const int total = s_cnt[fin];
    for (int k = cnt0 + threadIdx.x; k < total; k += 256) kp[k] = kept[k];
    if (threadIdx.x == 0) n_keep[img] = total;
}

// ---- _C.nms(dets, scores, thr) as ONE call on UNSORTED boxes (round 5): rank the scores (abr_sort_scores_desc: the proposal ranking's own key /
// partition / chunk-sort / merge kernels on order-preserving keys, ties by ascending index = torch.sort(stable, descending)), gather the boxes in
// that order, mask + sweep as above, then hand back the survivors' ORIGINAL indices in ascending order (nms.cu:127-130, nms_cpu.cpp:66) by
// marking them in a flag array and compacting it -- no second sort.
__global__ __launch_bounds__(256) void nms_gather_boxes_kernel(const float4* __restrict__ dets, const int64_t* __restrict__ order, int n, float4* __restrict__ out,
                                                               int32_t* __restrict__ counts) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = dets[order[i]];
    if (i == 0) counts[0] = n;
}
__global__ __launch_bounds__(1024) void nms_mark_compact_kernel(const int64_t* __restrict__ order, const int32_t* __restrict__ keep, const int32_t* __restrict__ n_keep,
                                                                int n, unsigned char* __restrict__ mark, int64_t* __restrict__ out) {
    __shared__ int s_wave[16];
    __shared__ int s_base;
    const int k = n_keep[0];
    for (int i = threadIdx.x; i < n; i += 1024) mark[i] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < k; i += 1024) mark[order[keep[i]]] = 1;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i0 = 0; i0 < n; i0 += 1024) {          // ascending index: a running prefix count over 1024-element slabs
        const int i = i0 + threadIdx.x;
        const int m = i < n ? mark[i] : 0;
        const unsigned long long b = __ballot(m);
        const int before = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wave] = __popcll(b);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wave; w++) off += s_wave[w];
        if (m) out[off + before] = i;
        __syncthreads();
        if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < 16; w++) t += s_wave[w]; s_base += t; }
        __syncthreads();
    }
}

}  // namespace

extern "C" int abr_sort_scores_desc(const float* scores, int n, int64_t* order, void* stream);   // topk.hip
extern "C" int64_t abr_sort_scores_max_n(void);

extern "C" int64_t abr_nms_workspace_bytes(int N, int n_max);
extern "C" int abr_nms_sorted_batched(const float* boxes, const int32_t* counts, int N, int n_max, float thr, int strict_gt, int max_keep, int32_t* keep,
                                      int32_t* n_keep, void* workspace, int64_t workspace_bytes, void* stream);

extern "C" int64_t abr_nms_unsorted_workspace_bytes(int n) {
    if (n <= 0) return 64;
    // order [n] int64, sorted boxes [n] float4, counts, keep [n] int32, marks [n], then the mask workspace
    return (int64_t)n * 8 + (int64_t)n * 16 + 64 + (((int64_t)n * 4 + 63) & ~63) + (((int64_t)n + 63) & ~63) + abr_nms_workspace_bytes(1, n);
}

extern "C" int abr_nms(const float* dets, const float* scores, int n, float thr, int strict_gt, int64_t* keep_out, int32_t* n_keep_out, void* workspace,
                       int64_t workspace_bytes, void* stream) {
    ABR_REQUIRE(n >= 0 && n_keep_out, "nms: bad args");
    hipStream_t st = abr::as_stream(stream);
    if (n == 0) {
        if (hipMemsetAsync(n_keep_out, 0, sizeof(int32_t), st) != hipSuccess) return ABR_E_LAUNCH;
        return ABR_OK;
    }
    ABR_REQUIRE(dets && scores && keep_out && workspace, "nms: null pointer");
    ABR_REQUIRE(n <= abr_sort_scores_max_n(), "nms: at most %d boxes per call (the in-LDS merge of the score sort)", (int)abr_sort_scores_max_n());
    ABR_REQUIRE(workspace_bytes >= abr_nms_unsorted_workspace_bytes(n), "nms: workspace too small");
    ABR_REQUIRE((reinterpret_cast<uintptr_t>(dets) & 15) == 0, "nms: dets must be 16-byte aligned");
    char* w = static_cast<char*>(workspace);
    int64_t* order = reinterpret_cast<int64_t*>(w);                       w += (size_t)n * 8;
    float4* sorted = reinterpret_cast<float4*>(w);                         w += (size_t)n * 16;
    int32_t* counts = reinterpret_cast<int32_t*>(w);                       w += 64;
    int32_t* keep = reinterpret_cast<int32_t*>(w);                         w += ((size_t)n * 4 + 63) & ~(size_t)63;
    unsigned char* mark = reinterpret_cast<unsigned char*>(w);             w += ((size_t)n + 63) & ~(size_t)63;
    int rc = abr_sort_scores_desc(scores, n, order, stream);
    if (rc != ABR_OK) return rc;
    nms_gather_boxes_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(reinterpret_cast<const float4*>(dets), order, n, sorted, counts);
    rc = abr_nms_sorted_batched(reinterpret_cast<const float*>(sorted), counts, 1, n, thr, strict_gt, n, keep, n_keep_out, w, abr_nms_workspace_bytes(1, n), stream);
    if (rc != ABR_OK) return rc;
    nms_mark_compact_kernel<<<1, 1024, 0, st>>>(order, keep, n_keep_out, n, mark, keep_out);
    ABR_CHECK_LAUNCH("nms");
    return ABR_OK;
}

extern "C" int64_t abr_nms_workspace_bytes(int N, int n_max) {
    const int64_t words = (n_max + 63) / 64;
    return (int64_t)N * n_max * words * 8;
}

extern "C" int abr_nms_sorted_batched(const float* boxes, const int32_t* counts, int N, int n_max, float thr,
                                      int strict_gt, int max_keep, int32_t* keep, int32_t* n_keep, void* workspace,
                                      int64_t workspace_bytes, void* stream) {
    ABR_REQUIRE(N >= 0 && n_max >= 0 && max_keep >= 0, "nms: bad shape");
    if (N == 0) return ABR_OK;
    ABR_REQUIRE(counts && n_keep, "nms: null counts");
    hipStream_t st = abr::as_stream(stream);
    if (n_max == 0 || max_keep == 0) {
        if (hipMemsetAsync(n_keep, 0, sizeof(int32_t) * N, st) != hipSuccess) return ABR_E_LAUNCH;
        return ABR_OK;
    }
    ABR_REQUIRE(boxes && keep && workspace, "nms: null pointer");
    ABR_REQUIRE(workspace_bytes >= abr_nms_workspace_bytes(N, n_max), "nms: workspace too small");
    const int words = (n_max + 63) / 64;
    // The mask rows are only partially written (the words the sweep reads: see nms_mask_chunk_kernel).  No memset of the mask.
    const size_t lds = (((size_t)max_keep * 4 + 15) & ~(size_t)15) + (2 * 4 * SW + 4 * SW * SW * 64 + 1) * 8 + 16;   // s_red, s_diag, s_nxt, s_cnt[2]
    ABR_REQUIRE(lds <= 158 * 1024, "nms: max_keep too large for the LDS keep list");
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nms_sweep_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (hipMemsetAsync(n_keep, 0, sizeof(int32_t) * N, st) != hipSuccess) return ABR_E_LAUNCH;
    const int chunks = (n_max + kChunk - 1) / kChunk;
    const int kept_tiles = (max_keep + 63) / 64;
    for (int c = 0; c < chunks; c++) {
        // (launches for chunks past the point where every image has its max_keep boxes, or past its box count, return at once)
        dim3 grid(kChunk / 64, kept_tiles + kChunk / 64, N);
        nms_mask_chunk_kernel<<<grid, 64, 0, st>>>(boxes, counts, n_max, words, thr, strict_gt, c, kept_tiles, max_keep, keep, n_keep,
                                                   (uint64_t*)workspace);
        nms_sweep_kernel<<<N, 256, lds, st>>>((const uint64_t*)workspace, counts, n_max, words, max_keep, c, keep, n_keep);
    }
    ABR_CHECK_LAUNCH("nms_sweep");
    return ABR_OK;
}
