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

// grid = (col tiles, row tiles, N); block = 64 (one wave)
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes, const int32_t* __restrict__ counts,
                                                       int n_max, int words, float thr, int strict_gt,
                                                       uint64_t* __restrict__ mask) {
    const int img = blockIdx.z, rt = blockIdx.y, ct = blockIdx.x;
    if (ct < rt) return;  // lower triangle never read
    const int n = counts[img];
    if (rt * 64 >= n || ct * 64 >= n) return;
    __shared__ float4 cb[64];
    const float4* b = reinterpret_cast<const float4*>(boxes) + (size_t)img * n_max;
    const int lane = threadIdx.x;
    const int cj = ct * 64 + lane;
    cb[lane] = cj < n ? b[cj] : make_float4(0, 0, -1, -1);
    __syncthreads();
    const int ri = rt * 64 + lane;
    if (ri >= n) return;
    const float4 a = b[ri];
    const float area_a = (a.z - a.x + 1) * (a.w - a.y + 1);
    uint64_t bits = 0;
    const int jstart = (rt == ct) ? lane + 1 : 0;
    const int jend = min(64, n - ct * 64);
    for (int j = jstart; j < jend; j++)
        if (suppresses(a, area_a, cb[j], thr, strict_gt)) bits |= 1ull << j;
    mask[((size_t)img * n_max + ri) * words + ct] = bits;
}

// grid = N, block = 256.  The kept list lives in LDS (max_keep ints).
// Column formulation: when the sweep reaches 64-box block bi, the boxes of that block already suppressed are
//   removed(bi) = OR over every box i kept so far of mask[i][bi]
// -- one 8-byte load per kept box, independent of each other, spread over the 256 threads and OR-reduced (<= max_keep/256
// loads per thread).  (The earlier row formulation OR-ed each kept row into all later words: ~188x more loads.)
// The sweep is a chain of dependent steps, so its time is latency per block x blocks; two things keep that latency short:
//   * the loads for block bi+1 are issued BEFORE block bi's greedy chain runs: the gather over the boxes kept before bi does not
//     depend on it, and what block bi itself will add is fetched unconditionally (lane b of wave 0 loads mask[bi*64+b][bi+1], and
//     the diagonal word of block bi+1) and selected by the keep bits afterwards;
//   * the greedy chain visits only the boxes it keeps (count-trailing-zeros over the not-yet-suppressed bits, one lane broadcast of
//     the kept box's diagonal row each) instead of stepping through all 64 -- at most max_keep + blocks iterations per image.
__global__ __launch_bounds__(256) void nms_sweep_kernel(const uint64_t* __restrict__ mask, const int32_t* __restrict__ counts,
                                                         int n_max, int words, int max_keep, int32_t* __restrict__ keep,
                                                         int32_t* __restrict__ n_keep) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* kept = reinterpret_cast<int*>(smem);                                  // [max_keep]
    uint64_t* s_red = reinterpret_cast<uint64_t*>(smem + (((size_t)max_keep * 4 + 15) & ~(size_t)15));  // [2][4] gathered words (double-buffered) + [1]
    int* s_cnt = reinterpret_cast<int*>(s_red + 9);
    const int img = blockIdx.x;
    const int n = counts[img];
    const uint64_t* m = mask + (size_t)img * n_max * words;
    int32_t* kp = keep + (size_t)img * max_keep;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) *s_cnt = 0;
    if (threadIdx.x < 8) s_red[threadIdx.x] = 0ull;
    __syncthreads();
    const int nblk = (n + 63) / 64;
    auto or_reduce = [&](uint64_t v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned lo = __shfl_xor((unsigned)(v & 0xffffffffu), o, 64);
            const unsigned hi = __shfl_xor((unsigned)(v >> 32), o, 64);
            v |= ((uint64_t)hi << 32) | lo;
        }
        return v;
    };
    // wave 0 state: the diagonal word of the current block (bits j > lane) and what its own kept boxes add to the next block's word
    uint64_t diag = 0ull, own_next = 0ull;
    if (wave == 0 && nblk > 0) { const int i = lane; diag = i < n ? m[(size_t)i * words] : 0ull; }
    for (int bi = 0; bi < nblk; bi++) {
        const int cnt = *s_cnt;                    // boxes kept before block bi
        if (cnt >= max_keep) break;
        const bool more = bi + 1 < nblk;
        // ---- loads for block bi+1, independent of block bi's chain
        uint64_t acc = 0ull;
        if (more)
            for (int k = threadIdx.x; k < cnt; k += 256) acc |= m[(size_t)kept[k] * words + bi + 1];
        uint64_t nxt = 0ull, diag_next = 0ull;
        if (wave == 0 && more) {
            const int i = bi * 64 + lane, i2 = i + 64;
            nxt = i < n ? m[(size_t)i * words + bi + 1] : 0ull;          // what box i adds to block bi+1 IF it is kept
            diag_next = i2 < n ? m[(size_t)i2 * words + bi + 1] : 0ull;
        }
        if (wave == 0) {
            // ---- greedy chain on block bi: registers and lane broadcasts only
            const uint64_t* red = s_red + (bi & 1) * 4;
            uint64_t cur = red[0] | red[1] | red[2] | red[3] | own_next;      // removed(bi)
            const int valid = min(64, n - bi * 64);
            if (valid < 64) cur |= ~0ull << valid;
            uint64_t avail = ~cur, keepm = 0ull;
            int c = cnt;
            while (avail != 0ull && c < max_keep) {
                const int b = __builtin_amdgcn_readfirstlane(__builtin_ctzll(avail));
                keepm |= 1ull << b;
                c++;
                const unsigned lo = __builtin_amdgcn_readlane((unsigned)(diag & 0xffffffffu), b);
                const unsigned hi = __builtin_amdgcn_readlane((unsigned)(diag >> 32), b);
                avail &= ~((((uint64_t)hi << 32) | lo) | (1ull << b));
            }
            if ((keepm >> lane) & 1ull) {           // kept boxes of this block, in index order
                const int pos = cnt + __popcll(keepm & ((1ull << lane) - 1ull));
                kept[pos] = bi * 64 + lane;
                kp[pos] = bi * 64 + lane;
            }
            if (lane == 0) *s_cnt = c;
            own_next = or_reduce(((keepm >> lane) & 1ull) ? nxt : 0ull);
            diag = diag_next;
        }
        acc = or_reduce(acc);
        if (lane == 0) s_red[((bi + 1) & 1) * 4 + wave] = acc;
        __syncthreads();
    }
    if (threadIdx.x == 0) n_keep[img] = *s_cnt;
}

}  // namespace

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
    // The mask rows are only partially written (upper triangle, rows < count): words left of the diagonal are
    // never read, words right of it are always written for rows < n.  No memset needed.
    dim3 grid(words, words, N);
    nms_mask_kernel<<<grid, 64, 0, st>>>(boxes, counts, n_max, words, thr, strict_gt, (uint64_t*)workspace);
    ABR_CHECK_LAUNCH("nms_mask");
    const size_t lds = (((size_t)max_keep * 4 + 15) & ~(size_t)15) + 9 * 8 + 16;
    ABR_REQUIRE(lds <= 150 * 1024, "nms: max_keep too large for the LDS keep list");
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nms_sweep_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    nms_sweep_kernel<<<N, 256, lds, st>>>((const uint64_t*)workspace, counts, n_max, words, max_keep, keep, n_keep);
    ABR_CHECK_LAUNCH("nms_sweep");
    return ABR_OK;
}
