// Convolution weight gradient on the fp32 matrix cores.
//
//   dW[n, k] += scale[n] * sum_m gy[m, n] * A[m, k]      n = cout, k = (r, s, cin), m = (b, ho, wo)
//
// Replaces cuDNN wgrad for the trainable convs of the reference (layer2/3, RPN head, layer4 head,
// predictor FCs: modeling/backbone/resnet.py:261-323, modeling/rpn/rpn.py:83-85, roi_box_predictors.py:17-19).
// `scale` is the FrozenBatchNorm2d scale that the forward epilogue applied after the conv
// (layers/batch_norm.py:27-31), so d/dW of (conv*scale+bias) carries it per output channel.
//
// Shape of the problem on this path: the OUTPUT is small (<= 1024 x 9216) and the REDUCTION axis m is huge
// (9.5k-37k pixels/RoI-cells), the opposite of forward.  So: a workgroup owns a 128(n) x 128(k) tile of dW
// and a contiguous slice of m (split-M); partial tiles are parked in a scratch with plain coalesced stores and summed in
// split order by wgrad_reduce_kernel (deterministic; ABR_WGRAD_REDUCE=0: fp32 atomics straight into the gradient buffer, the
// round-1 scheme).  `dw +=` also folds in the accumulation over the two RoI passes of one step (the 512-RoI detection pass and
// the 64-RoI distillation pass share the head weights).
// Both operands arrive m-major (gy rows, NHWC pixels), i.e. the MFMA's reduction index is the SLOW axis in memory.
// fp32 kernel: rows are staged as-is into LDS ([m][128], 512 B coalesced per row) and a lane picks its operand with one
// ds_read_b64 = two adjacent n (or k) of row m; accumulator tile t then holds n = base + 2i + t -- undone in the epilogue
// addressing, no transpose is ever materialised.  bf16x6 kernel (the default arithmetic): the loader transposes in registers
// (conv_wgrad_x6_kernel below).  Operand fetch is buffer loads with hardware range checking (no branches), single-buffered LDS
// (3 workgroups/CU); wide stride-1 3x3 convs go through the Winograd domain instead (36 batched plain GEMMs over the tile axis,
// conv_winograd.hip).
#include <algorithm>

#include <map>
#include <mutex>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int TN_ = 128;  // dW tile rows  (cout)
constexpr int TK_ = 128;  // dW tile cols  (r,s,cin)
constexpr int MR = 32;    // m rows per stage

// n / d for 0 <= n < 2^31 with a precomputed multiplier (a generic 32-bit division is ~40 VALU instructions, this is 3; the
// 3x3 path needs two per staged row per stage)
struct FastDiv {
    unsigned mul, shr, d;
    __host__ void init(unsigned div) {
        d = div;
        shr = 0;
        while ((1u << shr) < div) shr++;
        mul = (unsigned)((((unsigned long long)1 << 32) * (((unsigned long long)1 << shr) - div)) / div + 1);
    }
    __device__ __forceinline__ void divmod(unsigned n, unsigned& q, unsigned& r) const {
        q = (__umulhi(n, mul) + n) >> shr;
        r = n - q * d;
    }
};

struct WgP {
    unsigned long long* prof_ts;   // bench profiling: {first start, last end} stamp slot of this launch (abr::prof_stamp_slot) or nullptr
    int B, H, W, Cin, Cout, R, S, stride, pad, Ho, Wo;
    int M, K;
    int tiles_n, tiles_k, splits, mt_per_split;
    int plain;  // 1x1, stride 1, pad 0: A row m is x + m*Cin
    FastDiv d_howo, d_wo;
    const float* scale;
    unsigned x_bytes, gy_bytes;  // extents for the buffer-load range check (per batch)
    int overwrite;               // splits == 1 only: plain stores instead of atomic accumulation (the caller wants dw = ..., not +=)
    int nbatch, tiles_pb;        // batched mode (the 36 Winograd-domain gradients): tile -> (batch, tile inside the batch)
    long x_bs, gy_bs, dw_bs;
    int math;                    // ABR_MATH_F32 or ABR_MATH_BF16X6 (the bf16 mode has its own launch)
    unsigned* x6_flags;          // bf16x6: device word of the range guard (abr::x6_flags_ptr)
    int tile_fast;               // workgroup order: output tile fastest (1, default) or row slice fastest (0)
    // split-M partial sums (splits > 1): parked in `ws` (one 64 KB unit per workgroup, plain coalesced stores) and summed in split order by
    // wgrad_reduce_kernel right behind this launch -- deterministic, and one write per output element instead of `splits` fp32 atomics
    // (256 workgroups x 16 K atomics = 4 M per launch cost ~30 us whatever the shape).  nullptr: atomics into dw
    float* ws;
    int final_store;             // the reduction stores (dw = sum: the Winograd-domain dU, never zero-filled) instead of dw += sum
    int map4;                    // accumulator interleave of conv_wgrad_x6_kernel (n = n0 + 4 i + 2 wm + tm) instead of n0 + wm*64 + 2 i + tm
    // ABR_MATH_F16X3: the amax words of the two operands (their split scales; dW = s_gy s_x * sum); null otherwise
    const unsigned long long* gy_amax;
    const unsigned long long* x_amax;
    unsigned gy_epoch, x_epoch;
    unsigned long long* h3_stats;   // f16x3 range statistics (abr::h3_stats_ptr) or nullptr
    // round 5: with tickets the partial tiles are summed by the LAST workgroup of each output tile to arrive (in split order, its own partial
    // included: the same sum whoever is last) instead of by a wgrad_reduce_kernel launch behind every split launch (41 launches per step).
    // One counter per output tile, zero on entry, left zero.  nullptr: the reduction kernel follows.
    unsigned* tickets;
    unsigned ws_bytes;              // extent of `ws` for the coherent buffer accesses
};

// f16x3: the factors s_gy, s_x the accumulators carry (1, 1 in every other arithmetic); applied one after the other: their PRODUCT could leave
// fp32's normal range although neither the sum nor the result does
__device__ __forceinline__ float2 wg_operand_scale(const WgP& p) {
    if (!p.gy_amax) return make_float2(1.f, 1.f);
    float sg, ig, sx, ix;
    abr::h3_scales(abr::h3_amax_load(p.gy_amax, p.gy_epoch), sg, ig);
    abr::h3_scales(abr::h3_amax_load(p.x_amax, p.x_epoch), sx, ix);
    return make_float2(sg, sx);
}

typedef f32x16 wg_f32x16;

// Epilogue shared by the three weight-gradient kernels: tile (tm,tn) element (row i, col j) of a wave is
// dW[n0 + wm*64 + 2i + tm][k0 + wn*64 + 2j + tn].  unit = global tile index * splits + split.
constexpr unsigned kPartBytes = 16u * 256u * 16u;   // one partial tile (128 x 128 fp32), thread-major: quad (t,c) of thread tid at (t*4+c)*4096 + tid*16

__device__ __forceinline__ void wgrad_store_tile(const WgP& p, const float (&v)[4], int tm, int tn, int c, int n0, int k0, int wm, int wn, int l31, int lh,
                                                 bool store, float* __restrict__ dw, const float2 osc = make_float2(1.f, 1.f)) {
    const int k = p.map4 ? k0 + 4 * l31 + 2 * wn + tn : k0 + wn * 64 + 2 * l31 + tn;
    if (k >= p.K) return;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int r = 4 * c + e;
        const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int n = p.map4 ? n0 + 4 * i + 2 * wm + tm : n0 + wm * 64 + 2 * i + tm;
        if (n >= p.Cout) continue;
        const float sc = (p.scale ? p.scale[n] : 1.f) * fmaxf(osc.x, osc.y);   // (osc: powers of two -- the same value as scaling the sum first;
        const float val = (v[e] * fminf(osc.x, osc.y)) * sc;                 //  the smaller factor first: conv_igemm.hip::epilogue_rows)
        if (store) dw[(size_t)n * p.K + k] = val;
        else unsafeAtomicAdd(dw + (size_t)n * p.K + k, val);   // dw += (other launches may add to the same dw)
    }
}

__device__ __forceinline__ void wgrad_finish(const WgP& p, f32x16 (&acc)[2][2], int n0, int k0, int wm, int wn, int l31, int lh, int tid,
                                             int gtile, int split, float* __restrict__ dw) {
    if (p.ws && !p.tickets) {   // park the partial sums; wgrad_reduce_kernel adds them up
        float4* dst = reinterpret_cast<float4*>(reinterpret_cast<char*>(p.ws) + ((size_t)gtile * p.splits + split) * kPartBytes) + tid;
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const f32x16& a = acc[t >> 1][t & 1];
                dst[(t * 4 + c) * 256] = make_float4(a[4 * c], a[4 * c + 1], a[4 * c + 2], a[4 * c + 3]);
            }
        return;   // (the caller stamps the end: wgrad_finish is the kernel's last statement)
    }
    if (p.ws) {
        // park through system-coherent buffer stores (written through the XCD-local L2: no cache maintenance around the ticket, as in
        // conv_igemm_kernel's split-K), take the tile's ticket; the last arrival re-reads ALL partials in split order and finishes the tile
        constexpr int kCoherent = 0x11;   // sc0 | sc1
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t rws = __builtin_amdgcn_make_buffer_rsrc(p.ws, 0, p.ws_bytes, 0x00020000);
        const unsigned my_off = (unsigned)(((size_t)gtile * p.splits + split) * kPartBytes) + (unsigned)tid * 16u;
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const f32x16& a = acc[t >> 1][t & 1];
                const u32x4 v = {__float_as_uint(a[4 * c]), __float_as_uint(a[4 * c + 1]), __float_as_uint(a[4 * c + 2]), __float_as_uint(a[4 * c + 3])};
                __builtin_amdgcn_raw_buffer_store_b128(v, rws, (int)(my_off + (unsigned)(t * 4 + c) * 4096u), 0, kCoherent);
            }
        // every write-through store of this wave has landed before the ticket is taken.  The explicit wait matters: a workgroup-scope release
        // alone need not drain vmcnt (all waves of a workgroup share one L1), and with 16 short units per tile the last arrival then read partials
        // still in flight (round 5: wrong sums in 7-65 % of a tiny model's outputs, run to run).  No cache maintenance: sc0|sc1 on both sides.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        __shared__ int s_last;
        if (tid == 0) s_last = atomicAdd(p.tickets + gtile, 1u) == (unsigned)p.splits - 1u;
        __syncthreads();
        if (!s_last) return;
        if (tid == 0) p.tickets[gtile] = 0u;
        const float2 osc = wg_operand_scale(p);
        const unsigned base = (unsigned)((size_t)gtile * p.splits * kPartBytes) + (unsigned)tid * 16u;
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
                int q = 0;
                for (; q + 4 <= p.splits; q += 4) {   // four loads in flight; the additions stay in split order
                    u32x4 v[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = __builtin_amdgcn_raw_buffer_load_b128(rws, (int)(base + (unsigned)(q + e) * kPartBytes + (unsigned)(t * 4 + c) * 4096u), 0, kCoherent);
#pragma unroll
                    for (int e = 0; e < 4; e++) { sum.x += __uint_as_float(v[e].x); sum.y += __uint_as_float(v[e].y); sum.z += __uint_as_float(v[e].z); sum.w += __uint_as_float(v[e].w); }
                }
                for (; q < p.splits; q++) {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rws, (int)(base + (unsigned)q * kPartBytes + (unsigned)(t * 4 + c) * 4096u), 0, kCoherent);
                    sum.x += __uint_as_float(v.x); sum.y += __uint_as_float(v.y); sum.z += __uint_as_float(v.z); sum.w += __uint_as_float(v.w);
                }
                const float vv[4] = {sum.x, sum.y, sum.z, sum.w};
                wgrad_store_tile(p, vv, t >> 1, t & 1, c, n0, k0, wm, wn, l31, lh, p.final_store != 0, dw, osc);
            }
        return;
    }
    const float2 osc = wg_operand_scale(p);
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const f32x16& a = acc[t >> 1][t & 1];
            const float v[4] = {a[4 * c], a[4 * c + 1], a[4 * c + 2], a[4 * c + 3]};
            wgrad_store_tile(p, v, t >> 1, t & 1, c, n0, k0, wm, wn, l31, lh, p.overwrite != 0, dw, osc);
        }
}

// One workgroup per (output tile, accumulator quad): thread tid sums quad (t,c) of the tile's `splits` partials in split order and
// writes the four elements the producing thread tid of conv_wgrad*_kernel would have written.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgP p, float* __restrict__ dw_) {
    const int gtile = blockIdx.x >> 4, quad = blockIdx.x & 15;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float4* src = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p.ws) + (size_t)gtile * p.splits * kPartBytes) + quad * 256 + tid;
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    int q = 0;
    for (; q + 4 <= p.splits; q += 4) {   // four loads in flight; the additions stay in split order
        const float4 v0 = src[(size_t)q * (kPartBytes / 16)], v1 = src[(size_t)(q + 1) * (kPartBytes / 16)];
        const float4 v2 = src[(size_t)(q + 2) * (kPartBytes / 16)], v3 = src[(size_t)(q + 3) * (kPartBytes / 16)];
        sum.x += v0.x; sum.y += v0.y; sum.z += v0.z; sum.w += v0.w;
        sum.x += v1.x; sum.y += v1.y; sum.z += v1.z; sum.w += v1.w;
        sum.x += v2.x; sum.y += v2.y; sum.z += v2.z; sum.w += v2.w;
        sum.x += v3.x; sum.y += v3.y; sum.z += v3.z; sum.w += v3.w;
    }
    for (; q < p.splits; q++) {
        const float4 v = src[(size_t)q * (kPartBytes / 16)];
        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
    float* dw = dw_;
    int tile = gtile;
    if (p.nbatch > 1) {
        const int bt = tile / p.tiles_pb;
        tile -= bt * p.tiles_pb;
        dw += bt * p.dw_bs;
    }
    const int n0 = (tile % p.tiles_n) * TN_, k0 = (tile / p.tiles_n) * TK_;
    const float v[4] = {sum.x, sum.y, sum.z, sum.w};
    const int t = quad >> 2, c = quad & 3;
    wgrad_store_tile(p, v, t >> 1, t & 1, c, n0, k0, wave >> 1, wave & 1, lane & 31, lane >> 5, p.final_store != 0, dw, wg_operand_scale(p));
}

// SB: single-buffered operand LDS (two barriers per stage, 32 KB instead of 64 KB -> a third resident workgroup per CU)
template <bool SB>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgP p, const float* __restrict__ x_,
                                                          const float* __restrict__ gy_, float* __restrict__ dw_) {
    const float* x = x_;
    const float* gy = gy_;
    float* dw = dw_;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    abr::prof_stamp_begin(p.prof_ts);
    constexpr int NBUF = SB ? 1 : 2;
    float* Gs = smem;                       // [NBUF][MR][TN_]
    float* As = smem + NBUF * MR * TN_;     // [NBUF][MR][TK_]

    const unsigned bid = abr::xcd_remap(blockIdx.x, gridDim.x);
    // workgroup -> (row slice, output tile): the output tile is the FAST index, so the workgroups one XCD runs side by side (a
    // contiguous bid range, abr::xcd_remap) cover ALL output tiles of a few row slices.  They stream the same rows of gy and x in
    // lockstep: every operand element is then fetched from HBM once per XCD and served to the other tiles from that XCD's L2,
    // instead of being re-fetched by each of the N/128 (resp. K/128) tiles that need it (slice-fast order: ~3x the HBM traffic on the
    // 9576 x {1024 x 256} gradients, which made them bandwidth-bound).  ABR_WGRAD_TILE_FAST=0 restores the old order.
    const int total_tiles = p.tiles_pb * (p.nbatch > 1 ? p.nbatch : 1);
    const int split = p.tile_fast ? bid / total_tiles : bid % p.splits;
    int tile = p.tile_fast ? bid % total_tiles : bid / p.splits;
    const int gtile = tile;
    if (p.nbatch > 1) {
        const int bt = tile / p.tiles_pb;
        tile -= bt * p.tiles_pb;
        x += bt * p.x_bs; gy += bt * p.gy_bs; dw += bt * p.dw_bs;
    }
    const int tile_n = tile % p.tiles_n, tile_k = tile / p.tiles_n;
    const int n0 = tile_n * TN_, k0 = tile_k * TK_;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;  // wave tile: n in [wm*64, +64), k in [wn*64, +64)

    // staging: slot = tid + 256*i -> row = tid/32 + 8*i, q = tid%32 (16 B column group)
    const int q = tid & 31, r8 = tid >> 5;
    const int gn = n0 + q * 4;
    const bool g_ok = gn < p.Cout;  // Cout % 4 == 0
    const int ak = k0 + q * 4;
    const bool k_ok = ak < p.K;
    int fr = 0, fs = 0, fc = 0;
    if (k_ok) {
        const int rs = ak / p.Cin;
        fc = ak % p.Cin;
        fr = rs / p.S;
        fs = rs % p.S;
    }
    const int mt0 = split * p.mt_per_split;
    const int mt1 = min(mt0 + p.mt_per_split, (p.M + MR - 1) / MR);

    // Operand fetch through buffer loads (see conv_igemm.hip): rows past M fall outside the descriptor's range and read as zeros,
    // masked columns / halo taps get the out-of-range offset kOOB; 32-bit offsets, no branches.  gy rows and the A rows of a plain
    // 1x1 conv advance by a constant per stage: one add per load.
    constexpr unsigned kOOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rgy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gy), 0, p.gy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rxx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, p.x_bytes, 0x00020000);
    const unsigned g_voff = g_ok ? (unsigned)(r8 * p.Cout + gn) * 4u : kOOB;
    const unsigned a_voff_plain = k_ok ? (unsigned)(r8 * p.Cin + fc) * 4u : kOOB;
    const int a_const = ((fr - p.pad) * p.W + (fs - p.pad)) * p.Cin + fc;  // non-plain: tap offset of this thread's k column
    float4 rg[4], ra[4];
    auto fetch = [&](__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) -> float4 {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    auto load_tile = [&](int mt) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int mrow = mt * MR + 8 * i;  // scalar part of the row index (the lane adds r8)
            // the row goes into the VECTOR offset: that is the part the hardware range-checks (kOOB + row stays >= 2^31)
            rg[i] = fetch(rgy, g_voff + (unsigned)(mrow * p.Cout) * 4u, 0);
            if (p.plain) {
                ra[i] = fetch(rxx, a_voff_plain + (unsigned)(mrow * p.Cin) * 4u, 0);
            } else {
                const int m = mrow + r8;
                unsigned b, rem, ho, wo;
                p.d_howo.divmod((unsigned)m, b, rem);   // m >= M gives b >= B: the offset lands past the tensor -> zeros
                p.d_wo.divmod(rem, ho, wo);
                const int hi = (int)ho * p.stride - p.pad + fr, wi = (int)wo * p.stride - p.pad + fs;
                const bool ok = k_ok & (m < p.M) & ((unsigned)hi < (unsigned)p.H) & ((unsigned)wi < (unsigned)p.W);  // no short circuit
                const int off = (((int)b * p.H + (int)ho * p.stride) * p.W + (int)wo * p.stride) * p.Cin + a_const;
                ra[i] = fetch(rxx, ok ? (unsigned)off * 4u : kOOB, 0);
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            *reinterpret_cast<float4*>(Gs + (buf * MR + r8 + 8 * i) * TN_ + q * 4) = rg[i];
            *reinterpret_cast<float4*>(As + (buf * MR + r8 + 8 * i) * TK_ + q * 4) = ra[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    const int l31 = lane & 31, lh = lane >> 5;
    auto compute_tile = [&](int cur) {
        const float* g = Gs + cur * MR * TN_ + wm * 64 + 2 * l31;
        const float* a = As + cur * MR * TK_ + wn * 64 + 2 * l31;
#pragma unroll
        for (int s = 0; s < MR / 2; s++) {
            const float2 fg = *reinterpret_cast<const float2*>(g + (2 * s + lh) * TN_);
            const float2 fa = *reinterpret_cast<const float2*>(a + (2 * s + lh) * TK_);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fg.x, fa.x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fg.x, fa.y, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fg.y, fa.x, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fg.y, fa.y, acc[1][1], 0, 0, 0);
        }
    };
    if (mt0 < mt1) {
        load_tile(mt0);
        store_tile(0);
        __syncthreads();
        int mt = mt0;
        if (SB) {
            for (; mt + 1 < mt1; mt++) {  // steady state: one basic block; the last stage is peeled (nothing to fetch for it)
                load_tile(mt + 1);
                __builtin_amdgcn_sched_barrier(0);  // fetches stay ahead of the MFMA stream
                compute_tile(0);
                __syncthreads();
                store_tile(0);
                __syncthreads();
            }
            compute_tile(0);
        } else {  // double-buffered: stage mt+2 is fetched right behind the ds_writes of stage mt+1 (see conv_igemm.hip)
            if (mt + 1 < mt1) load_tile(mt + 1);
            for (; mt + 2 < mt1; mt++) {
                const int cur = (mt - mt0) & 1;
                compute_tile(cur);
                store_tile(cur ^ 1);
                load_tile(mt + 2);
                __syncthreads();
            }
            if (mt + 1 < mt1) {
                const int cur = (mt - mt0) & 1;
                compute_tile(cur);
                store_tile(cur ^ 1);
                __syncthreads();
                mt++;
            }
            compute_tile((mt - mt0) & 1);
        }
    }

    // epilogue: tile (tm,tn) element (row i, col j) is dW[n0 + wm*64 + 2i + tm][k0 + wn*64 + 2j + tn]
    wgrad_finish(p, acc, n0, k0, wm, wn, l31, lh, tid, gtile, split, dw);
    abr::prof_stamp_end(p.prof_ts);
}

// ------------------------------------------------------------------------------------------------------------------------
// bf16 math mode (abr_conv_desc::math == ABR_MATH_BF16): the same split-M tile scheme on v_mfma_f32_32x32x16_bf16, fp32
// accumulate, fp32 tensors in memory.  The reduction index m is still the slow axis in memory, and this MFMA wants 8 CONSECUTIVE
// reduction elements per lane -- but the order of k inside an MFMA is a free permutation as long as both operands agree, so no
// transpose is needed: a thread fetches the same 16 B column group of rows m and m+1, rounds both to bf16 (RNE) and stores the
// four (m, m+1) pairs as one ds_write_b128 into an LDS tile of [m-pair][column] dwords; a lane's operand is four ds_read_b64
// (m-pairs 8s+4h+{0..3}, columns 2*l31 and 2*l31+1), whose low / high dwords are the operands of the even / odd column sub-tile.
// Stage = 64 rows (four MFMA k-steps); operand LDS 32 KB single-buffered.
// ------------------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int MRH = 64;  // m rows per stage in bf16 mode

__global__ __launch_bounds__(256) void conv_wgrad_bf16_kernel(const WgP p, const float* __restrict__ x, const float* __restrict__ gy,
                                                               float* __restrict__ dw) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    abr::prof_stamp_begin(p.prof_ts);
    unsigned* Gs = reinterpret_cast<unsigned*>(smem);   // [MRH/2][TN_] (bf16 m, bf16 m+1)
    unsigned* As = Gs + (MRH / 2) * TN_;                // [MRH/2][TK_]

    const unsigned bid = abr::xcd_remap(blockIdx.x, gridDim.x);
    const int total_tiles = p.tiles_n * p.tiles_k;   // tile-fast order: see conv_wgrad_kernel
    const int split = p.tile_fast ? bid / total_tiles : bid % p.splits;
    const int tile = p.tile_fast ? bid % total_tiles : bid / p.splits;
    const int gtile = tile;
    const int tile_n = tile % p.tiles_n, tile_k = tile / p.tiles_n;
    const int n0 = tile_n * TN_, k0 = tile_k * TK_;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // staging: q = 16 B column group, rp = row pair; pairs rp + 8*i (i < 4) -> rows 2*(rp + 8i), +1
    const int q = tid & 31, rp = tid >> 5;
    const int gn = n0 + q * 4;
    const bool g_ok = gn < p.Cout;
    const int ak = k0 + q * 4;
    const bool k_ok = ak < p.K;
    int fr = 0, fs = 0, fc = 0;
    if (k_ok) {
        const int rs = ak / p.Cin;
        fc = ak % p.Cin;
        fr = rs / p.S;
        fs = rs % p.S;
    }
    const int mt0 = split * p.mt_per_split;
    const int mt1 = min(mt0 + p.mt_per_split, (p.M + MRH - 1) / MRH);

    constexpr unsigned kOOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rgy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gy), 0, p.gy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rxx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, p.x_bytes, 0x00020000);
    const unsigned g_voff = g_ok ? (unsigned)(2 * rp * p.Cout + gn) * 4u : kOOB;
    const unsigned a_voff_plain = k_ok ? (unsigned)(2 * rp * p.Cin + fc) * 4u : kOOB;
    const int a_const = ((fr - p.pad) * p.W + (fs - p.pad)) * p.Cin + fc;
    u32x4 rg[4][2], ra[4][2];
    auto load_tile = [&](int mt) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const int mrow = mt * MRH + 16 * i + e;  // scalar part of the row index (the lane adds 2*rp)
                rg[i][e] = __builtin_amdgcn_raw_buffer_load_b128(rgy, (int)(g_voff + (unsigned)(mrow * p.Cout) * 4u), 0, 0);
                if (p.plain) {
                    ra[i][e] = __builtin_amdgcn_raw_buffer_load_b128(rxx, (int)(a_voff_plain + (unsigned)(mrow * p.Cin) * 4u), 0, 0);
                } else {
                    const int m = mrow + 2 * rp;
                    unsigned b, rem, ho, wo;
                    p.d_howo.divmod((unsigned)m, b, rem);
                    p.d_wo.divmod(rem, ho, wo);
                    const int hi = (int)ho * p.stride - p.pad + fr, wi = (int)wo * p.stride - p.pad + fs;
                    const bool ok = k_ok & (m < p.M) & ((unsigned)hi < (unsigned)p.H) & ((unsigned)wi < (unsigned)p.W);
                    const int off = (((int)b * p.H + (int)ho * p.stride) * p.W + (int)wo * p.stride) * p.Cin + a_const;
                    ra[i][e] = __builtin_amdgcn_raw_buffer_load_b128(rxx, (int)(ok ? (unsigned)off * 4u : kOOB), 0, 0);
                }
            }
    };
    auto pack2 = [](unsigned lo, unsigned hi) -> unsigned {   // (row m, row m+1) of one column -> bf16x2, RNE
        const f32x2v f = {__uint_as_float(lo), __uint_as_float(hi)};
        const bf16x2 h = __builtin_convertvector(f, bf16x2);
        return *reinterpret_cast<const unsigned*>(&h);
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            u32x4 g, a;
            g.x = pack2(rg[i][0].x, rg[i][1].x); g.y = pack2(rg[i][0].y, rg[i][1].y);
            g.z = pack2(rg[i][0].z, rg[i][1].z); g.w = pack2(rg[i][0].w, rg[i][1].w);
            a.x = pack2(ra[i][0].x, ra[i][1].x); a.y = pack2(ra[i][0].y, ra[i][1].y);
            a.z = pack2(ra[i][0].z, ra[i][1].z); a.w = pack2(ra[i][0].w, ra[i][1].w);
            *reinterpret_cast<u32x4*>(Gs + (rp + 8 * i) * TN_ + q * 4) = g;
            *reinterpret_cast<u32x4*>(As + (rp + 8 * i) * TK_ + q * 4) = a;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    const int l31 = lane & 31, lh = lane >> 5;
    auto compute_tile = [&]() {
        const unsigned* g = Gs + wm * 64 + 2 * l31 + 4 * lh * TN_;
        const unsigned* a = As + wn * 64 + 2 * l31 + 4 * lh * TK_;
#pragma unroll
        for (int s = 0; s < MRH / 16; s++) {
            uint2 fg[4], fa[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                fg[t] = *reinterpret_cast<const uint2*>(g + (8 * s + t) * TN_);
                fa[t] = *reinterpret_cast<const uint2*>(a + (8 * s + t) * TK_);
            }
            const u32x4 g0 = {fg[0].x, fg[1].x, fg[2].x, fg[3].x}, g1 = {fg[0].y, fg[1].y, fg[2].y, fg[3].y};
            const u32x4 a0 = {fa[0].x, fa[1].x, fa[2].x, fa[3].x}, a1 = {fa[0].y, fa[1].y, fa[2].y, fa[3].y};
            const bf16x8 G0 = *reinterpret_cast<const bf16x8*>(&g0), G1 = *reinterpret_cast<const bf16x8*>(&g1);
            const bf16x8 A0 = *reinterpret_cast<const bf16x8*>(&a0), A1 = *reinterpret_cast<const bf16x8*>(&a1);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(G0, A0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(G0, A1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(G1, A0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(G1, A1, acc[1][1], 0, 0, 0);
        }
    };
    if (mt0 < mt1) {
        load_tile(mt0);
        store_tile();
        __syncthreads();
        int mt = mt0;
        for (; mt + 1 < mt1; mt++) {
            load_tile(mt + 1);
            __builtin_amdgcn_sched_barrier(0);
            compute_tile();
            __syncthreads();
            store_tile();
            __syncthreads();
        }
        compute_tile();
    }

    // epilogue: same interleave as the fp32 kernel -- tile (tm,tn) element (row i, col j) is dW[n0 + wm*64 + 2i + tm][k0 + wn*64 + 2j + tn]
    wgrad_finish(p, acc, n0, k0, wm, wn, l31, lh, tid, gtile, split, dw);
    abr::prof_stamp_end(p.prof_ts);
}

// ------------------------------------------------------------------------------------------------------------------------
// fp32-accurate weight gradient on the bf16 matrix cores (ABR_MATH_BF16X6, see conv_igemm.hip): each operand value is split
// exactly into three bf16 terms, the six cross products with i + j <= 2 are accumulated in fp32.  Stage = 32 rows of m.
//
// LDS image (48 KB: 2 operands x 3 planes x 8 KB): the MFMA wants, per lane, eight bf16 that are CONSECUTIVE IN m for one column, while both
// operands arrive m-major.  The transpose is done by the loader, in registers, for free: a thread fetches the SAME four columns of eight
// consecutive rows (eight 16 B loads; a half-wave still covers a full 512 B row segment per load), so after the split it owns, per plane and
// column, the eight m-values of one fragment half = one 16 B chunk = one ds_write_b128.  Chunks are laid out [octet of m][j][q] with column
// c = 4 q + j (q = the thread's column group): a store instruction's lanes write consecutive chunks, and a wave's fragment read (lanes =
// q 0..31 of ONE j) is a conflict-free ds_read_b128 whose four dwords ARE the MFMA operand -- the former [m-pair][column] image needed four
// ds_read_b64 and eight v_mov per fragment (96 v_mov per 48 MFMAs; the compiler fused the reads into half-rate ds_read2st64_b64).
// The price is the accumulator interleave: sub-tile (tm, tn) of wave (wm, wn), element (i, l) is dW[n0 + 4 i + 2 wm + tm][k0 + 4 l + 2 wn + tn]
// (WgP::map4; undone by the epilogue / reduction addressing like the 2 i + tm interleave of the other two kernels).
// Waves 0-1 stage gy, waves 2-3 stage x (wave-uniform branch): 8 loads, 32 values to split, 12 ds_write_b128 per thread and stage.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int MRX = 32;

// NP = 6: the bf16x6 arithmetic.  NP = 1: ABR_MATH_BF16 (operands rounded to bf16, one product) on the same loop: one plane per operand in LDS.
// NP = 3: ABR_MATH_F16X3 (two fp16 planes per operand, scaled by the operand's amax word; three products on v_mfma_f32_32x32x16_f16).
template <int NP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void conv_wgrad_x6_kernel(const WgP p, const float* __restrict__ x_, const float* __restrict__ gy_,
                                                             float* __restrict__ dw_) {
    const float* x = x_;
    const float* gy = gy_;
    float* dw = dw_;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    abr::prof_stamp_begin(p.prof_ts);
    constexpr int PL = (MRX / 8) * 128 * 4;              // dwords per plane: [4 octets][4 j][32 q] chunks of 4 dwords
    constexpr bool H3 = NP == 3;
    constexpr int NPL = NP == 1 ? 1 : (H3 ? 2 : 3);      // operand planes in use
    unsigned* Gs = reinterpret_cast<unsigned*>(smem);    // [NPL][PL]
    unsigned* As = Gs + NPL * PL;                        // [NPL][PL]

    const unsigned bid = abr::xcd_remap(blockIdx.x, gridDim.x);
    // workgroup -> (row slice, output tile): the output tile is the FAST index, so the workgroups one XCD runs side by side (a
    // contiguous bid range, abr::xcd_remap) cover ALL output tiles of a few row slices.  They stream the same rows of gy and x in
    // lockstep: every operand element is then fetched from HBM once per XCD and served to the other tiles from that XCD's L2,
    // instead of being re-fetched by each of the N/128 (resp. K/128) tiles that need it (slice-fast order: ~3x the HBM traffic on the
    // 9576 x {1024 x 256} gradients, which made them bandwidth-bound).  ABR_WGRAD_TILE_FAST=0 restores the old order.
    const int total_tiles = p.tiles_pb * (p.nbatch > 1 ? p.nbatch : 1);
    const int split = p.tile_fast ? bid / total_tiles : bid % p.splits;
    int tile = p.tile_fast ? bid % total_tiles : bid / p.splits;
    const int gtile = tile;
    if (p.nbatch > 1) {
        const int bt = tile / p.tiles_pb;
        tile -= bt * p.tiles_pb;
        x += bt * p.x_bs; gy += bt * p.gy_bs; dw += bt * p.dw_bs;
    }
    const int tile_n = tile % p.tiles_n, tile_k = tile / p.tiles_n;
    const int n0 = tile_n * TN_, k0 = tile_k * TK_;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const bool is_x = __builtin_amdgcn_readfirstlane(wave) >= 2;   // which operand this wave stages (in an SGPR: the resource descriptor and the branches on it stay scalar)
    const int q = tid & 31, oct = (tid >> 5) & 3; // 16 B column group; rows oct*8 .. +8 of the stage
    const int mt0 = split * p.mt_per_split;
    const int mt1 = min(mt0 + p.mt_per_split, (p.M + MRX - 1) / MRX);

    constexpr unsigned kOOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rsrc = is_x ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, p.x_bytes, 0x00020000)
                                             : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gy), 0, p.gy_bytes, 0x00020000);
    // gy / plain x: row m of the operand starts at m * ld floats; column group at col
    const int col = (is_x ? k0 : n0) + q * 4;
    const int ld = is_x ? p.Cin : p.Cout;
    const bool col_ok = col < (is_x ? p.K : p.Cout);
    int fr = 0, fs = 0, fc = col;
    if (is_x && col_ok && !p.plain) {
        const int rs = col / p.Cin;
        fc = col % p.Cin;
        fr = rs / p.S;
        fs = rs % p.S;
    }
    const bool gather = is_x && !p.plain;         // wave-uniform
    const unsigned voff_plain = col_ok ? (unsigned)(oct * 8 * ld + fc) * 4u : kOOB;
    const int a_const = ((fr - p.pad) * p.W + (fs - p.pad)) * p.Cin + fc;
    u32x4 rr[8];
    auto load_tile = [&](int mt) {
        if (!gather) {
            const unsigned base = voff_plain + (unsigned)(mt * MRX * ld) * 4u;
#pragma unroll
            for (int i = 0; i < 8; i++) rr[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(base + (unsigned)(i * ld) * 4u), 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int m = mt * MRX + oct * 8 + i;
                unsigned b, rem, ho, wo;
                p.d_howo.divmod((unsigned)m, b, rem);
                p.d_wo.divmod(rem, ho, wo);
                const int hi = (int)ho * p.stride - p.pad + fr, wi = (int)wo * p.stride - p.pad + fs;
                const bool ok = col_ok & (m < p.M) & ((unsigned)hi < (unsigned)p.H) & ((unsigned)wi < (unsigned)p.W);
                const int off = (((int)b * p.H + (int)ho * p.stride) * p.W + (int)wo * p.stride) * p.Cin + a_const;
                rr[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(ok ? (unsigned)off * 4u : kOOB), 0, 0);
            }
        }
    };
    // range guard of the exact split (abr_x6_range_flags): gy is inspected by the workgroups of the first k-tile column, x by those of
    // the first n-tile row -- every operand element once per GEMM, wave-uniform branch (see conv_igemm_x6_kernel)
    const bool chk = p.x6_flags && (is_x ? tile_n == 0 : tile_k == 0);
    unsigned bmin = 0xFFFFFFFFu;
    float nonfin = 0.f;
    auto inspect = [&](const u32x4 v) {
        const unsigned b0 = (v.x << 1) - 1u, b1 = (v.y << 1) - 1u, b2 = (v.z << 1) - 1u, b3 = (v.w << 1) - 1u;
        bmin = min(min(bmin, min(b0, b1)), min(b2, b3));
        nonfin = fmaf(__uint_as_float(v.x), 0.f, nonfin); nonfin = fmaf(__uint_as_float(v.y), 0.f, nonfin);
        nonfin = fmaf(__uint_as_float(v.z), 0.f, nonfin); nonfin = fmaf(__uint_as_float(v.w), 0.f, nonfin);
    };
    unsigned* const st_base = (is_x ? As : Gs) + (oct * 4 * 32 + q) * 4;   // chunk (oct, j, q) at + j * 128 dwords
    unsigned op_bits = 0;            // f16x3: this wave's operand's amax and the scale of its split
    float op_s = 1.f, op_inv = 1.f;
    if constexpr (H3) {
        op_bits = is_x ? abr::h3_amax_load(p.x_amax, p.x_epoch) : abr::h3_amax_load(p.gy_amax, p.gy_epoch);
        abr::h3_scales(op_bits, op_s, op_inv);
    }
    const unsigned small_thr = H3 ? abr::h3_small_threshold(op_bits) : 0u;
    unsigned nsmall = 0;
    auto store_tile = [&]() {
        if constexpr (H3) {
            if (chk) {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const u32x4 v = rr[i];
                    nsmall += (unsigned)(((v.x << 1) - 1u) < small_thr) + (unsigned)(((v.y << 1) - 1u) < small_thr) + (unsigned)(((v.z << 1) - 1u) < small_thr) +
                              (unsigned)(((v.w << 1) - 1u) < small_thr);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                u32x4 o0, o1;
#pragma unroll
                for (int pp = 0; pp < 4; pp++) {
                    unsigned a0, a1;
                    abr::h3_split2(__uint_as_float(rr[2 * pp][j]), __uint_as_float(rr[2 * pp + 1][j]), op_inv, a0, a1);
                    o0[pp] = a0; o1[pp] = a1;
                }
                *reinterpret_cast<u32x4*>(st_base + j * 128) = o0;
                *reinterpret_cast<u32x4*>(st_base + j * 128 + PL) = o1;
            }
            return;
        }
        if (chk) {
#pragma unroll
            for (int i = 0; i < 8; i++) inspect(rr[i]);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {         // column 4 q + j: eight m-values -> one 16 B chunk per plane
            u32x4 o0, o1, o2;
#pragma unroll
            for (int pp = 0; pp < 4; pp++) {  // rows (2 pp, 2 pp + 1) -> one bf16x2 dword per plane
                const f32x2v f = {__uint_as_float(rr[2 * pp][j]), __uint_as_float(rr[2 * pp + 1][j])};
                const bf16x2 h0 = __builtin_convertvector(f, bf16x2);
                const f32x2v h0f = __builtin_convertvector(h0, f32x2v);
                const f32x2v r1 = {abr::x6_sub(f.x, h0f.x), abr::x6_sub(f.y, h0f.y)};   // (not v_pk_add_f32: see conv_igemm.hip::x6_split4)
                const bf16x2 h1 = __builtin_convertvector(r1, bf16x2);
                const f32x2v h1f = __builtin_convertvector(h1, f32x2v);
                const f32x2v r2 = {abr::x6_sub(r1.x, h1f.x), abr::x6_sub(r1.y, h1f.y)};
                const bf16x2 h2 = __builtin_convertvector(r2, bf16x2);
                o0[pp] = *reinterpret_cast<const unsigned*>(&h0);
                o1[pp] = *reinterpret_cast<const unsigned*>(&h1);
                o2[pp] = *reinterpret_cast<const unsigned*>(&h2);
            }
            *reinterpret_cast<u32x4*>(st_base + j * 128) = o0;
            if constexpr (NP != 1) {
                *reinterpret_cast<u32x4*>(st_base + j * 128 + PL) = o1;
                *reinterpret_cast<u32x4*>(st_base + j * 128 + 2 * PL) = o2;
            }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    const int l31 = lane & 31, lh = lane >> 5;
    // fragment (s, t) of an operand: chunk (octet 2 s + lh, j = 2 w + t, q = l31)
    const unsigned* const g_rd = Gs + ((lh * 4 + 2 * wm) * 32 + l31) * 4;
    const unsigned* const a_rd = As + ((lh * 4 + 2 * wn) * 32 + l31) * 4;
    auto compute_tile = [&]() {
#pragma unroll
        for (int s = 0; s < MRX / 16; s++) {
            bf16x8 G[2][NPL], A[2][NPL];   // [column sub-tile][plane]
#pragma unroll
            for (int pl = 0; pl < NPL; pl++)
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    G[t][pl] = *reinterpret_cast<const bf16x8*>(g_rd + pl * PL + (s * 8 + t) * 128);
                    A[t][pl] = *reinterpret_cast<const bf16x8*>(a_rd + pl * PL + (s * 8 + t) * 128);
                }
            // the six products of a step (smallest terms first for every accumulator) interleaved over the four accumulators: no MFMA
            // waits on the result of the one issued just before it (as in conv_igemm_x6w_kernel)
            constexpr int pg[6] = {NP == 1 ? 0 : (H3 ? 1 : 2), 0, H3 ? 0 : 1, 1, 0, 0}, pa[6] = {0, NP == 1 ? 0 : (H3 ? 1 : 2), H3 ? 0 : 1, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < NP; t++)
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        if constexpr (H3) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, G[i][pg[t]]), __builtin_bit_cast(f16x8, A[j][pa[t]]), acc[i][j], 0, 0, 0);
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(G[i][pg[t]], A[j][pa[t]], acc[i][j], 0, 0, 0);
                    }
        }
    };
    if (mt0 < mt1) {
        load_tile(mt0);
        store_tile();
        __syncthreads();
        int mt = mt0;
        for (; mt + 1 < mt1; mt++) {
            load_tile(mt + 1);
            __builtin_amdgcn_sched_barrier(0);
            compute_tile();
            __syncthreads();
            store_tile();
            __syncthreads();
        }
        compute_tile();
    }
    if constexpr (H3) {
        if (chk) abr::h3_report(op_bits, nsmall, p.x6_flags, p.h3_stats);
    } else {
        if (chk) abr::x6_report(bmin, nonfin, p.x6_flags);
    }

    wgrad_finish(p, acc, n0, k0, wm, wn, l31, lh, tid, gtile, split, dw);
    abr::prof_stamp_end(p.prof_ts);
}

}  // namespace

// ABR_WGRAD_REDUCE=0: fp32 atomics into dw for the split-M partial sums (the round-1 scheme) instead of parked partials + reduction kernel
static bool wgrad_ticket_enabled() {
    static const bool on = !(getenv("ABR_WGRAD_REDUCE") && atoi(getenv("ABR_WGRAD_REDUCE")) == 0);
    return on;
}

// scratch of the partial-sum reduction: `units` partial tiles of 64 KB, grow-only, one buffer per stream (launches on a stream are ordered)
static float* wgrad_scratch(hipStream_t st, size_t units) {
    struct Ws { float* ws = nullptr; size_t units = 0; };
    static std::map<hipStream_t, Ws> pool;
    static std::mutex mu;   // host threads may issue weight gradients for different streams concurrently
    std::lock_guard<std::mutex> g(mu);
    Ws& w = pool[st];
    if (w.units < units) {
        if (w.ws) { (void)hipStreamSynchronize(st); (void)hipFree(w.ws); w.ws = nullptr; w.units = 0; }
        if (hipMalloc(&w.ws, units * 65536) != hipSuccess) return nullptr;
        w.units = units;
    }
    return w.ws;
}

// per-stream ticket counters of the in-kernel reduction (one per output tile; zeroed once, the kernels leave them zero)
static unsigned* wgrad_tickets(hipStream_t st, int tiles) {
    constexpr int kMaxTiles = 8192;
    // OPT-IN (ABR_WGRAD_INKERNEL_REDUCE=1).  Measured in the step (same session, two rounds): 19.38 ms with it against 18.62 with the reduction
    // kernel -- the write-through stores of the 64 KB partial tiles and the last arrival's 16 x splits coherent loads cost more than the 41
    // short launches they replace, which run beside other streams' kernels anyway.  Results are identical either way.
    static const bool on = getenv("ABR_WGRAD_INKERNEL_REDUCE") && atoi(getenv("ABR_WGRAD_INKERNEL_REDUCE")) != 0;
    if (!on || tiles > kMaxTiles) return nullptr;
    static std::map<hipStream_t, unsigned*> pool;
    static std::mutex mu;
    std::lock_guard<std::mutex> g(mu);
    unsigned*& t = pool[st];
    if (!t) {
        if (hipMalloc(&t, kMaxTiles * sizeof(unsigned)) != hipSuccess) { t = nullptr; return nullptr; }
        (void)hipMemset(t, 0, kMaxTiles * sizeof(unsigned));
    }
    return t;
}

// fills p.ws for a split launch (left null -> the kernel falls back to atomics; the caller must then have zeroed a final_store dw)
static void wgrad_plan_reduction(WgP& p, int tiles, hipStream_t st) {
    p.ws = nullptr;
    p.tickets = nullptr;
    p.ws_bytes = 0;
    if (p.splits <= 1) { if (p.final_store) p.overwrite = 1; return; }
    if (wgrad_ticket_enabled()) p.ws = wgrad_scratch(st, (size_t)tiles * p.splits);
    if (p.ws && (size_t)tiles * p.splits * 65536 < (size_t)0x7FFFFFF0) {
        p.tickets = wgrad_tickets(st, tiles);
        p.ws_bytes = (unsigned)((size_t)tiles * p.splits * 65536);
    }
}

static void wgrad_reduce(const WgP& p, int tiles, float* dw, hipStream_t st) {
    if (p.ws && !p.tickets) wgrad_reduce_kernel<<<(unsigned)tiles * 16u, 256, 0, st>>>(p, dw);
}

// split choice + launch for one (possibly batched) weight-gradient GEMM described by p (tiles_n / tiles_k / M / K filled in)
static void launch_wgrad(WgP p, const float* x, const float* gy, float* dw, void* stream) {
    const int nb = p.nbatch > 1 ? p.nbatch : 1;
    p.tiles_pb = p.tiles_n * p.tiles_k;
    const int m_tiles = (p.M + MR - 1) / MR;
    int32_t info[3];
    const int cus = abr_device_info(info) == ABR_OK ? info[0] : 256;
    const int tiles = p.tiles_n * p.tiles_k * nb;
    // split-M so that the grid fills the chip evenly: among the candidates pick the one with the best load balance
    // (workgroups / (ceil(workgroups / CUs) * CUs)), preferring fewer splits (less atomic traffic) on ties; every
    // workgroup keeps at least 8 stages (256 rows) of work.
    // Prefer two workgroups per CU when each still gets >= 64 stages: a lone workgroup cannot hide its own prologue / atomics
    // epilogue (head 1x1 gradients: 113 -> 126..134 TF alone; -1.0 ms per training step).  ABR_WGRAD_OCC2=0 turns it off.
    // Round 2 had this off (next to the heavier dgrad kernels of that round two concurrent 512-workgroup grids oversubscribed the CUs: +0.13 ms);
    // with the weights-direct dgrad kernels (35 KB of LDS, 3 waves / SIMD) it is worth -0.25 ms per step in four same-session A/B rounds and
    // lifts the kernel's own rate 132 -> 148 TF-eq, so it is ON by default since round 3.
    static const bool occ2 = !(getenv("ABR_WGRAD_OCC2") && atoi(getenv("ABR_WGRAD_OCC2")) == 0);
    static const int occ2_min_stages = getenv("ABR_WGRAD_OCC2_MINSTAGES") ? atoi(getenv("ABR_WGRAD_OCC2_MINSTAGES")) : 64;
    const int max_splits = std::max(1, (m_tiles + 7) / 8);
    int splits = 1;
    double best = -1.0;
    for (int sp = 1; sp <= max_splits; sp++) {
        const long wgs = (long)tiles * sp;
        if (wgs > 8L * cus && sp > 1) break;
        const long rounds = (wgs + cus - 1) / cus;
        double eff = (double)wgs / (double)(rounds * cus);
        if (wgs < cus) eff *= 0.5;                 // not even one workgroup per CU
        else if (occ2 && wgs < 2L * cus && m_tiles / (2 * sp) >= occ2_min_stages) eff *= 0.85;  // a lone workgroup per CU cannot hide its own prologue /
                                                   // atomics epilogue: take two when each still gets >= 64 stages (head 1x1s: +11..15 %)
        eff -= 0.0002 * sp;                        // tie-break: fewer partial sums
        if (eff > best) { best = eff; splits = sp; }
    }
    if (p.overwrite) splits = 1;  // one workgroup per output tile owns it: no zero-fill, no atomics
    p.splits = splits;
    p.mt_per_split = (m_tiles + splits - 1) / splits;
    wgrad_plan_reduction(p, tiles, abr::as_stream(stream));
    if (p.final_store && splits > 1 && !p.ws) {   // no scratch: atomics into a zeroed dw after all
        (void)hipMemsetAsync(dw, 0, sizeof(float) * (size_t)nb * p.Cout * p.K, abr::as_stream(stream));
        p.final_store = 0;
    }
    // single-buffered by default: every shape of the step is as fast or faster with three resident workgroups per CU (RPN 3x3
    // 101 -> 109 TF, layer2 3x3 50 -> 62 TF, layer4 +2..3 %); ABR_WGRAD_SB=0 selects the double-buffered variant for comparison
    static const int sb_mode = getenv("ABR_WGRAD_SB") ? atoi(getenv("ABR_WGRAD_SB")) : 1;
    const bool sb = sb_mode != 0;
    const size_t lds = sizeof(float) * (sb ? 1 : 2) * MR * (TN_ + TK_);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(sizeof(float) * 2 * MR * (TN_ + TK_)));
        attr_set = true;
    }
    if (p.math == ABR_MATH_BF16X6 || p.math == ABR_MATH_BF16 || p.math == ABR_MATH_F16X3) {   // same split plan (MRX == MR), three-plane (bf16: one-plane, f16x3: two-plane) LDS
        static bool attr6 = false;
        const bool one = p.math == ABR_MATH_BF16;   // round 4: the bf16 mode on the bf16x6 kernel's loader / LDS image, single product
        const bool h3 = p.math == ABR_MATH_F16X3;
        const size_t lds6 = sizeof(unsigned) * (one ? 1 : (h3 ? 2 : 3)) * (MRX / 2) * (TN_ + TK_);
        if (!attr6) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_x6_kernel<6>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)(sizeof(unsigned) * 3 * (MRX / 2) * (TN_ + TK_)));
            attr6 = true;
        }
        p.map4 = 1;
        p.x6_flags = (!one && abr::x6_guard_enabled()) ? abr::x6_flags_ptr() : nullptr;
        p.h3_stats = (h3 && p.x6_flags) ? abr::h3_stats_ptr() : nullptr;
        if (p.h3_stats) abr::h3_stats_inspected((double)nb * p.M * ((double)p.Cout + (double)p.K));   // gy by the first k-tile column, x by the first n-tile row
        // Timed with a HIP-event pair, not with in-kernel stamps: a kernel trace's duration of these kernels includes the write-back of the
        // parked partial tiles at kernel end, which first-workgroup-in / last-workgroup-out stamps miss by ~10 % (the raw event figure is
        // within 3 % of rocprofv3's here; for the forward / dgrad kernels it is the stamps that agree, within 2.5 %).
        const int pid6 = h3 ? abr::PROF_WGRAD_H3 : abr::PROF_WGRAD_BF16;
        const int rec6 = abr::prof_start(abr::as_stream(stream), pid6, 2.0 * (double)p.M * (double)p.Cout * (double)p.K * nb);
        p.prof_ts = abr::prof_clock_slot(rec6);
        // algorithmic bytes: x and gy once (x at its own size for a direct 3x3), dw written once
        abr::prof_add_bytes(pid6, 4.0 * nb * ((double)p.M * p.Cout + (p.plain ? (double)p.M * p.K : (double)p.B * p.H * p.W * p.Cin) + (double)p.Cout * p.K));
        if (h3) conv_wgrad_x6_kernel<3><<<(unsigned)(tiles * splits), 256, lds6, abr::as_stream(stream)>>>(p, x, gy, dw);
        else if (one) conv_wgrad_x6_kernel<1><<<(unsigned)(tiles * splits), 256, lds6, abr::as_stream(stream)>>>(p, x, gy, dw);
        else conv_wgrad_x6_kernel<6><<<(unsigned)(tiles * splits), 256, lds6, abr::as_stream(stream)>>>(p, x, gy, dw);
        abr::prof_stop(abr::as_stream(stream), rec6);
        wgrad_reduce(p, tiles, dw, abr::as_stream(stream));
        return;
    }
    const int rec = abr::prof_start(abr::as_stream(stream), abr::PROF_WGRAD, 2.0 * (double)p.M * (double)p.Cout * (double)p.K * nb);
    p.prof_ts = abr::prof_clock_slot(rec);
    if (sb) conv_wgrad_kernel<true><<<(unsigned)(tiles * splits), 256, lds, abr::as_stream(stream)>>>(p, x, gy, dw);
    else conv_wgrad_kernel<false><<<(unsigned)(tiles * splits), 256, lds, abr::as_stream(stream)>>>(p, x, gy, dw);
    abr::prof_stop(abr::as_stream(stream), rec);
    wgrad_reduce(p, tiles, dw, abr::as_stream(stream));
}


// bf16 math mode: the same split-M choice, in 64-row stages
static void launch_wgrad_bf16(WgP p, const float* x, const float* gy, float* dw, void* stream) {
    const int m_tiles = (p.M + MRH - 1) / MRH;
    int32_t info[3];
    const int cus = abr_device_info(info) == ABR_OK ? info[0] : 256;
    const int tiles = p.tiles_n * p.tiles_k;
    const int max_splits = std::max(1, (m_tiles + 3) / 4);
    int splits = 1;
    double best = -1.0;
    for (int sp = 1; sp <= max_splits; sp++) {
        const long wgs = (long)tiles * sp;
        if (wgs > 8L * cus && sp > 1) break;
        const long rounds = (wgs + cus - 1) / cus;
        double eff = (double)wgs / (double)(rounds * cus);
        if (wgs < 2L * cus) eff *= 0.5 + 0.25 * (double)wgs / (double)(2L * cus);   // latency-bound kernel: wants >= 2 workgroups per CU
        eff -= 0.0002 * sp;
        if (eff > best) { best = eff; splits = sp; }
    }
    p.splits = splits;
    p.mt_per_split = (m_tiles + splits - 1) / splits;
    wgrad_plan_reduction(p, tiles, abr::as_stream(stream));
    const size_t lds = sizeof(unsigned) * (MRH / 2) * (TN_ + TK_);
    const int rec = abr::prof_start(abr::as_stream(stream), abr::PROF_WGRAD_BF16, 2.0 * (double)p.M * (double)p.Cout * (double)p.K);
    p.prof_ts = abr::prof_clock_slot(rec);
    conv_wgrad_bf16_kernel<<<(unsigned)(tiles * splits), 256, lds, abr::as_stream(stream)>>>(p, x, gy, dw);
    abr::prof_stop(abr::as_stream(stream), rec);
    wgrad_reduce(p, tiles, dw, abr::as_stream(stream));
}

extern "C" int abr_conv_wgrad(const abr_conv_desc* d, const float* x, const float* gy, float* dw, void* stream) {
    ABR_REQUIRE(d && x && gy && dw, "conv_wgrad: null pointer");
    ABR_REQUIRE(d->Cin % 4 == 0 && d->Cout % 4 == 0, "conv_wgrad: Cin and Cout must be multiples of 4");
    ABR_REQUIRE(d->Ho == (d->H + 2 * d->pad - d->R) / d->stride + 1 && d->Wo == (d->W + 2 * d->pad - d->S) / d->stride + 1,
                "conv_wgrad: Ho/Wo inconsistent");
    WgP p;
    p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout; p.R = d->R; p.S = d->S;
    p.stride = d->stride; p.pad = d->pad; p.Ho = d->Ho; p.Wo = d->Wo;
    p.M = d->B * d->Ho * d->Wo;
    p.K = d->R * d->S * d->Cin;
    p.plain = (d->R == 1 && d->S == 1 && d->stride == 1 && d->pad == 0);
    p.scale = d->scale;
    const int64_t xb = (int64_t)d->B * d->H * d->W * d->Cin * 4, gb = (int64_t)p.M * d->Cout * 4;
    ABR_REQUIRE(xb < (int64_t)0x7FFFFFF0 && gb < (int64_t)0x7FFFFFF0, "conv_wgrad: activation / gradient tensors must be < 2 GB (32-bit buffer offsets)");
    p.x_bytes = (unsigned)xb; p.gy_bytes = (unsigned)gb;
    p.d_howo.init((unsigned)(d->Ho * d->Wo)); p.d_wo.init((unsigned)d->Wo);
    p.tiles_n = (p.Cout + TN_ - 1) / TN_;
    p.tiles_k = (p.K + TK_ - 1) / TK_;
    if (p.M == 0) return ABR_OK;
    p.nbatch = 1; p.tiles_pb = 0; p.x_bs = p.gy_bs = p.dw_bs = 0; p.overwrite = 0;
    p.ws = nullptr; p.final_store = 0; p.map4 = 0;
    p.x6_flags = nullptr;
    p.gy_amax = p.x_amax = nullptr; p.gy_epoch = p.x_epoch = 0; p.h3_stats = nullptr;
    p.tickets = nullptr; p.ws_bytes = 0;
    static const int tile_fast = !(getenv("ABR_WGRAD_TILE_FAST") && atoi(getenv("ABR_WGRAD_TILE_FAST")) == 0);
    p.tile_fast = tile_fast;
    p.math = (d->math == ABR_MATH_BF16X6 || d->math == ABR_MATH_F16X3) ? d->math : ABR_MATH_F32;   // (x6 / h3 handle any Cin % 4 == 0: no k-tile constraint here)
    static const bool bf16_on_x6 = !(getenv("ABR_BF16_WEIGHTS_DIRECT") && atoi(getenv("ABR_BF16_WEIGHTS_DIRECT")) == 0);
    hipStream_t st = abr::as_stream(stream);
    ABR_REQUIRE(d->math == ABR_MATH_F32 || d->math == ABR_MATH_BF16 || d->math == ABR_MATH_BF16X6 || d->math == ABR_MATH_F16X3, "conv_wgrad: unknown math mode");
    const bool h3 = p.math == ABR_MATH_F16X3;
    // f16x3: amax words of the two operands -- the caller's (written by the producers of x / gy), else reduced here
    abr::AmaxRef gy_ref{reinterpret_cast<unsigned long long*>(const_cast<uint64_t*>(d->gy_amax)), d->gy_amax_epoch};
    abr::AmaxRef x_ref{reinterpret_cast<unsigned long long*>(const_cast<uint64_t*>(d->x_amax)), d->x_amax_epoch};
    auto need_ref = [&](abr::AmaxRef& r, const float* t, int64_t n) -> bool {
        if (r.word) return true;
        r = abr::h3_amax_alloc();
        return r.word && abr::h3_amax_reduce(t, n, r, st) == 0;
    };
    if (d->math == ABR_MATH_BF16 && d->Cin % 64 == 0) {   // same layer set as the bf16 forward (the stem stays fp32)
        if (bf16_on_x6) {   // round 4: the loader-transposed LDS image and split plan of the default arithmetic, one product (direct form: no Winograd)
            p.math = ABR_MATH_BF16;
            launch_wgrad(p, x, gy, dw, stream);
            ABR_CHECK_LAUNCH("conv_wgrad (bf16 on the x6 loop)");
            return ABR_OK;
        }
        launch_wgrad_bf16(p, x, gy, dw, stream);
        ABR_CHECK_LAUNCH("conv_wgrad (bf16)");
        return ABR_OK;
    }
    // Winograd F(4x4,3x3) weight gradient for the wide stride-1 3x3 convs: dU[p] = sum_tiles (A dY A^T)[p]^T (B^T d B)[p] as 36 batched
    // GEMMs over the tile axis (4x fewer multiply-adds than the direct form), then dW += scale * G^T dU G.
    static const int wino_min_c = getenv("ABR_WINOGRAD_MIN_C") ? atoi(getenv("ABR_WINOGRAD_MIN_C")) : 128;
    static const bool wino_wgrad = !(getenv("ABR_WINOGRAD_WGRAD") && atoi(getenv("ABR_WINOGRAD_WGRAD")) == 0);
    if (wino_wgrad && wino_min_c > 0 && d->R == 3 && d->S == 3 && d->stride == 1 && d->pad == 1 && d->Cin % 4 == 0 &&
        d->Cin >= wino_min_c && d->Cout >= 128) {
        const int th_n = (d->H + 3) / 4, tw_n = (d->W + 3) / 4;
        const int64_t T = (int64_t)d->B * th_n * tw_n;
        const size_t nV = (size_t)36 * T * d->Cin, nM = (size_t)36 * T * d->Cout, nU = (size_t)36 * d->Cout * d->Cin;
        float* vin = d->wino_v;   // the forward pass kept its transformed input: no second B^T d B over x
        float* ws = T * (int64_t)std::max(d->Cin, d->Cout) * 4 < (int64_t)0x7FFFFFF0 ? abr::wino_ws(st, (vin ? 0 : nV) + nM + nU) : nullptr;
        if (ws) {
            float* V = vin ? vin : ws;
            float *Mg = ws + (vin ? 0 : nV), *dU = Mg + nM;
            // f16x3: the GEMM's operands are V and Mg: their amax words come from the transforms (a kept V: from the forward pass's, else reduced)
            abr::AmaxRef v_ref{nullptr, 0}, m_ref{nullptr, 0};
            if (h3) {
                if (vin) v_ref = abr::h3_amax_recall(vin);
                else v_ref = abr::h3_amax_alloc();
                m_ref = abr::h3_amax_alloc();
                ABR_REQUIRE(m_ref.word && (vin || v_ref.word), "conv_wgrad (f16x3): no amax words");
                if (vin && !v_ref.word) ABR_REQUIRE(need_ref(v_ref, vin, (int64_t)nV), "conv_wgrad (f16x3): amax reduction failed");
            }
            int bad = vin ? 0 : abr::wino_input_transform(x, d->B, d->H, d->W, d->Cin, V, st, h3 ? &v_ref : nullptr);
            bad |= abr::wino_outgrad_transform(gy, d->B, d->H, d->W, d->Cout, Mg, st, h3 ? &m_ref : nullptr);
            // enough output tiles to fill the chip without splitting the tile axis -> each workgroup owns its dU tile and writes
            // it directly; otherwise split-M with atomics into a zeroed dU
            int32_t info[3];
            const int cus_ = abr_device_info(info) == ABR_OK ? info[0] : 256;
            const long out_tiles = 36L * ((d->Cout + TN_ - 1) / TN_) * ((d->Cin + TK_ - 1) / TK_);
            const int own = out_tiles >= 2L * cus_;
            // split-M partial sums of dU go through the reduction kernel, which STORES the tile: no zero-fill of dU
            // (launch_wgrad zero-fills dU itself when it has to fall back to atomics)
            const bool ticket = wgrad_ticket_enabled();
            if (!own && !ticket) bad |= hipMemsetAsync(dU, 0, nU * sizeof(float), st) != hipSuccess;
            if (!bad) {
                WgP g = p;
                g.overwrite = own;
                g.final_store = (!own && ticket) ? 1 : 0;
                g.B = (int)T; g.H = g.W = 1; g.R = g.S = 1; g.stride = 1; g.pad = 0; g.Ho = g.Wo = 1;
                g.M = (int)T; g.K = d->Cin; g.plain = 1; g.scale = nullptr;
                g.d_howo.init(1u); g.d_wo.init(1u);
                g.tiles_n = (g.Cout + TN_ - 1) / TN_; g.tiles_k = (g.K + TK_ - 1) / TK_;
                g.x_bytes = (unsigned)(T * d->Cin * 4); g.gy_bytes = (unsigned)(T * d->Cout * 4);
                g.nbatch = 36; g.x_bs = (long)T * d->Cin; g.gy_bs = (long)T * d->Cout; g.dw_bs = (long)d->Cout * d->Cin;
                if (h3) { g.x_amax = v_ref.word; g.x_epoch = v_ref.epoch; g.gy_amax = m_ref.word; g.gy_epoch = m_ref.epoch; }
                launch_wgrad(g, V, Mg, dU, stream);
                bad = abr::wino_wgrad_inverse(dU, d->Cout, d->Cin, d->scale, dw, st);
            }
            ABR_REQUIRE(!bad, "conv_wgrad: winograd launch failed");
            ABR_CHECK_LAUNCH("conv_wgrad (winograd)");
            return ABR_OK;
        }
    }
    if (h3) {
        ABR_REQUIRE(need_ref(gy_ref, gy, (int64_t)p.M * d->Cout) && need_ref(x_ref, x, (int64_t)d->B * d->H * d->W * d->Cin), "conv_wgrad (f16x3): amax reduction failed");
        p.gy_amax = gy_ref.word; p.gy_epoch = gy_ref.epoch; p.x_amax = x_ref.word; p.x_epoch = x_ref.epoch;
    }
    launch_wgrad(p, x, gy, dw, stream);
    ABR_CHECK_LAUNCH("conv_wgrad");
    return ABR_OK;
}
