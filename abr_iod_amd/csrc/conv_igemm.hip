// Convolution forward / dgrad as implicit GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces, for the reference's hot path, cuDNN conv fwd + dgrad, FrozenBatchNorm2d, ReLU and the residual
// add (modeling/backbone/resnet.py:327-346,363-368; layers/batch_norm.py:19-31; modeling/rpn/rpn.py:114-121;
// roi_box_predictors.py:27-32 as a 1x1 conv on a 1x1 map).
//
//   out[m, n] = epilogue( sum_k A[m, k] * Wt[n, k] )     m = (b, ho, wo), n = cout, k = (r, s, cin)
//
// Layout: activations NHWC, weights OHWI -> BOTH operands are k-contiguous ("TN" GEMM): A rows are gathered
// straight from the NHWC image (zero-filled halo), weight rows are read as stored.  No im2col buffer.
// dgrad is the same kernel run on gy with the flipped/transposed weight copy from abr_conv_dgrad_weights;
// the dgrad of a stride-2 1x1 conv stores its rows at (2ho, 2wo) of a zeroed tensor (out_sh/out_sw).
//
// Tiling (gfx950): 256 threads = 4 waves; block tile BM x BN x 32; each wave owns TM x TN accumulators of
// 32x32 (16 VGPR each).  LDS tiles are [row][36] floats: the +4 pad makes the ds_read_b128 fragment reads
// conflict-free (16-lane groups land on 16 distinct 16 B slots).  A lane reads 4 consecutive k with one
// b128 and feeds them to 4 successive MFMAs; the lane-half h supplies k = 8u+4h+t for both operands, so the
// products match (k order inside a tile is a free permutation).
// Operand fetch: BUFFER loads with 32-bit offsets against range-checked descriptors (halo taps, tail rows and
// the K tail read as zeros in hardware: no branches), register-staged into LDS.  Two loop shapes: double-buffered
// LDS (one barrier per k-tile, tile kt+2 fetched right behind the ds_writes of kt+1; 2 workgroups/CU) for long K,
// single-buffered (two barriers, 3+ workgroups/CU) for K <= 512 and the smaller tiles.  The last k-tile is peeled,
// so the steady-state loop is one basic block.
// Grid shaping: XCD-aware tile order; split-K of the LAST round's tiles for long-K GEMMs whose grid does not fill
// it (partials through system-coherent buffer accesses, deterministic last-arrival reduce); batched mode for the
// 36 GEMMs of a Winograd F(4x4,3x3) convolution (conv_winograd.hip), which every wide stride-1 3x3 conv takes.
// fp32 MFMA is bit-for-bit an fmaf chain, so this kernel's results are exact-fp32.  The DEFAULT arithmetic of the library is the bf16x6
// kernel further down (conv_igemm_x6_kernel: fp32-accurate contractions on the bf16 matrix cores); this one runs with ABR_MATH_F32.
// Bound: MFMA (157.3 TFLOP/s fp32 matrix peak); DESIGN.md has the per-layer flop counts and measured rates.
#include <algorithm>
#include <map>

#include <map>
#include <mutex>
#include <vector>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int LDP = BK + 4;  // LDS row pitch in floats

constexpr int kMaxSplitUnits = 1024, kMaxSplitTiles = 512;  // split-K scratch capacity (units of one 128x128 partial tile)
constexpr int EPAD = 4;  // epilogue LDS pitch = wave-tile columns + 4 floats (keeps 16 B alignment)

// n / d for 0 <= n < 2^31 with a precomputed multiplier (the kernels divide by Ho*Wo, Wo, Cin, S on every tile / row;
// a generic 32-bit division is ~40 VALU instructions, this is 3).
struct FastDiv {
    unsigned mul, shr, d;
    __host__ void init(unsigned div) {
        d = div;
        shr = 0;
        while ((1u << shr) < div) shr++;
        mul = (unsigned)((((unsigned long long)1 << 32) * (((unsigned long long)1 << shr) - div)) / div + 1);
    }
    __device__ __forceinline__ unsigned div(unsigned n) const { return (__umulhi(n, mul) + n) >> shr; }
    __device__ __forceinline__ void divmod(unsigned n, unsigned& q, unsigned& r) const { q = div(n); r = n - q * d; }
};

struct ConvP {
    unsigned long long* prof_ts;   // bench profiling: {first start, last end} stamp slot of this launch (abr::prof_stamp_slot) or nullptr
    int B, H, W, Cin, Cout, R, S, stride, pad, Ho, Wo;
    int M, K;  // GEMM sizes
    int out_H, out_W, out_sh, out_sw;
    int relu, scatter;
    int tiles_m, tiles_n;
    int ngroup;    // weights-direct kernels: > 0 = tiles ordered n-group-major, `ngroup` n-tile columns at a time (see launch_x6w_np); 0 = m-major
    FastDiv d_howo, d_wo, d_cin, d_s;
    const float* scale;
    const float* bias;
    const float* residual;
    const float* mask;
    // split-K for badly quantised grids: tiles [0, n_full) are computed whole, every later tile by `split` workgroups that each take
    // a contiguous K range, park their partial accumulators in `ws` and let the LAST one to arrive (ticket in `cnt`) add them up in
    // fixed order and run the epilogue.
    int n_full, split;
    float* ws;
    int* cnt;
    unsigned x_bytes, w_bytes;  // extents for the buffer-load range check (per batch)
    // batched mode (the 36 Winograd GEMMs): tile -> (batch, tile inside the batch); operands / output advance by the strides
    int nbatch, tiles_pb;
    long a_bs, w_bs, o_bs;
    int math;  // ABR_MATH_*
    float* v_out;  // Winograd path: keep the transformed input here (abr_conv_desc::wino_v)
    const void* w_planes;  // bf16x6: the weights as fragment-packed bf16x3 planes (x6_pack_kernel), or NULL = split w in-kernel
    unsigned wp_bytes;     // bytes of one packed matrix (ceil(Cout/32)*32 * K * 6)
    long wp_bs;            // batched mode: bytes between two packed matrices
    int wp_nblocks;        // 32-row blocks in one packed matrix
    unsigned* x6_flags;      // bf16x6: device word of the range guard (abr::x6_flags_ptr)
    int64_t w_version;       // abr_conv_desc::w_version (0 = nothing derived from w may be cached)
    int nprod;               // 6 = bf16x6 (exact three-way split, six products); 1 = ABR_MATH_BF16 on the same kernels: plane 0 of both operands, one product;
                             // 3 = ABR_MATH_F16X3 (two-term fp16 split, three products: common.h)
    // f16x3: the A operand's amax word (scale of its split), the per-row scales of the packed weight planes (folded into the epilogue's column
    // scale; batched mode: Cout per batch), and -- any arithmetic -- the amax word the epilogue feeds with max |out|
    const unsigned long long* a_amax;
    unsigned a_epoch;
    const float* w_scales;
    unsigned long long* out_amax;
    unsigned out_epoch;
    unsigned long long* h3_stats;   // f16x3 range statistics (abr::h3_stats_ptr) or nullptr
};


// Returns the bits of max |value stored| by this thread (for the output's amax word).  wsc / sa (f16x3): per-column scales of the packed weight
// planes and the A operand's scale -- both powers of two, multiplied into the column scale, so the result equals scaling the accumulators first.
// PF (the weights-direct kernels; round 6): the residual / mask rows of a 32-row block are REQUESTED before the block's accumulators go through LDS, so
// their L2 / HBM latency runs under the transposition instead of in front of every store: the epilogue of a tile with a fused residual or mask was a
// chain of TM x (load -> wait -> store) round trips during which the workgroup holds a third of its CU.  Stand-alone (tools/dbg/igemm_probe.py,
// profiles/r06_igemm_epilogue_prefetch.txt): 36864x2048x512 + residual 306 -> 290 us, + residual + mask 350 -> 333, 9576x1024x256 45 -> 41, 37500x512x128
// 64 -> 58, 150000x256x64 104 -> 91; step 17.65 / 17.60 -> 17.49 / 17.52 ms.  (Two register sets -- block i + 1 requested before block i's stores --
// spill at three waves per SIMD.  A start-up skew of the first 768 workgroups, meant to keep a CU's three workgroups in different phases, LOST 0.1-0.2 ms.)
template <int TM, int TN, bool PF = false>
__device__ __forceinline__ unsigned epilogue_rows(const ConvP& p, f32x16 (&acc)[TM][TN], float* ep, int m_base, int n_base, int lane,
                                                  float* __restrict__ out, const float* __restrict__ wsc = nullptr, const float sa = 1.f) {
    unsigned amax_bits = 0;
    // one 32-row block of the wave tile at a time: the staging slice is 32 x (WC + EPAD) floats per wave (8.7 KB at WC = 64), so
    // the whole workgroup needs 34.8 KB -- it fits the single-buffered operand LDS as well as the double-buffered one.
    constexpr int WC = TN * 32, EP = WC + EPAD;
    const int l31 = lane & 31, lh = lane >> 5;
    constexpr int LPR = WC / 4;    // lanes per row (16 B each)
    constexpr int RPI = 64 / LPR;  // rows per iteration of the wave
    const int c4 = (lane % LPR) * 4;
    const int ncol = n_base + c4;
    const bool vec_ok = (p.Cout % 4 == 0) && (ncol + 3 < p.Cout);
    float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), bi4 = make_float4(0.f, 0.f, 0.f, 0.f), lo4 = make_float4(1.f, 1.f, 1.f, 1.f);
    if (ncol < p.Cout) {
        float* s_ = reinterpret_cast<float*>(&sc4);
        float* l_ = reinterpret_cast<float*>(&lo4);
        float* b_ = reinterpret_cast<float*>(&bi4);
#pragma unroll
        for (int e = 0; e < 4; e++)
            if (ncol + e < p.Cout) {
                if (p.scale) s_[e] = p.scale[ncol + e];
                if (p.bias) b_[e] = p.bias[ncol + e];
                if (wsc) {   // f16x3: two exact power-of-two steps, the smaller factor first -- their PRODUCT, or the accumulator times the
                             // larger one, could leave fp32's range although neither the accumulator nor the result does
                    const float ws = wsc[ncol + e];
                    l_[e] = fminf(sa, ws);
                    s_[e] *= fmaxf(sa, ws);
                }
            }
    }
    typedef float nt4 __attribute__((ext_vector_type(4)));
    constexpr int NIT = 32 / RPI;
    constexpr bool PRE = PF && NIT <= 4;
    auto row_offset = [&](int m) -> size_t {
        if (p.scatter) {
            unsigned b, rem, ho, wo;
            p.d_howo.divmod((unsigned)m, b, rem);
            p.d_wo.divmod(rem, ho, wo);
            return (((size_t)b * p.out_H + (size_t)ho * p.out_sh) * p.out_W + (size_t)wo * p.out_sw) * p.Cout;
        }
        return (size_t)m * p.Cout;
    };
    nt4 rr[1][PRE ? NIT : 1], mk[1][PRE ? NIT : 1];
    auto request = [&](int i, int buf) {   // the residual / mask rows of block i into register set `buf`
        if constexpr (PRE) {
            if (!vec_ok) return;
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const int m = m_base + i * 32 + it * RPI + lane / LPR;
                if (m >= p.M) continue;
                const size_t ro = row_offset(m) + ncol;
                if (p.residual) rr[buf][it] = __builtin_nontemporal_load(reinterpret_cast<const nt4*>(p.residual + ro));
                if (p.mask) mk[buf][it] = __builtin_nontemporal_load(reinterpret_cast<const nt4*>(p.mask + ro));
            }
        }
    };
#pragma unroll
    for (int i = 0; i < TM; i++) {
        if constexpr (PRE) request(i, 0);
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                ep[((r & 3) + 8 * (r >> 2) + 4 * lh) * EP + j * 32 + l31] = acc[i][j][r];
#pragma unroll 4
        for (int it = 0; it < NIT; it++) {
            const int row = it * RPI + lane / LPR;
            const int m = m_base + i * 32 + row;
            if (m >= p.M || ncol >= p.Cout) continue;
            const size_t row_off = row_offset(m);
            float4 v = *reinterpret_cast<const float4*>(ep + row * EP + c4);
            if (wsc) { v.x *= lo4.x; v.y *= lo4.y; v.z *= lo4.z; v.w *= lo4.w; }
            v.x = v.x * sc4.x + bi4.x; v.y = v.y * sc4.y + bi4.y; v.z = v.z * sc4.z + bi4.z; v.w = v.w * sc4.w + bi4.w;
            float* o = out + row_off + ncol;
            if (vec_ok) {
                if (p.residual) {
                    nt4 rr_;
                    if constexpr (PRE) rr_ = rr[0][it];
                    else rr_ = __builtin_nontemporal_load(reinterpret_cast<const nt4*>(p.residual + row_off + ncol));
                    v.x += rr_.x; v.y += rr_.y; v.z += rr_.z; v.w += rr_.w;
                }
                if (p.relu) { v.x = abr::relu_f(v.x); v.y = abr::relu_f(v.y); v.z = abr::relu_f(v.z); v.w = abr::relu_f(v.w); }
                if (p.mask) {
                    nt4 mm;
                    if constexpr (PRE) mm = mk[0][it];
                    else mm = __builtin_nontemporal_load(reinterpret_cast<const nt4*>(p.mask + row_off + ncol));
                    v.x = mm.x > 0.f ? v.x : 0.f; v.y = mm.y > 0.f ? v.y : 0.f; v.z = mm.z > 0.f ? v.z : 0.f; v.w = mm.w > 0.f ? v.w : 0.f;
                }
                *reinterpret_cast<float4*>(o) = v;
                if (p.out_amax)
                    amax_bits = max(max(amax_bits, max(__float_as_uint(v.x) & 0x7FFFFFFFu, __float_as_uint(v.y) & 0x7FFFFFFFu)),
                                    max(__float_as_uint(v.z) & 0x7FFFFFFFu, __float_as_uint(v.w) & 0x7FFFFFFFu));
            } else {
                const float* vv = reinterpret_cast<const float*>(&v);
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    if (ncol + e >= p.Cout) break;
                    float t = vv[e];
                    if (p.residual) t += p.residual[row_off + ncol + e];
                    if (p.relu) t = abr::relu_f(t);
                    if (p.mask) t = p.mask[row_off + ncol + e] > 0.f ? t : 0.f;
                    o[e] = t;
                    amax_bits = max(amax_bits, __float_as_uint(t) & 0x7FFFFFFFu);
                }
            }
        }
    }
    return amax_bits;
}

// SMALL_C: Cin is not a multiple of 32 (the 3->4 padded stem): (r,s,c) is derived per 16 B slot.
// SB: single-buffered operand LDS (two barriers per k-tile, half the LDS -> one more resident workgroup per CU).
template <int BM, int BN, int WM, int WN, bool SMALL_C, bool SB>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvP p, const float* __restrict__ x_,
                                                          const float* __restrict__ w_, float* __restrict__ out_) {
    const float* x = x_;
    const float* w = w_;
    float* out = out_;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int NA = BM / 32, NB = BN / 32;  // float4 staging loads per thread for A and B
    extern __shared__ __attribute__((aligned(16))) float smem[];
    abr::prof_stamp_begin(p.prof_ts);
    constexpr int NBUF = SB ? 1 : 2;
    float* As = smem;                        // [NBUF][BM][LDP]
    float* Bs = smem + NBUF * BM * LDP;      // [NBUF][BN][LDP]

    int tile, unit = -1, kt0 = 0, kt1 = (p.K + BK - 1) / BK;
    if ((int)blockIdx.x < p.n_full) {
        tile = (int)abr::xcd_remap(blockIdx.x, (unsigned)p.n_full);
    } else {  // the split tiles are the LAST workgroups of the grid: they start when the whole tiles have claimed their CUs
        unit = (int)blockIdx.x - p.n_full;
        tile = p.n_full + unit / p.split;
        const int si = unit % p.split, nk_all = kt1;
        kt0 = (int)((long)si * nk_all / p.split);
        kt1 = (int)((long)(si + 1) * nk_all / p.split);
    }
    if (p.nbatch > 1) {
        const int bt = tile / p.tiles_pb;
        tile -= bt * p.tiles_pb;
        x += bt * p.a_bs; w += bt * p.w_bs; out += bt * p.o_bs;
    }
    const int tile_m = tile / p.tiles_n, tile_n = tile % p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // ---- staging assignment: slot = tid + 256*i -> (row = slot/8, kq = slot%8)
    const int kq = tid & 7;
    const int srow = tid >> 3;  // 0..31 ; rows srow + 32*i
    // Operand fetch goes through BUFFER loads: 32-bit byte offsets against a resource descriptor whose range check returns zeros
    // for anything outside [0, bytes) -- halo taps, rows past M / Cout and the K tail are given the offset kOOB and need neither a
    // branch nor an exec mask, and the per-load address arithmetic is one add (was a 64-bit multiply-add chain inside an
    // exec-masked block per load: 87 VALU + 8 branches per k-tile; now the loads schedule into the shadow of the MFMAs).
    // Tensors on this path are < 2 GB (checked on the host), so offsets fit and kOOB = 2^31 is always out of range.
    constexpr unsigned kOOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, p.w_bytes, 0x00020000);
    int a_hi0[NA], a_wi0[NA], a_off0[NA];  // a_off0: element offset of (b, hi0, wi0, kq*4); may be "negative" for halo rows
    bool a_ok[NA];
#pragma unroll
    for (int i = 0; i < NA; i++) {
        const int m = m0 + srow + 32 * i;
        a_ok[i] = m < p.M;
        const int mm = a_ok[i] ? m : 0;
        unsigned b, rem, ho, wo;
        p.d_howo.divmod((unsigned)mm, b, rem);
        p.d_wo.divmod(rem, ho, wo);
        a_hi0[i] = (int)ho * p.stride - p.pad;
        a_wi0[i] = (int)wo * p.stride - p.pad;
        a_off0[i] = (((int)b * p.H + a_hi0[i]) * p.W + a_wi0[i]) * p.Cin + (SMALL_C ? 0 : kq * 4);
    }
    unsigned b_off0[NB];  // byte offset of (n, kq*4), or kOOB for rows past Cout
#pragma unroll
    for (int i = 0; i < NB; i++) {
        const int n = n0 + srow + 32 * i;
        b_off0[i] = n < p.Cout ? (unsigned)(n * p.K + kq * 4) * 4u : kOOB;
    }

    float4 ra[NA], rb[NB];
    auto fetch = [&](__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) -> float4 {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
        if (!SMALL_C) {
            unsigned rs, c0, r, s;
            p.d_cin.divmod((unsigned)k0, rs, c0);
            p.d_s.divmod(rs, r, s);
            const int delta = ((int)r * p.W + (int)s) * p.Cin + (int)c0;  // same for every row of the tile (scalar)
#pragma unroll
            for (int i = 0; i < NA; i++) {
                const int hi = a_hi0[i] + (int)r, wi = a_wi0[i] + (int)s;
                const bool ok = a_ok[i] & ((unsigned)hi < (unsigned)p.H) & ((unsigned)wi < (unsigned)p.W);  // no short circuit: no branch
                ra[i] = fetch(rx, ok ? (unsigned)(a_off0[i] + delta) * 4u : kOOB, 0);
            }
        } else {
            const int k = k0 + kq * 4;
            unsigned rs, c, r, s;
            p.d_cin.divmod((unsigned)k, rs, c);
            p.d_s.divmod(rs, r, s);
            const bool kin = k < p.K;
#pragma unroll
            for (int i = 0; i < NA; i++) {
                const int hi = a_hi0[i] + (int)r, wi = a_wi0[i] + (int)s;
                const bool ok = kin && a_ok[i] && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                ra[i] = fetch(rx, ok ? (unsigned)(a_off0[i] + ((int)r * p.W + (int)s) * p.Cin + (int)c) * 4u : kOOB, 0);
            }
        }
        // weights: rows are K floats long; the K tail (only the stem has one) is cut by the descriptor's range for the last row and
        // by `kin` for the others
        const bool kin = k0 + kq * 4 < p.K;
#pragma unroll
        for (int i = 0; i < NB; i++) rb[i] = fetch(rw, kin ? b_off0[i] : kOOB, k0 * 4);
    };
    auto store_tile = [&](int buf) {
        float* a = As + buf * BM * LDP;
        float* b = Bs + buf * BN * LDP;
#pragma unroll
        for (int i = 0; i < NA; i++) *reinterpret_cast<float4*>(a + (srow + 32 * i) * LDP + kq * 4) = ra[i];
#pragma unroll
        for (int i = 0; i < NB; i++) *reinterpret_cast<float4*>(b + (srow + 32 * i) * LDP + kq * 4) = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    const int l31 = lane & 31, lh = lane >> 5;
    const int a_row0 = wm * (TM * 32) + l31, b_row0 = wn * (TN * 32) + l31;

    auto compute_tile = [&](int cur) {
        const float* a = As + cur * BM * LDP + a_row0 * LDP + lh * 4;
        const float* b = Bs + cur * BN * LDP + b_row0 * LDP + lh * 4;
#pragma unroll
        for (int u = 0; u < BK / 8; u++) {
            float4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; i++) fa[i] = *reinterpret_cast<const float4*>(a + i * 32 * LDP + u * 8);
#pragma unroll
            for (int j = 0; j < TN; j++) fb[j] = *reinterpret_cast<const float4*>(b + j * 32 * LDP + u * 8);
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
                }
        }
    };

    load_tile(kt0);
    store_tile(0);
    __syncthreads();
    // steady state is ONE basic block (the loads are unconditional buffer loads): multiply tile kt, park tile kt+1 in LDS, fetch
    // tile kt+2.  The last tiles are peeled so that nothing is fetched or stored for a tile that does not exist.
    int kt = kt0;
    if (SB) {
        for (; kt + 1 < kt1; kt++) {
            load_tile(kt + 1);
            __builtin_amdgcn_sched_barrier(0);  // keep the fetches AHEAD of the MFMA stream: their latency is what the MFMAs hide
            compute_tile(0);
            __syncthreads();  // every wave is done reading the tile before it is overwritten
            store_tile(0);
            __syncthreads();
        }
        compute_tile(0);
    } else {
        // double-buffered: the fetch of tile kt+2 is issued right behind the barrier of iteration kt (its registers were just
        // emptied into LDS); the compiler sinks the tail of iteration kt's MFMAs below that barrier, so a fetch has a full
        // k-tile of MFMAs (~1.7 us) to land before its ds_write instead of two thirds of one.
        if (kt + 1 < kt1) load_tile(kt + 1);
        for (; kt + 2 < kt1; kt++) {
            const int cur = (kt - kt0) & 1;
            compute_tile(cur);
            store_tile(cur ^ 1);
            load_tile(kt + 2);   // program order puts the fetch BEFORE the barrier: it is issued while the MFMA tail is still to come
            // (explicit sched_group_barrier interleaves of the 64 MFMA / 16 DS read / 8 DS write / 8 VMEM ops were measured: no change)
            __syncthreads();
        }
        if (kt + 1 < kt1) {
            const int cur = (kt - kt0) & 1;
            compute_tile(cur);
            store_tile(cur ^ 1);
            __syncthreads();
            kt++;
        }
        compute_tile((kt - kt0) & 1);
    }
    __syncthreads();  // the epilogue reuses the operand LDS

    if (unit >= 0) {
        // ---- split tile: park the partial sums (thread-major: one coalesced 1 KB store per accumulator register), take a ticket,
        // and only the last arrival goes on.  It re-reads ALL `split` partials in index order -- its own included -- so the sum is
        // the same whichever workgroup happens to finish last (deterministic), then resets the ticket for the next launch.
        // Partials move as 16 B per lane through buffer accesses with sc0|sc1 set (system-coherent: written through / read past the
        // XCD-local L2, so no cache flush or invalidate is needed around the ticket) -- plain loads and stores to the scheduler, so
        // all of a reduction's loads are in flight at once (per-dword agent-scope atomics serialised into ~25 k-tiles of latency).
        constexpr int NQ = TM * TN * 4;                // float4 per thread per partial
        constexpr unsigned kPart = NQ * 256 * 16;      // bytes per partial tile
        constexpr int kCoherent = 0x11;                // cache-policy operand on gfx94x/gfx950: bit 0 = sc0, bit 4 = sc1 (bit 1 is nt)
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t rws = __builtin_amdgcn_make_buffer_rsrc(p.ws, 0, (unsigned)kMaxSplitUnits * kPart, 0x00020000);
        const unsigned my_off = (unsigned)unit * kPart + (unsigned)tid * 16u;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    u32x4 v;
                    v.x = __float_as_uint(acc[i][j][4 * c]); v.y = __float_as_uint(acc[i][j][4 * c + 1]);
                    v.z = __float_as_uint(acc[i][j][4 * c + 2]); v.w = __float_as_uint(acc[i][j][4 * c + 3]);
                    __builtin_amdgcn_raw_buffer_store_b128(v, rws, (int)(my_off + (unsigned)((i * TN + j) * 4 + c) * 4096u), 0, kCoherent);
                }
        // every write-through store of this wave has landed before the ticket is taken.  The explicit wait matters: a workgroup-scope release
        // alone need not drain vmcnt (all waves of a workgroup share one L1), and with 16 short units per tile the last arrival then read partials
        // still in flight (round 5: wrong sums in 7-65 % of a tiny model's outputs, run to run).  No cache maintenance: sc0|sc1 on both sides.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        __shared__ int s_last;
        const int st = unit / p.split;  // index of this split tile
        if (tid == 0) s_last = atomicAdd(p.cnt + st, 1) == p.split - 1;
        __syncthreads();
        if (!s_last) { abr::prof_stamp_end(p.prof_ts); return; }
        if (tid == 0) p.cnt[st] = 0;
        const unsigned base = (unsigned)st * (unsigned)p.split * kPart + (unsigned)tid * 16u;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
                    for (int q0 = 0; q0 < p.split; q0 += 4) {  // fixed order 0..split-1: the same sum whoever arrives last; four partials in flight per round
                        u32x4 v[4];                            // (one load per iteration waited for each in turn: 16 partials = 16 round trips past the L2)
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const int q = q0 + u < p.split ? q0 + u : p.split - 1;
                            v[u] = __builtin_amdgcn_raw_buffer_load_b128(rws, (int)(base + (unsigned)q * kPart + (unsigned)((i * TN + j) * 4 + c) * 4096u), 0, kCoherent);
                        }
#pragma unroll
                        for (int u = 0; u < 4; u++)
                            if (q0 + u < p.split) {
                                sum.x += __uint_as_float(v[u].x); sum.y += __uint_as_float(v[u].y); sum.z += __uint_as_float(v[u].z); sum.w += __uint_as_float(v[u].w);
                            }
                    }
                    acc[i][j][4 * c] = sum.x; acc[i][j][4 * c + 1] = sum.y; acc[i][j][4 * c + 2] = sum.z; acc[i][j][4 * c + 3] = sum.w;
                }
    }

    // ---- epilogue.  C/D layout of the 32x32 MFMA: col(n) = lane&31, row(m) = (r&3) + 8*(r>>2) + 4*(lane>>5): a lane holds ONE
    // column of 16 rows, so storing straight from the accumulators is 64 scalar stores of 128 B segments per wave (measured:
    // ~0.9 TB/s effective, 20-25 % of the kernel at K <= 1024).  Instead each wave transposes its TM*32 x TN*32 tile through its
    // own slice of the (now idle) operand LDS and streams whole rows: 16 B per lane, 128-256 B contiguous per row, with the
    // residual / mask read the same way.  Waves only touch their own slice, so no workgroup barrier is needed here.
    const unsigned ob = epilogue_rows<TM, TN>(p, acc, smem + wave * (32 * (TN * 32 + EPAD)), m0 + wm * (TM * 32), n0 + wn * (TN * 32), lane, out);
    if (p.out_amax) abr::h3_amax_emit(p.out_amax, p.out_epoch, ob);
    abr::prof_stamp_end(p.prof_ts);
}

// ------------------------------------------------------------------------------------------------------------------------
// bf16 math mode (abr_conv_desc::math == ABR_MATH_BF16; BASELINE.json configs[4] "bf16 MFMA backbone"): the SAME implicit GEMM
// with both operands rounded to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) on their way from the fetch registers into LDS
// and multiplied by v_mfma_f32_32x32x16_bf16 with fp32 accumulation; tensors in HBM stay fp32, so every other kernel of the
// step is unchanged and the epilogue is shared.  A bf16 x bf16 product is exact in fp32, so the result equals an fp32 (or
// float64) convolution of the ROUNDED operands up to summation order -- that is what the parity tests check.
// k-tile = 64 (all of this path's channel counts are multiples of 64; the 4-channel stem stays fp32); LDS rows are 64 bf16 +
// 8 pad = 144 B, the same conflict-free pitch as the fp32 tiles; a lane's fragment is one ds_read_b128 = 8 consecutive k.
// 32x32x16 bf16 issues 16x the flops per LDS byte of the fp32 MFMA: the kernel is bound by operand delivery (L2 -> LDS,
// fp32 sources), not by the matrix pipe.
// ------------------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

// Exact three-way bf16 split of four fp32 values (x = h0 + h1 + h2), one uint2 of four bf16 per plane.  Written on PAIRS: one v_cvt_pk_bf16_f32
// rounds two values, its two halves come back as fp32 by a shift and a mask -- as a 4-vector the compiler rounds every element a second time
// for the residual chain (56 instead of 24 v_cvt_pk per 16 values and k-tile in the MFMA loops, where VALU issue slots are what runs out).
// The residual subtractions are pinned to v_sub_f32 (abr::x6_sub): left alone the compiler packs them into v_pk_add_f32, which costs the step
// 0.23 ms next to the other streams' MFMAs (five of five same-session A/B rounds; the kernels' own rates are the same).
__device__ __forceinline__ void x6_split4(const float a, const float b, const float c, const float d, uint2& o0, uint2& o1, uint2& o2) {
    const f32x2v f0 = {a, b}, f1 = {c, d};
    const bf16x2 p0 = __builtin_convertvector(f0, bf16x2), q0 = __builtin_convertvector(f1, bf16x2);
    const f32x2v p0f = __builtin_convertvector(p0, f32x2v), q0f = __builtin_convertvector(q0, f32x2v);
    const f32x2v r0 = {abr::x6_sub(a, p0f.x), abr::x6_sub(b, p0f.y)}, s0 = {abr::x6_sub(c, q0f.x), abr::x6_sub(d, q0f.y)};
    const bf16x2 p1 = __builtin_convertvector(r0, bf16x2), q1 = __builtin_convertvector(s0, bf16x2);
    const f32x2v p1f = __builtin_convertvector(p1, f32x2v), q1f = __builtin_convertvector(q1, f32x2v);
    const f32x2v r1 = {abr::x6_sub(r0.x, p1f.x), abr::x6_sub(r0.y, p1f.y)}, s1 = {abr::x6_sub(s0.x, q1f.x), abr::x6_sub(s0.y, q1f.y)};
    const bf16x2 p2 = __builtin_convertvector(r1, bf16x2), q2 = __builtin_convertvector(s1, bf16x2);
    o0 = make_uint2(*reinterpret_cast<const unsigned*>(&p0), *reinterpret_cast<const unsigned*>(&q0));
    o1 = make_uint2(*reinterpret_cast<const unsigned*>(&p1), *reinterpret_cast<const unsigned*>(&q1));
    o2 = make_uint2(*reinterpret_cast<const unsigned*>(&p2), *reinterpret_cast<const unsigned*>(&q2));
}

constexpr int BKH = 64;        // k per tile
constexpr int LDH = BKH + 8;   // LDS row pitch in bf16 elements (144 B)

template <int BM, int BN, int WM, int WN, bool DB>
__global__ __launch_bounds__(256) void conv_igemm_bf16_kernel(const ConvP p, const float* __restrict__ x, const float* __restrict__ w,
                                                               float* __restrict__ out) {
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int NA = BM / 16, NB = BN / 16;  // float4 staging loads per thread (16 rows x 16 float4 per pass)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    abr::prof_stamp_begin(p.prof_ts);
    constexpr int NBUF = DB ? 2 : 1;
    __bf16* As = reinterpret_cast<__bf16*>(smem);  // [NBUF][BM][LDH]
    __bf16* Bs = As + NBUF * BM * LDH;             // [NBUF][BN][LDH]

    const int tile = (int)abr::xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = tile / p.tiles_n, tile_n = tile % p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int kq = tid & 15, srow = tid >> 4;  // 16 B slot inside the 256 B row segment, row srow + 16*i

    constexpr unsigned kOOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, p.w_bytes, 0x00020000);
    int a_hi0[NA], a_wi0[NA], a_off0[NA];
    bool a_ok[NA];
#pragma unroll
    for (int i = 0; i < NA; i++) {
        const int m = m0 + srow + 16 * i;
        a_ok[i] = m < p.M;
        const int mm = a_ok[i] ? m : 0;
        unsigned b, rem, ho, wo;
        p.d_howo.divmod((unsigned)mm, b, rem);
        p.d_wo.divmod(rem, ho, wo);
        a_hi0[i] = (int)ho * p.stride - p.pad;
        a_wi0[i] = (int)wo * p.stride - p.pad;
        a_off0[i] = (((int)b * p.H + a_hi0[i]) * p.W + a_wi0[i]) * p.Cin + kq * 4;
    }
    unsigned b_off0[NB];
#pragma unroll
    for (int i = 0; i < NB; i++) {
        const int n = n0 + srow + 16 * i;
        b_off0[i] = n < p.Cout ? (unsigned)(n * p.K + kq * 4) * 4u : kOOB;
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 ra[NA], rb[NB];
    auto load_tile = [&](int kt) {
        const int k0 = kt * BKH;
        unsigned rs, c0, r, s;
        p.d_cin.divmod((unsigned)k0, rs, c0);
        p.d_s.divmod(rs, r, s);
        const int delta = ((int)r * p.W + (int)s) * p.Cin + (int)c0;
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const int hi = a_hi0[i] + (int)r, wi = a_wi0[i] + (int)s;
            const bool ok = a_ok[i] & ((unsigned)hi < (unsigned)p.H) & ((unsigned)wi < (unsigned)p.W);
            ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(ok ? (unsigned)(a_off0[i] + delta) * 4u : kOOB), 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NB; i++) rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rw, (int)b_off0[i], k0 * 4, 0);
    };
    auto pack = [](const u32x4 v) -> uint2 {
        const f32x4v f = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
        const bf16x4 h = __builtin_convertvector(f, bf16x4);   // RNE
        return *reinterpret_cast<const uint2*>(&h);
    };
    auto store_tile = [&](int buf) {
        __bf16* a = As + buf * BM * LDH;
        __bf16* b = Bs + buf * BN * LDH;
#pragma unroll
        for (int i = 0; i < NA; i++) *reinterpret_cast<uint2*>(a + (srow + 16 * i) * LDH + kq * 4) = pack(ra[i]);
#pragma unroll
        for (int i = 0; i < NB; i++) *reinterpret_cast<uint2*>(b + (srow + 16 * i) * LDH + kq * 4) = pack(rb[i]);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    const int l31 = lane & 31, lh = lane >> 5;
    const __bf16* a_frag = As + (wm * (TM * 32) + l31) * LDH + lh * 8;
    const __bf16* b_frag = Bs + (wn * (TN * 32) + l31) * LDH + lh * 8;
    auto compute_tile = [&](int cur) {
        const __bf16* af = a_frag + cur * BM * LDH;
        const __bf16* bf = b_frag + cur * BN * LDH;
#pragma unroll
        for (int u = 0; u < BKH / 16; u++) {
            bf16x8 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; i++) fa[i] = *reinterpret_cast<const bf16x8*>(af + i * 32 * LDH + u * 16);
#pragma unroll
            for (int j = 0; j < TN; j++) fb[j] = *reinterpret_cast<const bf16x8*>(bf + j * 32 * LDH + u * 16);
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    };

    const int nk = p.K / BKH;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    int kt = 0;
    if (!DB) {
        for (; kt + 1 < nk; kt++) {
            load_tile(kt + 1);
            __builtin_amdgcn_sched_barrier(0);
            compute_tile(0);
            __syncthreads();
            store_tile(0);
            __syncthreads();
        }
        compute_tile(0);
    } else {  // one barrier per k-tile, tile kt+2 fetched right behind the ds_writes of kt+1 (same shape as the fp32 loop)
        if (kt + 1 < nk) load_tile(kt + 1);
        for (; kt + 2 < nk; kt++) {
            const int cur = kt & 1;
            compute_tile(cur);
            store_tile(cur ^ 1);
            load_tile(kt + 2);
            __syncthreads();
        }
        if (kt + 1 < nk) {
            const int cur = kt & 1;
            compute_tile(cur);
            store_tile(cur ^ 1);
            __syncthreads();
            kt++;
        }
        compute_tile(kt & 1);
    }
    __syncthreads();  // the epilogue reuses the operand LDS
    const unsigned ob = epilogue_rows<TM, TN>(p, acc, smem + wave * (32 * (TN * 32 + EPAD)), m0 + wm * (TM * 32), n0 + wn * (TN * 32), lane, out);
    if (p.out_amax) abr::h3_amax_emit(p.out_amax, p.out_epoch, ob);
    abr::prof_stamp_end(p.prof_ts);
}

template <int BM, int BN, int WM, int WN, bool DB>
int launch_bf16(const ConvP& p, const float* x, const float* w, float* out, hipStream_t st) {
    ConvP q = p;
    q.tiles_m = (p.M + BM - 1) / BM;
    q.tiles_n = (p.Cout + BN - 1) / BN;
    q.tiles_pb = q.tiles_m * q.tiles_n;
    q.nbatch = 1; q.n_full = q.tiles_pb; q.split = 1; q.ws = nullptr; q.cnt = nullptr;
    constexpr size_t lds_op = sizeof(__bf16) * (DB ? 2 : 1) * (BM + BN) * LDH;
    constexpr size_t lds_ep = sizeof(float) * 4 * 32 * (BN / WN + EPAD);
    const size_t lds = lds_op > lds_ep ? lds_op : lds_ep;
    auto kern = conv_igemm_bf16_kernel<BM, BN, WM, WN, DB>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    q.prof_ts = abr::prof_stamp_slot(abr::PROF_IGEMM_BF16, 2.0 * (double)p.M * (double)p.Cout * (double)p.K);
    kern<<<(unsigned)q.tiles_pb, 256, lds, st>>>(q, x, w, out);
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------------
// fp32-accurate math on the bf16 matrix cores (abr_conv_desc::math == ABR_MATH_BF16X6; the host's DEFAULT since round 2).
// Every fp32 operand is split EXACTLY into three bf16 terms, x = x0 + x1 + x2 (x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0
// - x1): 3 x (8 bits + sign) cover the 24-bit significand; both subtractions are exact in fp32), and the product is formed from
// the six cross terms with i + j <= 2 -- x0w0, x0w1, x1w0, x0w2, x1w1, x2w0 -- each an exact bf16 x bf16 product accumulated in
// fp32 by v_mfma_f32_32x32x16_bf16.  The three dropped terms are below 2^-26 |x||w|, under the rounding of a single fp32
// product, so the result carries the same error bound as an fp32 FMA chain (tests: 1e-5 of the output scale against float64,
// TIGHTER than the fp32 kernels' criterion).  Six bf16 MFMAs deliver 2.5 PF / 6 = 417 TFLOP/s of fp32-equivalent work against
// 157 for v_mfma_f32_32x32x2_f32.
// k-tile 32; three bf16 planes per operand in LDS (pitch 80 B: conflict-free ds_read_b128), 61.4 KB single-buffered = two
// workgroups per CU; per 16-k step a wave reads (TM + TN) x 3 fragments for TM x TN x 6 MFMAs.  Handles the batched
// (Winograd-domain) GEMMs like the fp32 kernel.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int BKX = 32;
constexpr int LDX = BKX + 8;   // LDS row pitch in bf16 elements (80 B)

// (rounds 1-5 also carried conv_igemm_x6_kernel, which split the WEIGHT tile in every workgroup, for calls without packed planes, and the opt-in
//  intra-workgroup split-K conv_igemm_x6wk_kernel; round 6 retired both: a call without planes packs them into stream scratch first -- same
//  products in the same order, bit-identical results -- and every bf16x6 / f16x3 / bf16 contraction runs on the weights-direct kernel below.)

// ------------------------------------------------------------------------------------------------------------------------
// bf16x6 with the WEIGHT operand fed straight from global memory into the MFMA registers (conv_igemm_x6w_kernel).
// The weights of a conv change once per optimiser step (never, for the frozen source model), so their exact bf16x3 split is
// made once per version by x6_pack_kernel and stored in MFMA-FRAGMENT order:
//     chunk(nb, ks, pl) = 64 lanes x 16 B; lane l holds row nb*32 + (l & 31), k = ks*16 + (l >> 5)*8 .. +8 of plane pl
//     byte address     = (((nb * K/16 + ks) * 3 + pl) * 64 + l) * 16
// A wave's B fragment is then ONE coalesced 1 KB buffer load into the registers the MFMA reads: no LDS store, no fragment read and
// no split arithmetic for the weights; LDS carries only the three A planes (30.7 KB per 128-row tile instead of 61.4), which lets
// three workgroups (or more) share a CU.  Per 16-k step a wave issues TM*3 ds_read_b128 + TN*3 global loads for TM*TN*6 MFMAs.
// The B fragments of step u of k-tile kt+1 are requested right behind the last MFMA that reads step u of tile kt, so every load has
// half a k-tile to a whole one (768 - 1536 MFMA cycles per wave) to land.  The six products of a step are interleaved over the
// wave's accumulators (each accumulator still receives them smallest-first), not issued as six dependent MFMAs back to back.
// Measured (tools/x6lab/lab7.hip, random operands): 32768x2048x1024 0.70 -> 0.63 ms, 32768x512x2048 0.38 -> 0.32, 9600x1024x256 +22 %.
// ------------------------------------------------------------------------------------------------------------------------
// PLAIN: 1x1, stride 1, no padding (every GEMM but the strided 1x1s and the direct 3x3s; all 36 x n Winograd GEMMs): row m of A is x + m * Cin, the
// k-tile advances through the load's scalar offset -- no per-row address / halo arithmetic in the MFMA loop (it was 36 VALU instructions per k-tile,
// four of them quarter-rate 32-bit multiplies the compiler rematerialised the offsets with).
// NP: products per multiply-add.  6 = the bf16x6 arithmetic; 1 = ABR_MATH_BF16 (both operands ROUNDED to bf16, one product): the SAME kernel reading
// plane 0 of the packed weights and keeping only the first plane of the activation split -- a third of the LDS and weight-fragment traffic and a sixth
// of the MFMAs, with the tile shapes, the weights-direct operand path and the epilogue of the default arithmetic (round 4; the round-1 bf16 kernels
// converted fp32 operands of BOTH sides in the loop and ran the step 20 % slower than bf16x6).
template <int BM, int BN, int WM, int WN, bool PLAIN, int NP = 6>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void conv_igemm_x6w_kernel(const ConvP p, const float* __restrict__ x_,
                                                                                                        float* __restrict__ out_) {
    const float* x = x_;
    float* out = out_;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int NA = BM / 32;
    constexpr bool H3 = NP == 3;           // ABR_MATH_F16X3: two fp16 planes per operand, three products (common.h)
    constexpr int NPL = NP == 1 ? 1 : (H3 ? 2 : 3);   // operand planes in use
    constexpr int NPW = H3 ? 2 : 3;        // planes per k-step in the packed weights (the bf16 mode reads plane 0 of the bf16x3 packing)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    abr::prof_stamp_begin(p.prof_ts);
    __bf16* As = reinterpret_cast<__bf16*>(smem);  // [3][BM][LDX]

    int tile = (int)abr::xcd_remap(blockIdx.x, gridDim.x);
    const char* wp = reinterpret_cast<const char*>(p.w_planes);
    int wsc_bt = 0;
    if (p.nbatch > 1) {
        const int bt = tile / p.tiles_pb;
        wsc_bt = bt;
        tile -= bt * p.tiles_pb;
        x += bt * p.a_bs; out += bt * p.o_bs; wp += bt * p.wp_bs;
    }
    int tile_m = tile / p.tiles_n, tile_n = tile % p.tiles_n;
    if (p.ngroup > 0) {
        // n-group-major: `ngroup` n-tile columns at a time, all their m-tiles, n fastest inside the group.  An XCD's contiguous tile range
        // (abr::xcd_remap) then stays on a few weight columns whose packed planes fit its L2 -- the weight fragments, which go global -> registers
        // and are what the MFMAs wait for first, are L2 hits -- and streams the activation rows (prefetched a k-tile ahead) past them.
        const int per = p.tiles_m * p.ngroup;
        const int ng = tile / per, rem = tile - ng * per;
        const int gn = min(p.ngroup, p.tiles_n - ng * p.ngroup);   // (the last group may be narrower)
        tile_m = rem / gn;
        tile_n = ng * p.ngroup + rem - tile_m * gn;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // A thread stores 8 B per plane at (row * 80 + kq * 8) B.  ds_write_b64 is served in groups of 16 consecutive lanes over 32 banks: with rows
    // (s, s + 1) in a group the two 64 B runs start 20 dwords apart and share 4 banks (every store took 2 LDS cycles per group: 25 - 33 % of all LDS
    // cycles, SQ_LDS_BANK_CONFLICT of profiles/r03_pmc_mfma.json).  Rows (s, s + 4) start 80 dwords = 16 banks apart: conflict-free.  Lane bit 3
    // therefore carries row bit 2 and lane bit 5 row bit 0; each 8-lane set still fetches one 128 B row segment, the fragment reads do not change.
    const int kq = tid & 7, srow = ((tid >> 3) & ~5) | (((tid >> 3) & 1) << 2) | ((tid >> 5) & 1);

    constexpr unsigned kOOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwp = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wp), 0, p.wp_bytes, 0x00020000);
    int a_hi0[NA], a_wi0[NA], a_off0[NA];
    bool a_ok[NA];
    unsigned a_voff[NA];   // PLAIN: byte offset of (row, kq) or out of range
#pragma unroll
    for (int i = 0; i < NA; i++) {
        const int m = m0 + srow + 32 * i;
        a_ok[i] = m < p.M;
        if constexpr (PLAIN) {
            a_voff[i] = a_ok[i] ? (unsigned)(m * p.Cin + kq * 4) * 4u : kOOB;
        } else {
            const int mm = a_ok[i] ? m : 0;
            unsigned b, rem, ho, wo;
            p.d_howo.divmod((unsigned)mm, b, rem);
            p.d_wo.divmod(rem, ho, wo);
            a_hi0[i] = (int)ho * p.stride - p.pad;
            a_wi0[i] = (int)wo * p.stride - p.pad;
            a_off0[i] = (((int)b * p.H + a_hi0[i]) * p.W + a_wi0[i]) * p.Cin + kq * 4;
        }
    }
    const int KS = p.K / 16;
    unsigned bo[TN];   // byte offset of chunk (nb, 0, 0) for this lane; 32-row blocks past the packed matrix read as zeros
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int nb = (n0 + wn * (TN * 32)) / 32 + j;
        bo[j] = nb < p.wp_nblocks ? (unsigned)(((size_t)nb * KS * NPW * 64 + lane) * 16) : kOOB;
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 ra[NA];
    u32x4 fbr[2][TN][NPL];   // [step of the k-tile][n-block][plane]
    auto load_a = [&](int kt) {
        const int k0 = kt * BKX;
        if constexpr (PLAIN) {
#pragma unroll
            for (int i = 0; i < NA; i++) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)a_voff[i], k0 * 4, 0);
        } else {
            unsigned rs, c0, r, s;
            p.d_cin.divmod((unsigned)k0, rs, c0);
            p.d_s.divmod(rs, r, s);
            const int delta = ((int)r * p.W + (int)s) * p.Cin + (int)c0;
#pragma unroll
            for (int i = 0; i < NA; i++) {
                const int hi = a_hi0[i] + (int)r, wi = a_wi0[i] + (int)s;
                const bool ok = a_ok[i] & ((unsigned)hi < (unsigned)p.H) & ((unsigned)wi < (unsigned)p.W);
                ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(ok ? (unsigned)(a_off0[i] + delta) * 4u : kOOB), 0, 0);
            }
        }
    };
    auto load_b = [&](int kt, int u) {
        const int ks = kt * 2 + u;
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int pl = 0; pl < NPL; pl++) fbr[u][j][pl] = __builtin_amdgcn_raw_buffer_load_b128(rwp, (int)bo[j], (ks * NPW + pl) * 1024, 0);
    };
    // range guard: the A rows are inspected by the workgroups of the first n-tile column; the weights were inspected when they were packed
    const bool chk_a = p.x6_flags && tile_n == 0;
    unsigned bmin = 0xFFFFFFFFu;
    float nonfin = 0.f;
    // f16x3: the scale of the A operand's split from its amax word (every workgroup reads the same 8 bytes)
    unsigned a_bits = 0;
    float sa = 1.f, inv_sa = 1.f;
    if constexpr (H3) {
        a_bits = abr::h3_amax_load(p.a_amax, p.a_epoch);
        abr::h3_scales(a_bits, sa, inv_sa);
    }
    const unsigned small_thr = H3 ? abr::h3_small_threshold(a_bits) : 0u;
    unsigned nsmall = 0;
    auto store_a = [&]() {
        if constexpr (H3) {
            if (chk_a) {
#pragma unroll
                for (int i = 0; i < NA; i++) {
                    const u32x4 v = ra[i];
                    nsmall += (unsigned)(((v.x << 1) - 1u) < small_thr) + (unsigned)(((v.y << 1) - 1u) < small_thr) + (unsigned)(((v.z << 1) - 1u) < small_thr) +
                              (unsigned)(((v.w << 1) - 1u) < small_thr);
                }
            }
#pragma unroll
            for (int i = 0; i < NA; i++) {
                const u32x4 v = ra[i];
                __bf16* dst = As + (srow + 32 * i) * LDX + kq * 4;
                uint2 o0, o1;
                abr::h3_split2(__uint_as_float(v.x), __uint_as_float(v.y), inv_sa, o0.x, o1.x);
                abr::h3_split2(__uint_as_float(v.z), __uint_as_float(v.w), inv_sa, o0.y, o1.y);
                *reinterpret_cast<uint2*>(dst) = o0;
                *reinterpret_cast<uint2*>(dst + BM * LDX) = o1;
            }
            return;
        }
        if (chk_a) {
#pragma unroll
            for (int i = 0; i < NA; i++) {
                const u32x4 v = ra[i];
                const unsigned b0 = (v.x << 1) - 1u, b1 = (v.y << 1) - 1u, b2 = (v.z << 1) - 1u, b3 = (v.w << 1) - 1u;
                bmin = min(min(bmin, min(b0, b1)), min(b2, b3));
                nonfin = fmaf(__uint_as_float(v.x), 0.f, nonfin); nonfin = fmaf(__uint_as_float(v.y), 0.f, nonfin);
                nonfin = fmaf(__uint_as_float(v.z), 0.f, nonfin); nonfin = fmaf(__uint_as_float(v.w), 0.f, nonfin);
            }
        }
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const u32x4 v = ra[i];
            __bf16* dst = As + (srow + 32 * i) * LDX + kq * 4;
            uint2 o0, o1, o2;
            x6_split4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w), o0, o1, o2);
            *reinterpret_cast<uint2*>(dst) = o0;
            if constexpr (NP != 1) {
                *reinterpret_cast<uint2*>(dst + BM * LDX) = o1;
                *reinterpret_cast<uint2*>(dst + 2 * BM * LDX) = o2;
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    const int l31 = lane & 31, lh = lane >> 5;
    const __bf16* a_frag = As + (wm * (TM * 32) + l31) * LDX + lh * 8;
    const int nk = p.K / BKX;
    auto compute_tile = [&](int kt_next) {   // kt_next < nk: the B fragments of that tile are requested as this tile's are consumed
#pragma unroll
        for (int u = 0; u < BKX / 16; u++) {
            bf16x8 fa[TM][NPL], fb[TN][NPL];
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int pl = 0; pl < NPL; pl++) fa[i][pl] = *reinterpret_cast<const bf16x8*>(a_frag + pl * BM * LDX + i * 32 * LDX + u * 16);
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int pl = 0; pl < NPL; pl++) fb[j][pl] = *reinterpret_cast<const bf16x8*>(&fbr[u][j][pl]);
            constexpr int pa[6] = {NP == 1 ? 0 : (H3 ? 1 : 2), 0, H3 ? 0 : 1, 1, 0, 0}, pb[6] = {0, NP == 1 ? 0 : (H3 ? 1 : 2), H3 ? 0 : 1, 0, 1, 0};   // (A plane, B plane) of the products, smallest first
#pragma unroll
            for (int t = 0; t < NP; t++)
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++) {
                        if constexpr (H3) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[i][pa[t]]), __builtin_bit_cast(f16x8, fb[j][pb[t]]), acc[i][j], 0, 0, 0);
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][pa[t]], fb[j][pb[t]], acc[i][j], 0, 0, 0);
                    }
            if (kt_next < nk) load_b(kt_next, u);
            if (u == 0) __builtin_amdgcn_sched_barrier(0);   // keeps step 0's refill ahead of step 1's MFMAs (lab7: +3 %)
        }
    };
    load_b(0, 0);
    load_b(0, 1);
    load_a(0);
    store_a();
    __syncthreads();
    for (int kt = 0; kt + 1 < nk; kt++) {
        load_a(kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        compute_tile(kt + 1);
        __syncthreads();
        store_a();
        __syncthreads();
    }
    compute_tile(nk);
    if constexpr (H3) {
        if (chk_a) abr::h3_report(a_bits, nsmall, p.x6_flags, p.h3_stats);
    } else {
        if (chk_a) abr::x6_report(bmin, nonfin, p.x6_flags);
    }
    __syncthreads();  // the epilogue reuses the operand LDS
    const float* wsc = nullptr;
    if constexpr (H3) wsc = p.w_scales + (p.nbatch > 1 ? (size_t)wsc_bt * p.Cout : 0);
    const unsigned ob = epilogue_rows<TM, TN, true>(p, acc, smem + wave * (32 * (TN * 32 + EPAD)), m0 + wm * (TM * 32), n0 + wn * (TN * 32), lane, out, wsc, sa);
    if (p.out_amax) abr::h3_amax_emit(p.out_amax, p.out_epoch, ob);
    abr::prof_stamp_end(p.prof_ts);
}


// ------------------------------------------------------------------------------------------------------------------------
// Fused TAIL of a 64-wide bottleneck that has no backward pass (the frozen stem's layer1, run by BOTH models every step):
//     o2  = relu(bn2(conv3x3(o1)))          150000 x 64 x 576   -- conv_igemm_x6w_kernel<128,64,2,2,false>'s main loop, unchanged
//     out = relu(bn3(conv1x1(o2)) + idt)    150000 x 256 x 64   -- the same products in the same order as the stand-alone launch
// in ONE launch: a workgroup owns 128 output pixels and all channels; o2 (128 x 64) never leaves the CU -- it goes from the accumulators
// through the conv epilogue arithmetic (scale, bias, ReLU) and the exact bf16x3 split straight into LDS planes in A-fragment order
// (pitch 144 B: conflict-free ds_read_b128) and is multiplied by conv3's packed weights (global -> MFMA registers) right there.  Saved per
// block: o2's write and read (77 MB at B = 4), one launch, and conv3's prologue / short K = 64 main loop -- that launch moved 346 MB to
// execute 4.9 GF and ran at 59 TF-eq.  Results are bit-identical to the two launches (tests/test_gpu_ops.py::test_bottleneck_tail_fused).
// Reference: maskrcnn_benchmark/modeling/backbone/resnet.py:327-346 (conv2 -> bn2 -> relu -> conv3 -> bn3 -> += identity -> relu).
// p = conv2 (R = S = 3, stride 1, pad 1, Cin = Cout = 64), q = conv3 (1x1, Cin = 64, Cout = 256, residual, relu); both with packed planes.
constexpr int T64_O2P = 72;   // O2 plane pitch in bf16 elements (144 B)
constexpr size_t kTail64Lds = sizeof(__bf16) * 3 * 128 * T64_O2P + sizeof(float) * 4 * 32 * (32 + EPAD);

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void conv_tail64_x6w_kernel(const ConvP p, const ConvP q, const float* __restrict__ x,
                                                                                                       float* __restrict__ out) {
    constexpr int BM = 128, WN = 2, TM = 2, NA = BM / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    abr::prof_stamp_begin(p.prof_ts);
    __bf16* As = reinterpret_cast<__bf16*>(smem);   // phase A: [3][BM][LDX] operand planes;  phase B: [3][BM][T64_O2P] planes of o2 (aliased)
    float* ep_all = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + sizeof(__bf16) * 3 * 128 * T64_O2P);

    const int tile_m = (int)abr::xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = tile_m * BM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int kq = tid & 7, srow = ((tid >> 3) & ~5) | (((tid >> 3) & 1) << 2) | ((tid >> 5) & 1);   // (conflict-free store rows, see the x6w kernel)

    constexpr unsigned kOOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwp = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w_planes), 0, p.wp_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwq = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(q.w_planes), 0, q.wp_bytes, 0x00020000);
    int a_hi0[NA], a_wi0[NA], a_off0[NA];
    bool a_ok[NA];
#pragma unroll
    for (int i = 0; i < NA; i++) {
        const int m = m0 + srow + 32 * i;
        a_ok[i] = m < p.M;
        const int mm = a_ok[i] ? m : 0;
        unsigned b, rem, ho, wo;
        p.d_howo.divmod((unsigned)mm, b, rem);
        p.d_wo.divmod(rem, ho, wo);
        a_hi0[i] = (int)ho - p.pad;
        a_wi0[i] = (int)wo - p.pad;
        a_off0[i] = (((int)b * p.H + a_hi0[i]) * p.W + a_wi0[i]) * p.Cin + kq * 4;
    }
    const int KS = p.K / 16;
    const unsigned bo = (unsigned)(((size_t)wn * KS * 3 * 64 + lane) * 16);   // conv2's n-block wn
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 ra0[NA], ra1[NA], ra2[NA];   // THREE k-tiles of A in flight (tile t in set t % 3): see the loop
    u32x4 fbr[2][3];
    auto load_a = [&](int kt, u32x4 (&ra)[NA]) {
        unsigned rs, c0, r, s;
        p.d_cin.divmod((unsigned)(kt * BKX), rs, c0);
        p.d_s.divmod(rs, r, s);
        const int delta = ((int)r * p.W + (int)s) * p.Cin + (int)c0;
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const int hi = a_hi0[i] + (int)r, wi = a_wi0[i] + (int)s;
            const bool ok = a_ok[i] & ((unsigned)hi < (unsigned)p.H) & ((unsigned)wi < (unsigned)p.W);
            ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(ok ? (unsigned)(a_off0[i] + delta) * 4u : kOOB), 0, 0);
        }
    };
    auto load_b = [&](int kt, int u) {
        const int ks = kt * 2 + u;
#pragma unroll
        for (int pl = 0; pl < 3; pl++) fbr[u][pl] = __builtin_amdgcn_raw_buffer_load_b128(rwp, (int)bo, (ks * 3 + pl) * 1024, 0);
    };
    unsigned bmin = 0xFFFFFFFFu;
    float nonfin = 0.f;
    auto store_a = [&](u32x4 (&ra)[NA], int buf) {
        if (p.x6_flags) {
#pragma unroll
            for (int i = 0; i < NA; i++) {
                const u32x4 v = ra[i];
                const unsigned b0 = (v.x << 1) - 1u, b1 = (v.y << 1) - 1u, b2 = (v.z << 1) - 1u, b3 = (v.w << 1) - 1u;
                bmin = min(min(bmin, min(b0, b1)), min(b2, b3));
                nonfin = fmaf(__uint_as_float(v.x), 0.f, nonfin); nonfin = fmaf(__uint_as_float(v.y), 0.f, nonfin);
                nonfin = fmaf(__uint_as_float(v.z), 0.f, nonfin); nonfin = fmaf(__uint_as_float(v.w), 0.f, nonfin);
            }
        }
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const u32x4 v = ra[i];
            __bf16* dst = As + buf * (3 * BM * LDX) + (srow + 32 * i) * LDX + kq * 4;
            uint2 o0, o1, o2;
            x6_split4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w), o0, o1, o2);
            *reinterpret_cast<uint2*>(dst) = o0;
            *reinterpret_cast<uint2*>(dst + BM * LDX) = o1;
            *reinterpret_cast<uint2*>(dst + 2 * BM * LDX) = o2;
        }
    };
    f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    const int l31 = lane & 31, lh = lane >> 5;
    const __bf16* a_frag = As + (wm * (TM * 32) + l31) * LDX + lh * 8;
    const int nk = p.K / BKX;
    constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};   // (A plane, B plane) of the six products, smallest first
    auto compute_tile = [&](int kt_next, int buf) {
#pragma unroll
        for (int u = 0; u < BKX / 16; u++) {
            bf16x8 fa[TM][3], fb[3];
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int pl = 0; pl < 3; pl++) fa[i][pl] = *reinterpret_cast<const bf16x8*>(a_frag + buf * (3 * BM * LDX) + pl * BM * LDX + i * 32 * LDX + u * 16);
#pragma unroll
            for (int pl = 0; pl < 3; pl++) fb[pl] = *reinterpret_cast<const bf16x8*>(&fbr[u][pl]);
#pragma unroll
            for (int t = 0; t < 6; t++)
#pragma unroll
                for (int i = 0; i < TM; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][pa[t]], fb[pb[t]], acc[i], 0, 0, 0);
            if (kt_next < nk) load_b(kt_next, u);
            if (u == 0) __builtin_amdgcn_sched_barrier(0);
        }
    };
    // A 128 x 64 tile gives a wave only 24 MFMAs (~0.4 us) per k-tile, and this kernel runs at two waves per SIMD (LDS-bound occupancy): the
    // registers for THREE k-tiles of A in flight are free, and the operand planes are double-buffered in LDS (61 KB of the 74 KB this kernel
    // owns anyway) so that the split / store of tile kt + 1 shares ONE barrier interval with the MFMAs of tile kt instead of waiting behind them.
    load_b(0, 0);
    load_b(0, 1);
    load_a(0, ra0);
    if (nk > 1) load_a(1, ra1);
    if (nk > 2) load_a(2, ra2);
    store_a(ra0, 0);
    __syncthreads();
    auto k_iter = [&](int kt, u32x4 (&r_free)[NA], u32x4 (&r_next)[NA]) {   // LDS buffer kt & 1 holds tile kt; r_free held it; r_next holds tile kt + 1
        if (kt + 3 < nk) load_a(kt + 3, r_free);
        __builtin_amdgcn_sched_barrier(0);
        compute_tile(kt + 1, kt & 1);
        store_a(r_next, (kt + 1) & 1);   // the other buffer: last read for tile kt - 1, before the previous barrier
        __syncthreads();
    };
    {
        int kt = 0;
        for (; kt + 3 < nk; kt += 3) {
            k_iter(kt, ra0, ra1);
            k_iter(kt + 1, ra1, ra2);
            k_iter(kt + 2, ra2, ra0);
        }
        if (kt + 1 < nk) {
            k_iter(kt, ra0, ra1);
            if (kt + 2 < nk) k_iter(kt + 1, ra1, ra2);
        }
    }
    compute_tile(nk, (nk - 1) & 1);
    if (p.x6_flags) abr::x6_report(bmin, nonfin, p.x6_flags);

// ---- conv3's weight fragments of this wave's first 32 output channels: on their way while o2 is formed
    const int KS3 = q.K / 16;   // 4
    u32x4 fq[4][3];             // [k-step][plane] of the n-block in hand (the second block's are requested behind the first block's MFMAs)
    auto load_q = [&](int j) {
        const unsigned bq = (unsigned)(((size_t)(2 * wave + j) * KS3 * 3 * 64 + lane) * 16);
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int pl = 0; pl < 3; pl++) fq[u][pl] = __builtin_amdgcn_raw_buffer_load_b128(rwq, (int)bq, (u * 3 + pl) * 1024, 0);
    };
    load_q(0);
    __syncthreads();   // every wave is done with conv2's operand planes: the region becomes o2's

    // ---- epilogue of conv2 (the arithmetic of epilogue_rows: v * scale + bias, ReLU), exact split, planes of o2 into LDS
    {
        const int n = wn * 32 + l31;
        const float sc = p.scale ? p.scale[n] : 1.f, bi = p.bias ? p.bias[n] : 0.f;
        unsigned bmin2 = 0xFFFFFFFFu;
        float nonfin2 = 0.f;
        __bf16* o2p = As + n;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                float v0 = acc[i][r] * sc + bi, v1 = acc[i][r + 1] * sc + bi;
                if (p.relu) { v0 = abr::relu_f(v0); v1 = abr::relu_f(v1); }
                if (q.x6_flags) {   // conv3's operand: inspected once, as the stand-alone launch's first n-tile column does
                    const unsigned b0 = (__float_as_uint(v0) << 1) - 1u, b1 = (__float_as_uint(v1) << 1) - 1u;
                    bmin2 = min(bmin2, min(b0, b1));
                    nonfin2 = fmaf(v0, 0.f, nonfin2); nonfin2 = fmaf(v1, 0.f, nonfin2);
                }
                uint2 h0, h1, h2;
                x6_split4(v0, v1, 0.f, 0.f, h0, h1, h2);
                const int row = wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;   // rows of accumulator registers r, r + 1
                unsigned short* d0 = reinterpret_cast<unsigned short*>(o2p + row * T64_O2P);
                d0[0] = (unsigned short)(h0.x & 0xFFFFu);                 d0[T64_O2P] = (unsigned short)(h0.x >> 16);
                d0[128 * T64_O2P] = (unsigned short)(h1.x & 0xFFFFu);     d0[128 * T64_O2P + T64_O2P] = (unsigned short)(h1.x >> 16);
                d0[2 * 128 * T64_O2P] = (unsigned short)(h2.x & 0xFFFFu); d0[2 * 128 * T64_O2P + T64_O2P] = (unsigned short)(h2.x >> 16);
            }
        if (q.x6_flags) abr::x6_report(bmin2, nonfin2, q.x6_flags);
    }
    __syncthreads();

    // ---- conv3: [128 x 64] (LDS planes) x [64 x 256] (packed planes, global -> registers); wave w owns output channels [64 w, 64 w + 64)
    // in two passes of 32.  The output phase moves 256 KB per workgroup (residual in, result out) with only eight waves on the CU: every
    // residual vector of a pass (16 x 16 B per lane) is requested BEFORE the pass's MFMAs -- with epilogue_rows' four loads in flight per
    // wave this phase alone took 107 of the launch's 214 us.  The arithmetic is epilogue_rows': (v * scale + bias) + residual, ReLU.
    const __bf16* o_frag = As + l31 * T64_O2P + lh * 8;
    float* ep = ep_all + wave * (32 * (32 + EPAD));
    constexpr int EP = 32 + EPAD;
    typedef float nt4 __attribute__((ext_vector_type(4)));
    const int c4 = (lane & 7) * 4, erow = lane >> 3;   // 8 lanes per 128 B row segment, 8 rows per wave iteration
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int ncol = (2 * wave + j) * 32 + c4;
        nt4 res[4][4];
        if (q.residual) {
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int it = 0; it < 4; it++) {
                    const int m = m0 + i * 32 + it * 8 + erow;
                    res[i][it] = m < q.M ? __builtin_nontemporal_load(reinterpret_cast<const nt4*>(q.residual + (size_t)m * q.Cout + ncol)) : nt4{0.f, 0.f, 0.f, 0.f};
                }
        }
        const float4 sc4 = q.scale ? *reinterpret_cast<const float4*>(q.scale + ncol) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 bi4 = q.bias ? *reinterpret_cast<const float4*>(q.bias + ncol) : make_float4(0.f, 0.f, 0.f, 0.f);
        f32x16 acc3[4];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc3[i][r] = 0.f;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            bf16x8 fa[4][3], fb[3];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int pl = 0; pl < 3; pl++) fa[i][pl] = *reinterpret_cast<const bf16x8*>(o_frag + pl * 128 * T64_O2P + i * 32 * T64_O2P + u * 16);
#pragma unroll
            for (int pl = 0; pl < 3; pl++) fb[pl] = *reinterpret_cast<const bf16x8*>(&fq[u][pl]);
#pragma unroll
            for (int t = 0; t < 6; t++)
#pragma unroll
                for (int i = 0; i < 4; i++) acc3[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][pa[t]], fb[pb[t]], acc3[i], 0, 0, 0);
        }
        if (j == 0) load_q(1);
#pragma unroll
        for (int i = 0; i < 4; i++) {
#pragma unroll
            for (int r = 0; r < 16; r++) ep[((r & 3) + 8 * (r >> 2) + 4 * lh) * EP + l31] = acc3[i][r];   // wave-private transpose (as epilogue_rows)
#pragma unroll
            for (int it = 0; it < 4; it++) {
                const int row = it * 8 + erow, m = m0 + i * 32 + row;
                float4 v = *reinterpret_cast<const float4*>(ep + row * EP + c4);
                v.x = v.x * sc4.x + bi4.x; v.y = v.y * sc4.y + bi4.y; v.z = v.z * sc4.z + bi4.z; v.w = v.w * sc4.w + bi4.w;
                if (q.residual) { v.x += res[i][it].x; v.y += res[i][it].y; v.z += res[i][it].z; v.w += res[i][it].w; }
                if (q.relu) { v.x = abr::relu_f(v.x); v.y = abr::relu_f(v.y); v.z = abr::relu_f(v.z); v.w = abr::relu_f(v.w); }
                if (m < q.M) *reinterpret_cast<float4*>(out + (size_t)m * q.Cout + ncol) = v;
            }
        }
    }
    abr::prof_stamp_end(p.prof_ts);
}

// fp32 matrix [rows][K] (K % 16 == 0) -> fragment-packed bf16x3 planes (layout above), rows padded with zeros to a multiple of 32.
// One workgroup = one 32-row block x 64 k: the fp32 block comes in as whole 256 B row segments (coalesced), goes through LDS, and
// leaves as 4 k-steps x 3 planes x 1 KB chunks.  Every weight element is range-checked here (abr_x6_range_flags), once per version.
__global__ __launch_bounds__(256) void x6_pack_kernel(const float* __restrict__ w, int rows, int K, uint4* __restrict__ planes, unsigned* flags) {
    __shared__ float t[32][68];
    const int nb = blockIdx.y, k0 = blockIdx.x * 64, tid = threadIdx.x;
    unsigned bmin = 0xFFFFFFFFu;
    float nonfin = 0.f;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int r = (tid >> 4) + 16 * h, c = (tid & 15) * 4, row = nb * 32 + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < rows && k0 + c < K) v = *reinterpret_cast<const float4*>(w + (size_t)row * K + k0 + c);
        *reinterpret_cast<float4*>(&t[r][c]) = v;
        const unsigned b0 = (__float_as_uint(v.x) << 1) - 1u, b1 = (__float_as_uint(v.y) << 1) - 1u, b2 = (__float_as_uint(v.z) << 1) - 1u,
                       b3 = (__float_as_uint(v.w) << 1) - 1u;
        bmin = min(min(bmin, min(b0, b1)), min(b2, b3));
        nonfin = fmaf(v.x, 0.f, nonfin); nonfin = fmaf(v.y, 0.f, nonfin); nonfin = fmaf(v.z, 0.f, nonfin); nonfin = fmaf(v.w, 0.f, nonfin);
    }
    if (flags) abr::x6_report(bmin, nonfin, flags);
    __syncthreads();
    const int ksl = tid >> 6, lane = tid & 63, ks = k0 / 16 + ksl;
    if (ks * 16 >= K) return;
    const float* src = &t[lane & 31][ksl * 16 + (lane >> 5) * 8];
    __bf16 h[3][8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const float v = src[e];
        const __bf16 h0 = (__bf16)v;
        const float r1 = v - (float)h0;
        const __bf16 h1 = (__bf16)r1;
        h[0][e] = h0; h[1][e] = h1; h[2][e] = (__bf16)(r1 - (float)h1);
    }
    const size_t c = ((size_t)nb * (K / 16) + ks) * 3;
#pragma unroll
    for (int pl = 0; pl < 3; pl++) planes[(c + pl) * 64 + lane] = *reinterpret_cast<const uint4*>(h[pl]);
}

static int64_t x6_packed_bytes(int64_t rows, int64_t K) { return (rows + 31) / 32 * 32 * K * 6; }

static int x6_pack(const float* w, int64_t rows, int K, void* planes, hipStream_t st) {
    dim3 grid((unsigned)((K + 63) / 64), (unsigned)((rows + 31) / 32));
    x6_pack_kernel<<<grid, 256, 0, st>>>(w, (int)rows, K, reinterpret_cast<uint4*>(planes), abr::x6_guard_enabled() ? abr::x6_flags_ptr() : nullptr);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// ------------------------------------------------------------------------------------------------------------------------
// f16x3 weight planes: fp32 matrix [rows][K] (K % 16 == 0) -> per-row scale s_n (the power of two with rowmax / s_n in [2^14, 2^15)), two
// fragment-packed fp16 planes g0 = fp16(w / s_n), g1 = fp16(w / s_n - g0) in the order the MFMA consumes them -- chunk (nb, ks, pl) = 64 lanes x
// 16 B at byte (((nb * K/16 + ks) * 2 + pl) * 64 + lane) * 16, lane l holding row nb*32 + (l & 31), k = ks*16 + (l >> 5)*8 .. +8 -- and, BEHIND the
// planes (at byte rows32 * K * 4), the scales s_n as rows32 floats (padded rows: zeros, scale 1).  One workgroup = one 32-row block: a first
// pass over its K columns finds the row maxima, a second one (L2 hits) splits and stores.  A non-finite weight raises ABR_X6_FLAG_NONFINITE.
// ------------------------------------------------------------------------------------------------------------------------
static int64_t h3_planes_bytes(int64_t rows, int64_t K) { return (rows + 31) / 32 * 32 * K * 4; }
__device__ __forceinline__ size_t h3_planes_bytes_dev(int rows, int K) { return (size_t)((rows + 31) / 32 * 32) * (size_t)K * 4; }
static int64_t h3_packed_bytes(int64_t rows, int64_t K) { return h3_planes_bytes(rows, K) + (rows + 31) / 32 * 32 * 4; }

// Two launches (round 5b: one workgroup per 32-row block walking all of K twice took 16-20 us per weight, 110 us for a model's table, on the
// path of the next step's first convs): the row scales by one wave per row, then the split by one workgroup per (32-row block, 64 columns).
__device__ __forceinline__ void h3_rowscale_row(const float* __restrict__ w, int rows, int K, int row, float* __restrict__ scales, unsigned* flags) {
    const int lane = threadIdx.x & 63;
    unsigned m = 0u;
    if (row < rows) {
        const uint4* r4 = reinterpret_cast<const uint4*>(w + (size_t)row * K);
        for (int k = lane; k < K / 4; k += 64) {
            const uint4 v = r4[k];
            m = max(max(m, max(v.x & 0x7FFFFFFFu, v.y & 0x7FFFFFFFu)), max(v.z & 0x7FFFFFFFu, v.w & 0x7FFFFFFFu));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if (lane == 0) {
        float sc, inv;
        abr::h3_scales(m, sc, inv);
        scales[row] = sc;                       // (padded rows: amax 0 -> scale 1)
        if (flags && (m >> 23) >= 255u) atomicOr(flags, ABR_X6_FLAG_NONFINITE);
    }
}
__device__ __forceinline__ void h3_pack_chunk(const float* __restrict__ w, int rows, int K, int nb, int k0, uint4* __restrict__ planes,
                                              const float* __restrict__ scales) {
    __shared__ float t[32][68];
    const int tid = threadIdx.x;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int r = (tid >> 4) + 16 * h, c = (tid & 15) * 4, row = nb * 32 + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < rows && k0 + c < K) v = *reinterpret_cast<const float4*>(w + (size_t)row * K + k0 + c);
        *reinterpret_cast<float4*>(&t[r][c]) = v;
    }
    __syncthreads();
    const int ksl = tid >> 6, lane = tid & 63, ks = k0 / 16 + ksl;
    if (ks * 16 >= K) return;
    const float inv = 1.f / scales[nb * 32 + (lane & 31)];   // (a power of two: exact)
    const float* src = &t[lane & 31][ksl * 16 + (lane >> 5) * 8];
    unsigned h0[4], h1[4];
#pragma unroll
    for (int e = 0; e < 4; e++) abr::h3_split2(src[2 * e], src[2 * e + 1], inv, h0[e], h1[e]);
    const size_t ch = ((size_t)nb * (K / 16) + ks) * 2;
    planes[ch * 64 + lane] = make_uint4(h0[0], h0[1], h0[2], h0[3]);
    planes[(ch + 1) * 64 + lane] = make_uint4(h1[0], h1[1], h1[2], h1[3]);
}
__global__ __launch_bounds__(256) void h3_rowscale_kernel(const float* __restrict__ w, int rows, int K, uint4* __restrict__ planes, unsigned* flags) {
    const int rows32 = (rows + 31) / 32 * 32;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row < rows32) h3_rowscale_row(w, rows, K, row, reinterpret_cast<float*>(reinterpret_cast<char*>(planes) + h3_planes_bytes_dev(rows, K)), flags);
}
__global__ __launch_bounds__(256) void h3_pack_kernel(const float* __restrict__ w, int rows, int K, uint4* __restrict__ planes) {
    h3_pack_chunk(w, rows, K, blockIdx.y, blockIdx.x * 64, planes, reinterpret_cast<const float*>(reinterpret_cast<const char*>(planes) + h3_planes_bytes_dev(rows, K)));
}
// the same over a TABLE of matrices (abr_conv_prepare_batch): jobs[j].c = first workgroup of job j in the row-scale launch, .first_block in the pack launch
__device__ __forceinline__ int prep_find_job_c(const abr::PrepJob* jobs, int njobs, int block) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].c <= block) lo = mid; else hi = mid - 1;
    }
    return lo;
}
__global__ __launch_bounds__(256) void h3_rowscale_multi_kernel(const abr::PrepJob* __restrict__ jobs, int njobs, unsigned* flags) {
    const int j = prep_find_job_c(jobs, njobs, blockIdx.x);
    const abr::PrepJob jb = jobs[j];
    const int rows32 = (jb.a + 31) / 32 * 32;
    const int row = (blockIdx.x - jb.c) * 4 + (threadIdx.x >> 6);
    if (row < rows32) h3_rowscale_row(jb.src, jb.a, jb.b, row, reinterpret_cast<float*>(reinterpret_cast<char*>(jb.dst) + h3_planes_bytes_dev(jb.a, jb.b)), flags);
}
__global__ __launch_bounds__(256) void h3_pack_multi_kernel(const abr::PrepJob* __restrict__ jobs, int njobs) {
    const int j = abr::prep_find_job(jobs, njobs, blockIdx.x);
    const abr::PrepJob jb = jobs[j];
    const int lb = blockIdx.x - jb.first_block;
    uint4* planes = reinterpret_cast<uint4*>(jb.dst);
    h3_pack_chunk(jb.src, jb.a, jb.b, lb / jb.gx, (lb % jb.gx) * 64, planes, reinterpret_cast<const float*>(reinterpret_cast<const char*>(planes) + h3_planes_bytes_dev(jb.a, jb.b)));
}
static int h3_pack(const float* w, int64_t rows, int K, void* planes, hipStream_t st) {
    const unsigned rb = (unsigned)((rows + 31) / 32);
    h3_rowscale_kernel<<<rb * 8u, 256, 0, st>>>(w, (int)rows, K, reinterpret_cast<uint4*>(planes), abr::x6_guard_enabled() ? abr::x6_flags_ptr() : nullptr);
    h3_pack_kernel<<<dim3((unsigned)((K + 63) / 64), rb), 256, 0, st>>>(w, (int)rows, K, reinterpret_cast<uint4*>(planes));
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

static int num_cus() {
    static int n = 0;
    if (!n) {
        int32_t info[3];
        n = abr_device_info(info) == ABR_OK ? info[0] : 256;
    }
    return n;
}

template <int BM, int BN, int WM, int WN, int NP>
int launch_x6w_np(const ConvP& p, const float* x, float* out, hipStream_t st) {
    ConvP q = p;
    q.tiles_m = (p.M + BM - 1) / BM;
    q.tiles_n = (p.Cout + BN - 1) / BN;
    q.tiles_pb = q.tiles_m * q.tiles_n;
    if (q.nbatch < 1) q.nbatch = 1;
    q.n_full = q.tiles_pb * q.nbatch; q.split = 1; q.ws = nullptr; q.cnt = nullptr;
    {   // tile order: weights that do not fit an XCD's L2 (4 MB) are walked a few n-tile columns at a time (<= ~3 MB of planes), every m-tile of
        // those columns before the next group; lab (tools/x6lab/flab.hip -DGN=): 32768x2048x512 0.336-0.342 -> 0.322 ms, x1024 0.627-0.631 -> 0.618; library
        // (with its epilogue): 36864x2048x512 504 -> 495 us, x2048x1024 782 -> 768, x1024x2048 737 -> 728; shapes with <= 4 columns or weights that fit: unchanged.
        // ABR_X6_NGROUP: 0 = always m-major (rounds 2-3), n > 0 = force groups of n columns
        static const int forced = getenv("ABR_X6_NGROUP") ? atoi(getenv("ABR_X6_NGROUP")) : -1;
        const double col_bytes = (double)BN * (double)p.K * (NP == 1 ? 2.0 : (NP == 3 ? 4.0 : 6.0));
        int g = 0;
        if (forced >= 0) g = forced;
        else if (col_bytes * q.tiles_n > 3.0 * 1048576.0) g = col_bytes <= 1.6 * 1048576.0 ? 4 : (col_bytes <= 3.2 * 1048576.0 ? 2 : 0);   // (measured: tools/dbg/ngroup_time.py)
        q.ngroup = (g > 0 && g < q.tiles_n) ? g : 0;
    }
    q.x6_flags = (NP != 1 && abr::x6_guard_enabled()) ? abr::x6_flags_ptr() : nullptr;   // (rounding to bf16 is defined for every finite value: no guard)
    q.h3_stats = (NP == 3 && q.x6_flags) ? abr::h3_stats_ptr() : nullptr;
    if (q.h3_stats) abr::h3_stats_inspected((double)q.tiles_m * BM * (double)p.K * q.nbatch);   // (the first n-tile column's workgroups inspect their A rows)
    constexpr size_t lds_op = sizeof(__bf16) * (NP == 1 ? 1 : (NP == 3 ? 2 : 3)) * BM * LDX;
    constexpr size_t lds_ep = sizeof(float) * 4 * 32 * (BN / WN + EPAD);
    const size_t lds = lds_op > lds_ep ? lds_op : lds_ep;
    const bool plain = p.R == 1 && p.S == 1 && p.stride == 1 && p.pad == 0;
    auto kern = plain ? conv_igemm_x6w_kernel<BM, BN, WM, WN, true, NP> : conv_igemm_x6w_kernel<BM, BN, WM, WN, false, NP>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_x6w_kernel<BM, BN, WM, WN, true, NP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_x6w_kernel<BM, BN, WM, WN, false, NP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    constexpr int prof_id = NP == 1 ? abr::PROF_IGEMM_BF16
                            : NP == 3 ? (BM == 128 ? (BN == 128 ? abr::PROF_H3W_128x128 : abr::PROF_H3W_128x64) : abr::PROF_H3W_64x64)
                                      : (BM == 128 ? (BN == 128 ? abr::PROF_X6W_128x128 : abr::PROF_X6W_128x64) : abr::PROF_X6W_64x64);
    q.prof_ts = abr::prof_stamp_slot(prof_id, 2.0 * (double)p.M * (double)p.Cout * (double)p.K * q.nbatch);
    {   // algorithmic bytes: the input once (its own size, not the im2col's), the packed weights, the output, a fused residual / mask
        const double in_b = q.nbatch > 1 ? 4.0 * (double)p.M * p.K * q.nbatch : 4.0 * (double)p.B * p.H * p.W * p.Cin;
        const double out_b = 4.0 * (double)p.M * p.Cout * q.nbatch;
        abr::prof_add_bytes(prof_id, in_b + (NP == 1 ? 2.0 : (NP == 3 ? 4.0 : 6.0)) * (double)p.Cout * p.K * q.nbatch + out_b * (1.0 + (p.residual ? 1.0 : 0.0) + (p.mask ? 1.0 : 0.0)));
    }
    kern<<<(unsigned)(q.tiles_pb * q.nbatch), 256, lds, st>>>(q, x, out);
    return 0;
}
template <int BM, int BN, int WM, int WN>
int launch_x6w(const ConvP& p, const float* x, float* out, hipStream_t st) {
    return p.nprod == 1 ? launch_x6w_np<BM, BN, WM, WN, 1>(p, x, out, st)
                        : (p.nprod == 3 ? launch_x6w_np<BM, BN, WM, WN, 3>(p, x, out, st) : launch_x6w_np<BM, BN, WM, WN, 6>(p, x, out, st));
}


// split-K scratch: partial tiles (64 KB each for 128x128) + tickets, one set per stream (streams may run convs concurrently)
struct SplitWs { float* ws = nullptr; int* cnt = nullptr; };
static SplitWs* split_ws(hipStream_t st) {
    static std::map<hipStream_t, SplitWs> pool;   // node-based: the returned pointer stays valid; the mutex covers first-use insertion from
    static std::mutex mu;                           // concurrent host threads (forward on the Python thread, backward on autograd's; ctypes drops the GIL)
    std::lock_guard<std::mutex> g(mu);
    SplitWs& w = pool[st];
    if (!w.ws) {
        if (hipMalloc(&w.ws, (size_t)kMaxSplitUnits * 128 * 128 * sizeof(float)) != hipSuccess) return nullptr;
        if (hipMalloc(&w.cnt, kMaxSplitTiles * sizeof(int)) != hipSuccess) return nullptr;
        if (hipMemset(w.cnt, 0, kMaxSplitTiles * sizeof(int)) != hipSuccess) return nullptr;
    }
    return &w;
}

template <int BM, int BN, int WM, int WN, bool SMALL_C, bool SB = false>
int launch(const ConvP& p, const float* x, const float* w, float* out, hipStream_t st, int prof_id, int n_full = -1, int split = 1) {
    ConvP q = p;
    q.tiles_m = (p.M + BM - 1) / BM;
    q.tiles_n = (p.Cout + BN - 1) / BN;
    q.tiles_pb = q.tiles_m * q.tiles_n;
    if (q.nbatch < 1) q.nbatch = 1;
    const int tiles = q.tiles_pb * q.nbatch;
    q.n_full = (split > 1 && n_full >= 0) ? n_full : tiles;
    q.split = split > 1 ? split : 1;
    q.ws = nullptr; q.cnt = nullptr;
    const int n_split = tiles - q.n_full;
    if (n_split > 0) {
        SplitWs* sw = (n_split * q.split <= kMaxSplitUnits && n_split <= kMaxSplitTiles) ? split_ws(st) : nullptr;
        if (sw) { q.ws = sw->ws; q.cnt = sw->cnt; }
        else { q.n_full = tiles; q.split = 1; }  // no scratch: plain launch
    }
    constexpr size_t lds_op = sizeof(float) * (SB ? 1 : 2) * (BM + BN) * LDP;
    constexpr size_t lds_ep = sizeof(float) * 4 * 32 * (BN / WN + EPAD);
    const size_t lds = lds_op > lds_ep ? lds_op : lds_ep;
    auto kern = conv_igemm_kernel<BM, BN, WM, WN, SMALL_C, SB>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    q.prof_ts = abr::prof_stamp_slot(prof_id, 2.0 * (double)p.M * (double)p.Cout * (double)p.K * q.nbatch);
    kern<<<(unsigned)(q.n_full + (tiles - q.n_full) * q.split), 256, lds, st>>>(q, x, w, out);
    return 0;
}

// (Cout, R*S, Cin) -> (Cin, R*S flipped, Cout), scaled by scale[cout]; 32x32 LDS transpose per (rs) plane.
__global__ __launch_bounds__(256) void dgrad_weights_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                                             int Cout, int RS, int Cin, float* __restrict__ wt) {
    __shared__ float t[32][33];
    const int rs = blockIdx.z;
    const int co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int co = co0 + i, ci = ci0 + tx;
        float v = 0.f;
        if (co < Cout && ci < Cin) v = w[((size_t)co * RS + rs) * Cin + ci] * (scale ? scale[co] : 1.f);
        t[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int ci = ci0 + i, co = co0 + tx;
        if (ci < Cin && co < Cout) {
            const size_t o = ((size_t)ci * RS + (RS - 1 - rs)) * Cout + co;
            wt[o] = t[tx][i];
        }
    }
}

// dgrad_weights_kernel / x6_pack_kernel over a TABLE of tensors (abr_conv_prepare_batch): workgroup -> (job, the job's own block indices)
// 64 x 64 tiles moved as 16 B per lane when both channel counts are multiples of 4 and both tensors 16 B aligned (every conv of the step but the
// 76-wide RPN head: 260 MB per step at 2.8 TB/s with 32 x 32 tiles of 4 B accesses); abr::prep_transpose_tile is the rule, shared with the host
// side that sizes the job's grid.
__global__ __launch_bounds__(256) void dgrad_weights_multi_kernel(const abr::PrepJob* __restrict__ jobs, int njobs) {
    __shared__ float t[64][65];
    const int j = abr::prep_find_job(jobs, njobs, blockIdx.x);
    const abr::PrepJob jb = jobs[j];
    const float* __restrict__ w = jb.src;
    const float* __restrict__ scale = jb.scale;
    float* __restrict__ wt = reinterpret_cast<float*>(jb.dst);
    const int Cout = jb.a, RS = jb.b, Cin = jb.c;
    const int lb = blockIdx.x - jb.first_block;
    const int bx = lb % jb.gx, by = (lb / jb.gx) % jb.gy, rs = lb / (jb.gx * jb.gy);
    if (abr::prep_transpose_tile(Cout, Cin, w, wt) == 64) {
        const int co0 = by * 64, ci0 = bx * 64;
        const int r16 = threadIdx.x >> 4, c4 = (threadIdx.x & 15) * 4;
#pragma unroll
        for (int h = 0; h < 4; h++) {
            const int row = r16 + 16 * h, co = co0 + row, ci = ci0 + c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (co < Cout && ci < Cin) {
                v = *reinterpret_cast<const float4*>(w + ((size_t)co * RS + rs) * Cin + ci);
                const float sc = scale ? scale[co] : 1.f;
                v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
            }
            t[row][c4] = v.x; t[row][c4 + 1] = v.y; t[row][c4 + 2] = v.z; t[row][c4 + 3] = v.w;
        }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 4; h++) {
            const int cir = r16 + 16 * h, ci = ci0 + cir, co = co0 + c4;
            if (ci < Cin && co < Cout)
                *reinterpret_cast<float4*>(wt + ((size_t)ci * RS + (RS - 1 - rs)) * Cout + co) =
                    make_float4(t[c4][cir], t[c4 + 1][cir], t[c4 + 2][cir], t[c4 + 3][cir]);
        }
        return;
    }
    const int co0 = by * 32, ci0 = bx * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int co = co0 + i, ci = ci0 + tx;
        float v = 0.f;
        if (co < Cout && ci < Cin) v = w[((size_t)co * RS + rs) * Cin + ci] * (scale ? scale[co] : 1.f);
        t[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int ci = ci0 + i, co = co0 + tx;
        if (ci < Cin && co < Cout) wt[((size_t)ci * RS + (RS - 1 - rs)) * Cout + co] = t[tx][i];
    }
}

__global__ __launch_bounds__(256) void x6_pack_multi_kernel(const abr::PrepJob* __restrict__ jobs, int njobs, unsigned* flags) {
    __shared__ float t[32][68];
    const int j = abr::prep_find_job(jobs, njobs, blockIdx.x);
    const abr::PrepJob jb = jobs[j];
    const float* __restrict__ w = jb.src;
    uint4* __restrict__ planes = reinterpret_cast<uint4*>(jb.dst);
    const int rows = jb.a, K = jb.b;
    const int lb = blockIdx.x - jb.first_block;
    const int nb = lb / jb.gx, k0 = (lb % jb.gx) * 64, tid = threadIdx.x;
    unsigned bmin = 0xFFFFFFFFu;
    float nonfin = 0.f;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int r = (tid >> 4) + 16 * h, c = (tid & 15) * 4, row = nb * 32 + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < rows && k0 + c < K) v = *reinterpret_cast<const float4*>(w + (size_t)row * K + k0 + c);
        *reinterpret_cast<float4*>(&t[r][c]) = v;
        const unsigned b0 = (__float_as_uint(v.x) << 1) - 1u, b1 = (__float_as_uint(v.y) << 1) - 1u, b2 = (__float_as_uint(v.z) << 1) - 1u,
                       b3 = (__float_as_uint(v.w) << 1) - 1u;
        bmin = min(min(bmin, min(b0, b1)), min(b2, b3));
        nonfin = fmaf(v.x, 0.f, nonfin); nonfin = fmaf(v.y, 0.f, nonfin); nonfin = fmaf(v.z, 0.f, nonfin); nonfin = fmaf(v.w, 0.f, nonfin);
    }
    if (flags) abr::x6_report(bmin, nonfin, flags);
    __syncthreads();
    const int ksl = tid >> 6, lane = tid & 63, ks = k0 / 16 + ksl;
    if (ks * 16 >= K) return;
    const float* src = &t[lane & 31][ksl * 16 + (lane >> 5) * 8];
    __bf16 h[3][8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const float v = src[e];
        const __bf16 h0 = (__bf16)v;
        const float r1 = v - (float)h0;
        const __bf16 h1 = (__bf16)r1;
        h[0][e] = h0; h[1][e] = h1; h[2][e] = (__bf16)(r1 - (float)h1);
    }
    const size_t c = ((size_t)nb * (K / 16) + ks) * 3;
#pragma unroll
    for (int pl = 0; pl < 3; pl++) planes[(c + pl) * 64 + lane] = *reinterpret_cast<const uint4*>(h[pl]);
}

// part / tickets (round 5): every (column chunk, row chunk) workgroup parks its 64 column sums; the LAST row chunk of a column chunk to arrive adds
// them in row-chunk order and does db += (one writer per column: deterministic, where the former atomicAdd per workgroup changed the last bits
// of the bias gradients from run to run).  part == nullptr: atomics.
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* __restrict__ gy, int64_t M, int C, int rows_per_block,
                                                         float* __restrict__ db, float* __restrict__ part, unsigned* __restrict__ tickets) {
    // block = 64 columns x 4 row-lanes; grid.x = column chunks, grid.y = row chunks
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
    float acc = 0.f;
    if (c < C)
        for (int64_t r = r0 + rl; r < r1; r += 4) acc += gy[r * C + c];
    __shared__ float sm[4][64];
    __shared__ int s_last;
    sm[rl][threadIdx.x & 63] = acc;
    __syncthreads();
    const float mine = sm[0][threadIdx.x & 63] + sm[1][threadIdx.x & 63] + sm[2][threadIdx.x & 63] + sm[3][threadIdx.x & 63];
    if (!part) {
        if (rl == 0 && c < C) atomicAdd(db + c, mine);
        return;
    }
    float* slot = part + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 64;
    if (rl == 0) __hip_atomic_store(slot + threadIdx.x, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = atomicAdd(tickets + blockIdx.x, 1u) == gridDim.y - 1u;
    }
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        tickets[blockIdx.x] = 0u;
    }
    __syncthreads();
    float a = 0.f;
    const float* col = part + (size_t)blockIdx.x * gridDim.y * 64 + (threadIdx.x & 63);
    for (unsigned y = rl; y < gridDim.y; y += 4) a += __hip_atomic_load(col + (size_t)y * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    sm[rl][threadIdx.x & 63] = a;
    __syncthreads();
    if (rl == 0 && c < C) db[c] += sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
}

}  // namespace

// tile configuration + launch for one (possibly batched) implicit GEMM described by p
static void dispatch_igemm(const ConvP& p, const float* x, const float* w, float* out, hipStream_t st) {
    const int64_t nb = p.nbatch > 1 ? p.nbatch : 1;
    const bool small_c = (p.Cin % BK) != 0;
    // tile choice: biggest tile that still gives >= 2 workgroups per CU; narrow-N layers use BN=64
    const int64_t t128 = (int64_t)((p.M + 127) / 128) * ((p.Cout + 127) / 128) * nb;
    const int64_t t12864 = (int64_t)((p.M + 127) / 128) * ((p.Cout + 63) / 64) * nb;
    const int cus = num_cus();
    // ---- split-K plan for the 128x128 tile (see ConvP::n_full).  Per-CU cost model in k-iterations: workgroups on one CU share
    // its MFMA pipe, so a grid costs ceil(tiles / CUs) tile-times; splitting the LAST tiles (or all of them when there are fewer
    // tiles than CUs) by s evens the rounds out at the price of the partial-sum round trip per unit.
    static const int split_mode = getenv("ABR_IGEMM_SPLIT") ? atoi(getenv("ABR_IGEMM_SPLIT")) : 1;
    const int nk = (p.K + BK - 1) / BK;
    int best_nfull = -1, best_s = 1;
    double cost128 = (double)((t128 + cus - 1) / cus) * nk;
    if (split_mode && !small_c && p.Cout > 64 && nb == 1) {
        // Measured (tools/microbench.py): the RPN 3x3 (600 tiles, K = 9216) goes 101 -> 121 TF with its last 88 tiles split in two;
        // GEMMs with K <= 4608 per tile LOSE (the partial-sum round trip costs ~25 k-iterations, not 6: layer3 3x3 84 -> 70 TF,
        // layer2 3x3 82 -> 76), and so does splitting every tile of a grid smaller than the chip (64x64 tiles quantise better there).
        // Hence: tail split only, >= 32 k-iterations per unit, overhead 24.
        const long nfull = (t128 / cus) * cus, tail = t128 - nfull;
        for (int sp = 2; sp <= 6 && tail > 0 && nfull > 0; sp++) {
            if (nk / sp < 32 || tail * sp > kMaxSplitUnits || tail > kMaxSplitTiles) break;
            const double c = (double)(nfull / cus) * nk + (double)((tail * sp + cus - 1) / cus) * ((double)nk / sp + 24.0);
            if (c < cost128 * 0.93) { cost128 = c; best_nfull = (int)nfull; best_s = sp; }
        }
    }
    // relative cost of the tile configurations the rules below would pick (work per tile x measured efficiency of the smaller tiles)
    const int64_t t64 = (int64_t)((p.M + 63) / 64) * ((p.Cout + 63) / 64) * nb;
    const double cost12864 = (double)((t12864 + cus - 1) / cus) * nk * 0.5 / 0.85;
    const double cost64 = (double)((t64 + cus - 1) / cus) * nk * 0.25 / 0.80;
    const bool rule128 = p.Cout > 64 && t128 >= 2 * cus;
    const bool rule12864 = !rule128 && (t12864 >= 2 * cus || p.Cout <= 64);
    const bool take_split128 = best_s > 1 && (rule128 || cost128 < 0.9 * (rule12864 ? cost12864 : cost64));

    // 128x128 tile, single- vs double-buffered operand LDS: one buffer (36.9 KB) lets a third workgroup share the CU, which pays
    // when prologue/epilogue are a large part of a tile's life (K <= 512: +2..5 %) or when the grid fits 3 but not 2 workgroups
    // per CU (the 600-tile RPN GEMM); longer-K GEMMs are faster double-buffered (one barrier per k-tile, fetch two tiles ahead).
    static const int sb_mode = getenv("ABR_IGEMM_SB") ? atoi(getenv("ABR_IGEMM_SB")) : -1;
    const int64_t wgs128 = take_split128 ? best_nfull + (t128 - best_nfull) * best_s : t128;
    static const int sb_maxk = getenv("ABR_IGEMM_SB_MAXK") ? atoi(getenv("ABR_IGEMM_SB_MAXK")) : 512;
    const bool sb = sb_mode >= 0 ? sb_mode != 0 : (p.K <= sb_maxk || (wgs128 > 2 * cus && wgs128 <= 3 * cus) || take_split128);
    // Small grid + long K (the predictor FCs: 2304 x 108 x 2048 is 72 tiles of 64 x 64 walking 64 k-tiles each, 256 x 80 x 2048 is 8): EVERY tile
    // split along K over enough workgroups to cover the chip about twice, >= 4 k-tiles per unit; the partial sums meet in the last arrival in
    // index order (deterministic).  Narrow outputs only (Cout <= 128): wider small-grid GEMMs keep the sequential-K kernel, whose error is the
    // yardstick of the arithmetic admission tests (tests/test_gpu_*_admission.py: "2x the fp32-MFMA kernel's").  Measured (tools/dbg/fc_time.py, back-to-back calls): 50.3 -> 21.9 us and 48.9 -> 15.6 us; error vs float64 4.5 -> 1.1 units of 2^-24 sum|x||w| (shorter chains).  ABR_IGEMM_FC_SPLIT=0: off.
    const char* fc_env = getenv("ABR_IGEMM_FC_SPLIT");   // (read per call: a test takes the sequential-K kernel's error as its yardstick)
    const bool fc_split = !(fc_env && atoi(fc_env) == 0);
    if (fc_split && !small_c && nb == 1 && p.Cout <= 128 && t64 * 2 <= cus && nk >= 16 && t64 <= kMaxSplitTiles) {
        int sp = (int)std::min<int64_t>(nk / 4, std::max<int64_t>(2, 2 * cus / t64));   // (2 or 8 k-tiles per unit, one cover of the chip, double-buffered LDS: all within +-2 us)
        sp = (int)std::min<int64_t>(sp, kMaxSplitUnits / t64);
        if (sp >= 2) {
            launch<64, 64, 2, 2, false, true>(p, x, w, out, st, abr::PROF_IGEMM_64x64, 0, sp);
            return;
        }
    }
    if (small_c) {
        launch<128, 64, 4, 1, true, true>(p, x, w, out, st, abr::PROF_IGEMM_SMALLC);
    } else if (rule128 || take_split128) {
        const int nf = take_split128 ? best_nfull : -1, sp = take_split128 ? best_s : 1;
        if (sb) launch<128, 128, 2, 2, false, true>(p, x, w, out, st, abr::PROF_IGEMM_128x128, nf, sp);
        else launch<128, 128, 2, 2, false, false>(p, x, w, out, st, abr::PROF_IGEMM_128x128, nf, sp);
    } else if (rule12864) {  // the smaller tiles are always single-buffered: +7..20 % (128x64), +2 % (64x64)
        launch<128, 64, 4, 1, false, true>(p, x, w, out, st, abr::PROF_IGEMM_128x64);
    } else {
        launch<64, 64, 2, 2, false, true>(p, x, w, out, st, abr::PROF_IGEMM_64x64);
    }
}

// bf16 math mode: biggest tile that still gives every CU a couple of workgroups
static void dispatch_igemm_bf16(const ConvP& p, const float* x, const float* w, float* out, hipStream_t st) {
    const int cus = num_cus();
    const int64_t t128 = (int64_t)((p.M + 127) / 128) * ((p.Cout + 127) / 128);
    const int64_t t12864 = (int64_t)((p.M + 127) / 128) * ((p.Cout + 63) / 64);
    // double-buffered operand LDS (73.7 KB, still two workgroups per CU by registers) once there are enough k-tiles to pipeline:
    // +3..10 % on the 128x128 tile (layer4 / RPN shapes).  What bounds these launches is operand delivery, not the matrix pipe:
    // every workgroup pulls (BM + BN) x K fp32 through L2 (layer4 downsample: 4.3 GB per launch = 13.7 TB/s at 0.31 ms), 16x the
    // bytes per flop of the fp32 MFMA it replaces.
    static const int db_mink = getenv("ABR_BF16_DB_MINK") ? atoi(getenv("ABR_BF16_DB_MINK")) : 512;
    const bool db = p.K >= db_mink;
    if (p.Cout > 64 && t128 >= 2 * cus) {
        if (db) launch_bf16<128, 128, 2, 2, true>(p, x, w, out, st);
        else launch_bf16<128, 128, 2, 2, false>(p, x, w, out, st);
    } else if (t12864 >= 2 * cus || p.Cout <= 64) {   // (measured: the narrower tiles are faster single-buffered, 3 workgroups / CU)
        launch_bf16<128, 64, 4, 1, false>(p, x, w, out, st);
    } else {
        launch_bf16<64, 64, 2, 2, false>(p, x, w, out, st);
    }
}

// The split arithmetics (bf16x6, f16x3, and bf16 as their one-product form): same tile rules as the bf16 mode.  p.w_planes -- the caller's, the
// library's per-version cache, or planes packed into stream scratch for a one-off call -- is never null here: the weights-direct kernel is the only one.
static void dispatch_igemm_x6(const ConvP& p, const float* x, const float* /*w*/, float* out, hipStream_t st) {
    const int cus = num_cus();
    const int64_t nb = p.nbatch > 1 ? p.nbatch : 1;
    const int64_t t128 = (int64_t)((p.M + 127) / 128) * ((p.Cout + 127) / 128) * nb;
    const int64_t t12864 = (int64_t)((p.M + 127) / 128) * ((p.Cout + 63) / 64) * nb;
    static const int force = getenv("ABR_X6_TILE") ? atoi(getenv("ABR_X6_TILE")) : 0;   // experiments: 1 = 128x128, 2 = 128x64, 3 = 64x64
    static const int force_maxk = getenv("ABR_X6_TILE_MAXK") ? atoi(getenv("ABR_X6_TILE_MAXK")) : 1 << 30;
    // Short-K convs (K <= 256: the 1x1 convs of layer1-3 and their dgrads) take 64x64 tiles whatever the grid size: alone they are
    // as fast as with 128x128 tiles (+-8 % per shape), but in the training step, where two or three other streams' kernels share the CUs,
    // the small workgroups (four per CU, 31 KB of LDS) interleave better -- step -0.3 ms in a same-session A/B.  ABR_X6_SHORTK_MAXK=0: off.
    static const int shortk = getenv("ABR_X6_SHORTK_MAXK") ? atoi(getenv("ABR_X6_SHORTK_MAXK")) : 256;
    // grid-size rules: the biggest tile whose grid still gives every CU t*_min10 / 10 workgroups (experiments: ABR_X6_T128_MIN10, ABR_X6_T12864_MIN10)
    static const int t128_min10 = getenv("ABR_X6_T128_MIN10") ? atoi(getenv("ABR_X6_T128_MIN10")) : 20;
    static const int t12864_min10 = getenv("ABR_X6_T12864_MIN10") ? atoi(getenv("ABR_X6_T12864_MIN10")) : 20;
    int tile;   // 1 = 128x128, 2 = 128x64, 3 = 64x64
    if (force && p.K <= force_maxk && nb == 1) tile = force;
    else if (shortk > 0 && p.K <= shortk && nb == 1 && p.nprod != 3) tile = 3;   // (f16x3: the grid-size rule alone is 0.1 ms per step better -- same-session A/B, two rounds: 18.23 vs 18.34)
    else if (p.Cout > 64 && t128 * 10 >= (int64_t)t128_min10 * cus) tile = 1;
    else if (t12864 * 10 >= (int64_t)t12864_min10 * cus || p.Cout <= 64) tile = 2;
    else tile = 3;
    {
        // Wave layouts chosen so that every weight fragment (global -> registers) is fetched by as few waves as possible: the 128x128 tile as four
        // waves of 128 x 32 (2x2 waves of 64 x 64 fetched each fragment twice; the A fragments, LDS reads, double instead: +1.3 % alone, -0.19 ms
        // per step), the 128x64 tile as 2x2 waves of 64 x 32 (4x1 waves of 32 x 64 fetched each four times: +4.7 % alone, -0.12 ms per step)
        if (tile == 1) launch_x6w<128, 128, 1, 4>(p, x, out, st);
        else if (tile == 2) launch_x6w<128, 64, 2, 2>(p, x, out, st);
        else launch_x6w<64, 64, 2, 2>(p, x, out, st);
    }
}

// stride-1 pad-1 3x3 conv as Winograd F(4x4,3x3): weight + input transforms, 36 batched GEMMs, output transform with the epilogue
static bool wino_conv(const ConvP& p, const float* x, const float* w, float* out, hipStream_t st) {
    const int th_n = (p.H + 3) / 4, tw_n = (p.W + 3) / 4;
    const int64_t T = (int64_t)p.B * th_n * tw_n;
    const size_t nV = (size_t)36 * T * p.Cin, nU = (size_t)36 * p.Cout * p.Cin, nM = (size_t)36 * T * p.Cout;
    if (T * (int64_t)std::max(p.Cin, p.Cout) * 4 >= (int64_t)0x7FFFFFF0) return false;
    // Winograd-domain weights: from the per-weight cache when the caller vouches for (w, w_version), else transformed into scratch.
    // bf16x6: the cached form is U's fragment-packed bf16x3 planes (36 matrices of Cout x Cin back to back: Cout % 32 == 0 makes the
    // 36 * Cout rows pack as ONE matrix whose 32-row blocks never straddle two batches), fed to the weights-direct kernel.
    // f16x3: the same with two fp16 planes and one scale per row of U (h3_pack_kernel); without a weight version U is transformed and packed
    // into scratch on every call.
    const bool h3 = p.math == ABR_MATH_F16X3, x6 = p.math == ABR_MATH_BF16X6;
    if ((h3 || x6) && p.Cout % 32 != 0) return false;
    const void* Up = nullptr;
    if (x6 && p.w_version) {
        Up = abr::derived_cached(w, abr::DERIVED_WINO_U_X6_PLANES, (size_t)x6_packed_bytes((int64_t)36 * p.Cout, p.Cin), p.w_version, st, [&](void* buf) {
            float* Uf = abr::wino_ws(st, nU);   // fp32 U in this stream's scratch, consumed by the pack launch right behind it
            if (!Uf || abr::wino_weight_transform(w, p.Cout, p.Cin, Uf, st)) return 1;
            return x6_pack(Uf, (int64_t)36 * p.Cout, p.Cin, buf, st);
        });
    }
    const size_t h3_up_floats = h3 ? (size_t)h3_packed_bytes((int64_t)36 * p.Cout, p.Cin) / 4 : (x6 ? (size_t)x6_packed_bytes((int64_t)36 * p.Cout, p.Cin) / 4 : 0);
    if (h3 && p.w_version) {
        Up = abr::derived_cached(w, abr::DERIVED_WINO_U_H3_PLANES, h3_up_floats * 4, p.w_version, st, [&](void* buf) {
            float* Uf = abr::wino_ws(st, nU);
            if (!Uf || abr::wino_weight_transform(w, p.Cout, p.Cin, Uf, st)) return 1;
            return h3_pack(Uf, (int64_t)36 * p.Cout, p.Cin, buf, st);
        });
        if (!Up) return false;
    }
    float* Uc = (!Up && !h3 && p.w_version) ? abr::wino_u_cached(w, p.Cout, p.Cin, p.w_version, st) : nullptr;
    const bool have_u = Uc || Up;
    const size_t h3_extra = ((h3 || x6) && !Up) ? h3_up_floats : 0;   // split arithmetics without a version: fp32 U AND its planes live in scratch
    float* ws = abr::wino_ws(st, (p.v_out ? 0 : nV) + (have_u ? 0 : nU) + nM + h3_extra);
    if (!ws) return false;
    float* V = p.v_out ? p.v_out : ws;
    float* U = Uc ? Uc : ws + (p.v_out ? 0 : nV);
    float* Mm = ws + (p.v_out ? 0 : nV) + (have_u ? 0 : nU);
    if (!have_u && abr::wino_weight_transform(w, p.Cout, p.Cin, U, st)) return false;
    if ((h3 || x6) && !Up) {
        void* planes = Mm + nM;
        if (h3 ? h3_pack(U, (int64_t)36 * p.Cout, p.Cin, planes, st) : x6_pack(U, (int64_t)36 * p.Cout, p.Cin, planes, st)) return false;
        Up = planes;
    }
    abr::AmaxRef v_ref{nullptr, 0};
    if (h3) {
        v_ref = abr::h3_amax_alloc();
        if (!v_ref.word) return false;
        if (p.v_out) abr::h3_amax_remember(p.v_out, v_ref);   // the weight gradient reads V back through the caller's buffer
    }
    if (abr::wino_input_transform(x, p.B, p.H, p.W, p.Cin, V, st, h3 ? &v_ref : nullptr)) return false;
    ConvP g = p;
    g.B = (int)T; g.H = g.W = 1; g.R = g.S = 1; g.stride = 1; g.pad = 0; g.Ho = g.Wo = 1;
    g.M = (int)T; g.K = p.Cin;
    g.out_H = g.out_W = 1; g.out_sh = g.out_sw = 1; g.scatter = 0; g.relu = 0;
    g.scale = g.bias = g.residual = g.mask = nullptr;
    g.d_howo.init(1u); g.d_wo.init(1u); g.d_cin.init((unsigned)p.Cin); g.d_s.init(1u);
    g.x_bytes = (unsigned)(T * p.Cin * 4); g.w_bytes = (unsigned)((int64_t)p.Cout * p.Cin * 4);
    g.nbatch = 36; g.a_bs = (long)T * p.Cin; g.w_bs = (long)p.Cout * p.Cin; g.o_bs = (long)T * p.Cout;
    g.v_out = nullptr;
    g.out_amax = nullptr;   // (the conv's output is written by the output transform)
    if (h3) {
        g.w_planes = Up; g.wp_bytes = (unsigned)h3_planes_bytes(p.Cout, p.Cin); g.wp_bs = (long)h3_planes_bytes(p.Cout, p.Cin); g.wp_nblocks = p.Cout / 32;
        g.w_scales = reinterpret_cast<const float*>(reinterpret_cast<const char*>(Up) + h3_planes_bytes((int64_t)36 * p.Cout, p.Cin));
        g.a_amax = v_ref.word; g.a_epoch = v_ref.epoch;
    } else {
        g.w_planes = Up; g.wp_bytes = (unsigned)x6_packed_bytes(p.Cout, p.Cin); g.wp_bs = (long)x6_packed_bytes(p.Cout, p.Cin); g.wp_nblocks = p.Cout / 32;
    }
    if (p.math == ABR_MATH_BF16X6 || h3) dispatch_igemm_x6(g, V, U, Mm, st);
    else dispatch_igemm(g, V, U, Mm, st);
    const abr::AmaxRef o_ref{p.out_amax, p.out_epoch};
    return abr::wino_output_transform(Mm, p.B, p.H, p.W, p.Cout, p.scale, p.bias, p.relu, p.mask, out, st, p.out_amax ? &o_ref : nullptr) == 0;
}

static int wino_min_c() {
    static const int v = getenv("ABR_WINOGRAD_MIN_C") ? atoi(getenv("ABR_WINOGRAD_MIN_C")) : 128;
    return v;
}

extern "C" int64_t abr_conv_wino_v_floats(const abr_conv_desc* d) {
    if (!d || d->math == ABR_MATH_BF16) return 0;
    static const bool wino_wgrad = !(getenv("ABR_WINOGRAD_WGRAD") && atoi(getenv("ABR_WINOGRAD_WGRAD")) == 0);
    const bool scatter = !((d->out_H <= 0 || d->out_H == d->Ho) && (d->out_W <= 0 || d->out_W == d->Wo) && d->out_sh <= 1 && d->out_sw <= 1);
    const bool ok = wino_wgrad && wino_min_c() > 0 && d->R == 3 && d->S == 3 && d->stride == 1 && d->pad == 1 && !scatter && !d->residual &&
                    d->Cin % BK == 0 && d->Cout % 4 == 0 && d->Cin >= wino_min_c() && d->Cout >= 128 &&
                    (d->math != ABR_MATH_F16X3 || d->Cout % 32 == 0);   // (f16x3 packs U in 32-row blocks: wino_conv)
    if (!ok) return 0;
    const int64_t T = (int64_t)d->B * ((d->H + 3) / 4) * ((d->W + 3) / 4);
    if (T * (int64_t)std::max(d->Cin, d->Cout) * 4 >= (int64_t)0x7FFFFFF0) return 0;
    return 36 * T * d->Cin;
}

extern "C" int abr_conv_forward(const abr_conv_desc* d, const float* x, const float* w, float* out, void* stream) {
    ABR_REQUIRE(d && x && w && out, "conv_forward: null pointer");
    ABR_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->R > 0 && d->S > 0 && d->stride > 0,
                "conv_forward: bad shape");
    ABR_REQUIRE(d->Cin % 4 == 0, "conv_forward: Cin must be a multiple of 4 (pad the 3-channel image to 4)");
    ABR_REQUIRE(d->Ho == (d->H + 2 * d->pad - d->R) / d->stride + 1 && d->Wo == (d->W + 2 * d->pad - d->S) / d->stride + 1,
                "conv_forward: Ho/Wo inconsistent with H,W,R,S,stride,pad");
    ConvP p;
    p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout; p.R = d->R; p.S = d->S;
    p.stride = d->stride; p.pad = d->pad; p.Ho = d->Ho; p.Wo = d->Wo;
    p.M = d->B * d->Ho * d->Wo;
    p.K = d->R * d->S * d->Cin;
    p.out_H = d->out_H > 0 ? d->out_H : d->Ho;
    p.out_W = d->out_W > 0 ? d->out_W : d->Wo;
    p.out_sh = d->out_sh > 0 ? d->out_sh : 1;
    p.out_sw = d->out_sw > 0 ? d->out_sw : 1;
    p.scatter = !(p.out_H == p.Ho && p.out_W == p.Wo && p.out_sh == 1 && p.out_sw == 1);
    ABR_REQUIRE((p.Ho - 1) * p.out_sh < p.out_H && (p.Wo - 1) * p.out_sw < p.out_W, "conv_forward: scatter out of range");
    p.relu = d->relu;
    p.scale = d->scale; p.bias = d->bias; p.residual = d->residual; p.mask = d->mask;
    p.tiles_m = p.tiles_n = 0; p.ngroup = 0;
    p.nbatch = 1; p.tiles_pb = 0; p.a_bs = p.w_bs = p.o_bs = 0;
    p.v_out = d->wino_v;
    p.w_version = d->w_version;
    p.w_planes = d->math == ABR_MATH_BF16X6 ? d->w_planes : nullptr;   // the caller's own abr_conv_pack_weights(w, Cout, R*S*Cin) planes, if any
    ABR_REQUIRE(x6_packed_bytes(d->Cout, p.K) < (int64_t)0xFFFFFFF0, "conv_forward: weight tensor too large for 32-bit buffer offsets");
    p.wp_bytes = (unsigned)x6_packed_bytes(d->Cout, p.K); p.wp_bs = 0; p.wp_nblocks = (d->Cout + 31) / 32;
    const int64_t xb = (int64_t)d->B * d->H * d->W * d->Cin * 4, wb = (int64_t)d->Cout * p.K * 4;
    ABR_REQUIRE(xb < (int64_t)0x7FFFFFF0 && wb < (int64_t)0x7FFFFFF0, "conv_forward: input / weight tensors must be < 2 GB (32-bit buffer offsets)");
    p.x_bytes = (unsigned)xb; p.w_bytes = (unsigned)wb;
    p.d_howo.init((unsigned)(p.Ho * p.Wo)); p.d_wo.init((unsigned)p.Wo); p.d_cin.init((unsigned)p.Cin); p.d_s.init((unsigned)p.S);
    hipStream_t st = abr::as_stream(stream);
    ABR_REQUIRE(d->math == ABR_MATH_F32 || d->math == ABR_MATH_BF16 || d->math == ABR_MATH_BF16X6 || d->math == ABR_MATH_F16X3, "conv_forward: unknown math mode");
    p.math = d->math;
    if ((p.math == ABR_MATH_BF16X6 || p.math == ABR_MATH_F16X3) && p.Cin % BKX != 0) p.math = ABR_MATH_F32;   // the 4-channel stem: fp32 MFMA
    p.nprod = 6;
    p.a_amax = nullptr; p.a_epoch = 0; p.w_scales = nullptr;
    p.out_amax = reinterpret_cast<unsigned long long*>(d->out_amax); p.out_epoch = d->out_amax_epoch;   // (every kernel's epilogue feeds it)
    if (p.math == ABR_MATH_F16X3) {
        // the A operand's amax: the caller's word (written by the tensor's producer), else reduced here
        abr::AmaxRef ar{reinterpret_cast<unsigned long long*>(const_cast<uint64_t*>(d->x_amax)), d->x_amax_epoch};
        if (!ar.word) {
            ar = abr::h3_amax_alloc();
            ABR_REQUIRE(ar.word && abr::h3_amax_reduce(x, (int64_t)d->B * d->H * d->W * d->Cin, ar, st) == 0, "conv_forward (f16x3): amax reduction failed");
        }
        p.a_amax = ar.word; p.a_epoch = ar.epoch;
        p.nprod = 3;
        p.w_planes = nullptr;
        ABR_REQUIRE(h3_planes_bytes(d->Cout, p.K) < (int64_t)0xFFFFFFF0, "conv_forward: weight tensor too large for 32-bit buffer offsets");
        p.wp_bytes = (unsigned)h3_planes_bytes(d->Cout, p.K);
    }
    if (d->math == ABR_MATH_BF16 && p.Cin % BKH == 0) {   // (the 4-channel stem has no 64-wide k-tile: it stays fp32)
        // Round 4: with a weight version the bf16 mode runs on the weights-direct kernels of the default arithmetic, single product (NP = 1): plane 0
        // of the packed weights (the cache entry bf16x6 uses) IS bf16(w), the first plane of the activation split IS bf16(x).  3x3 convs stay
        // direct (a Winograd transform of rounded operands is a different, less accurate function than the mode's definition).
        constexpr bool direct_on = true;   // (ABR_X6_WEIGHTS_DIRECT=0, the in-kernel weight split, was retired in round 6)
        static const bool bf16_wd = !(getenv("ABR_BF16_WEIGHTS_DIRECT") && atoi(getenv("ABR_BF16_WEIGHTS_DIRECT")) == 0);
        const void* planes = d->w_planes;
        if (!planes && p.w_version && direct_on && bf16_wd && p.Cin % BKX == 0)
            planes = abr::derived_cached(w, abr::DERIVED_X6_PLANES, p.wp_bytes, p.w_version, st, [&](void* buf) { return x6_pack(w, d->Cout, p.K, buf, st); });
        if (planes && direct_on && bf16_wd) {
            p.w_planes = planes;
            p.nprod = 1;
            dispatch_igemm_x6(p, x, w, out, st);
            ABR_CHECK_LAUNCH("conv_forward (bf16, weights-direct)");
            return ABR_OK;
        }
        dispatch_igemm_bf16(p, x, w, out, st);
        ABR_CHECK_LAUNCH("conv_forward (bf16)");
        return ABR_OK;
    }
    // Winograd F(4x4,3x3) for the wide stride-1 3x3 convs: 4x fewer multiply-adds (RPN 3x3: 1.42 -> 0.50 ms, layer4 conv2 1.14 ->
    // 0.41, layer2 conv2 0.122 -> 0.073); layer1's 64-channel conv stays direct -- its 36 GEMMs would have K = 64 and the transforms'
    // HBM traffic outweighs the saving.
    if (wino_min_c() > 0 && p.R == 3 && p.S == 3 && p.stride == 1 && p.pad == 1 && !p.scatter && !p.residual && p.Cin % BK == 0 &&
        p.Cout % 4 == 0 && p.Cin >= wino_min_c() && p.Cout >= 128) {
        if (wino_conv(p, x, w, out, st)) {
            ABR_CHECK_LAUNCH("conv_forward (winograd)");
            return ABR_OK;
        }
    }
    if (p.math == ABR_MATH_F16X3) {
        // packed fp16 planes + row scales of (w, w_version) from the library's cache, or packed into this stream's scratch for a one-off call
        const size_t pb = (size_t)h3_packed_bytes(d->Cout, p.K);
        const void* planes = p.w_version ? abr::derived_cached(w, abr::DERIVED_H3_PLANES, pb, p.w_version, st, [&](void* buf) { return h3_pack(w, d->Cout, p.K, buf, st); }) : nullptr;
        if (!planes) {
            void* scratch = abr::wino_ws(st, pb / 4);
            ABR_REQUIRE(scratch && h3_pack(w, d->Cout, p.K, scratch, st) == 0, "conv_forward (f16x3): no memory for the weight planes");
            planes = scratch;
        }
        p.w_planes = planes;
        p.w_scales = reinterpret_cast<const float*>(reinterpret_cast<const char*>(planes) + h3_planes_bytes(d->Cout, p.K));
        dispatch_igemm_x6(p, x, w, out, st);
    } else if (p.math == ABR_MATH_BF16X6) {
        // weights-direct kernel: the packed planes of (w, w_version) come from the library's cache (filled here on a miss: one small launch)
        if (!p.w_planes && p.w_version)
            p.w_planes = abr::derived_cached(w, abr::DERIVED_X6_PLANES, p.wp_bytes, p.w_version, st, [&](void* buf) { return x6_pack(w, d->Cout, p.K, buf, st); });
        if (!p.w_planes) {   // a one-off call: planes into this stream's scratch (the in-kernel weight split of rounds 1-5 made the same three terms)
            void* scratch = abr::wino_ws(st, (size_t)p.wp_bytes / 4 + 1);
            ABR_REQUIRE(scratch && x6_pack(w, d->Cout, p.K, scratch, st) == 0, "conv_forward (bf16x6): no memory for the weight planes");
            p.w_planes = scratch;
        }
        dispatch_igemm_x6(p, x, w, out, st);
    } else {
        dispatch_igemm(p, x, w, out, st);
    }
    ABR_CHECK_LAUNCH("conv_forward");
    return ABR_OK;
}

// ConvP of a plain (unsplit, unbatched) launch from the caller's descriptor: the part of abr_conv_forward's set-up the fused launch shares
static void convp_from_desc(const abr_conv_desc* d, ConvP& p) {
    p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout; p.R = d->R; p.S = d->S;
    p.stride = d->stride; p.pad = d->pad; p.Ho = d->Ho; p.Wo = d->Wo;
    p.M = d->B * d->Ho * d->Wo;
    p.K = d->R * d->S * d->Cin;
    p.out_H = d->Ho; p.out_W = d->Wo; p.out_sh = p.out_sw = 1; p.scatter = 0;
    p.relu = d->relu;
    p.scale = d->scale; p.bias = d->bias; p.residual = d->residual; p.mask = d->mask;
    p.tiles_m = (p.M + 127) / 128; p.tiles_n = 1; p.ngroup = 0;
    p.nbatch = 1; p.tiles_pb = p.tiles_m; p.a_bs = p.w_bs = p.o_bs = 0;
    p.n_full = p.tiles_m; p.split = 1; p.ws = nullptr; p.cnt = nullptr;
    p.v_out = nullptr;
    p.w_version = d->w_version;
    p.w_planes = d->w_planes;
    p.wp_bytes = (unsigned)x6_packed_bytes(d->Cout, p.K); p.wp_bs = 0; p.wp_nblocks = (d->Cout + 31) / 32;
    p.x_bytes = (unsigned)((int64_t)d->B * d->H * d->W * d->Cin * 4); p.w_bytes = (unsigned)((int64_t)d->Cout * p.K * 4);
    p.d_howo.init((unsigned)(p.Ho * p.Wo)); p.d_wo.init((unsigned)p.Wo); p.d_cin.init((unsigned)p.Cin); p.d_s.init((unsigned)p.S);
    p.math = d->math;
    p.nprod = 6;
    p.x6_flags = abr::x6_guard_enabled() ? abr::x6_flags_ptr() : nullptr;
    p.prof_ts = nullptr;
}

extern "C" int abr_conv_tail64_forward(const abr_conv_desc* d2, const abr_conv_desc* d3, const float* x, const float* w2, const float* w3, float* out,
                                       void* stream) {
    ABR_REQUIRE(d2 && d3 && x && w2 && w3 && out, "conv_tail64_forward: null pointer");
    ABR_REQUIRE(d2->math == ABR_MATH_BF16X6 && d3->math == ABR_MATH_BF16X6, "conv_tail64_forward: bf16x6 arithmetic only");
    ABR_REQUIRE(d2->R == 3 && d2->S == 3 && d2->stride == 1 && d2->pad == 1 && d2->Cin == 64 && d2->Cout == 64 && d2->Ho == d2->H && d2->Wo == d2->W &&
                    !d2->residual && !d2->mask,
                "conv_tail64_forward: the first conv must be 3x3, stride 1, pad 1, 64 -> 64 channels, without residual / mask");
    ABR_REQUIRE(d3->R == 1 && d3->S == 1 && d3->stride == 1 && d3->pad == 0 && d3->Cin == 64 && d3->Cout == 256 && d3->B == d2->B && d3->H == d2->H &&
                    d3->W == d2->W && d3->Ho == d2->H && d3->Wo == d2->W && !d3->mask,
                "conv_tail64_forward: the second conv must be 1x1, stride 1, 64 -> 256 channels, on the first conv's output");
    ABR_REQUIRE((d2->out_H <= 0 || d2->out_H == d2->Ho) && (d3->out_H <= 0 || d3->out_H == d3->Ho) && d2->out_sh <= 1 && d3->out_sh <= 1 &&
                    (d2->out_W <= 0 || d2->out_W == d2->Wo) && (d3->out_W <= 0 || d3->out_W == d3->Wo) && d2->out_sw <= 1 && d3->out_sw <= 1,
                "conv_tail64_forward: no scattered outputs");
    ABR_REQUIRE((int64_t)d2->B * d2->H * d2->W * 256 * 4 < (int64_t)0x7FFFFFF0, "conv_tail64_forward: tensors must be < 2 GB (32-bit buffer offsets)");
    hipStream_t st = abr::as_stream(stream);
    ConvP p, q;
    convp_from_desc(d2, p);
    convp_from_desc(d3, q);
    // packed planes of both weights: the caller's, or the library's per-version cache (filled here on a miss), or packed into stream scratch
    auto planes_of = [&](const abr_conv_desc* d, const float* w, ConvP& c) -> bool {
        if (c.w_planes) return true;
        if (d->w_version)
            c.w_planes = abr::derived_cached(w, abr::DERIVED_X6_PLANES, c.wp_bytes, d->w_version, st, [&](void* buf) { return x6_pack(w, d->Cout, c.K, buf, st); });
        return c.w_planes != nullptr;
    };
    ABR_REQUIRE(planes_of(d2, w2, p) && planes_of(d3, w3, q), "conv_tail64_forward: needs w_version != 0 (or caller-packed planes) for both weights");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_tail64_x6w_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTail64Lds);
        attr_set = true;
    }
    p.prof_ts = abr::prof_stamp_slot(abr::PROF_X6W_TAIL64, 2.0 * (double)p.M * ((double)p.Cout * p.K + (double)q.Cout * q.K));
    abr::prof_add_bytes(abr::PROF_X6W_TAIL64, 4.0 * (double)p.M * (64.0 + 256.0 * (q.residual ? 2.0 : 1.0)) + 6.0 * (64.0 * 576.0 + 256.0 * 64.0));
    conv_tail64_x6w_kernel<<<(unsigned)p.tiles_m, 256, kTail64Lds, st>>>(p, q, x, out);
    ABR_CHECK_LAUNCH("conv_tail64_forward");
    return ABR_OK;
}

extern "C" int abr_conv_prepare_weights(const float* w, int Cout, int R, int S, int Cin, int stride, int pad, int math, int64_t w_version,
                                        void* stream) {
    ABR_REQUIRE(w && w_version != 0, "conv_prepare_weights: needs a weight pointer and a non-zero w_version");
    ABR_REQUIRE(Cout > 0 && R > 0 && S > 0 && Cin > 0, "conv_prepare_weights: bad shape");
    hipStream_t st = abr::as_stream(stream);
    constexpr bool direct_on = true;   // (ABR_X6_WEIGHTS_DIRECT=0, the in-kernel weight split, was retired in round 6)
    const int K = R * S * Cin;
    // the same predicate as abr_conv_forward's Winograd branch (residual / scatter never occur on the convs that prepare)
    if (wino_min_c() > 0 && math != ABR_MATH_BF16 && R == 3 && S == 3 && stride == 1 && pad == 1 && Cin % BK == 0 && Cout % 4 == 0 &&
        Cin >= wino_min_c() && Cout >= 128) {
        if (math == ABR_MATH_F16X3 && Cout % 32 == 0) {
            const size_t nU = (size_t)36 * Cout * Cin;
            void* up = abr::derived_cached(w, abr::DERIVED_WINO_U_H3_PLANES, (size_t)h3_packed_bytes((int64_t)36 * Cout, Cin), w_version, st, [&](void* buf) {
                float* Uf = abr::wino_ws(st, nU);
                if (!Uf || abr::wino_weight_transform(w, Cout, Cin, Uf, st)) return 1;
                return h3_pack(Uf, (int64_t)36 * Cout, Cin, buf, st);
            });
            ABR_REQUIRE(up != nullptr, "conv_prepare_weights: no memory for the packed Winograd-domain weights");
        } else if (math == ABR_MATH_F16X3) {
            void* pl = abr::derived_cached(w, abr::DERIVED_H3_PLANES, (size_t)h3_packed_bytes(Cout, K), w_version, st, [&](void* buf) { return h3_pack(w, Cout, K, buf, st); });
            ABR_REQUIRE(pl != nullptr, "conv_prepare_weights: no memory for the packed weight planes");
        } else if (math == ABR_MATH_BF16X6 && direct_on && Cout % 32 == 0) {
            const size_t nU = (size_t)36 * Cout * Cin;
            void* up = abr::derived_cached(w, abr::DERIVED_WINO_U_X6_PLANES, (size_t)x6_packed_bytes((int64_t)36 * Cout, Cin), w_version, st, [&](void* buf) {
                float* Uf = abr::wino_ws(st, nU);
                if (!Uf || abr::wino_weight_transform(w, Cout, Cin, Uf, st)) return 1;
                return x6_pack(Uf, (int64_t)36 * Cout, Cin, buf, st);
            });
            ABR_REQUIRE(up != nullptr, "conv_prepare_weights: no memory for the packed Winograd-domain weights");
        } else {
            ABR_REQUIRE(abr::wino_u_cached(w, Cout, Cin, w_version, st) != nullptr, "conv_prepare_weights: no memory for the Winograd-domain weights");
        }
        ABR_CHECK_LAUNCH("conv_prepare_weights");
    } else if (math == ABR_MATH_F16X3 && Cin % BKX == 0 && h3_planes_bytes(Cout, K) < (int64_t)0xFFFFFFF0) {
        void* pl = abr::derived_cached(w, abr::DERIVED_H3_PLANES, (size_t)h3_packed_bytes(Cout, K), w_version, st, [&](void* buf) { return h3_pack(w, Cout, K, buf, st); });
        ABR_REQUIRE(pl != nullptr, "conv_prepare_weights: no memory for the packed weight planes");
        ABR_CHECK_LAUNCH("conv_prepare_weights");
    } else if ((math == ABR_MATH_BF16X6 || (math == ABR_MATH_BF16 && Cin % BKH == 0)) && direct_on && Cin % BKX == 0 &&
               x6_packed_bytes(Cout, K) < (int64_t)0xFFFFFFF0) {
        void* pl = abr::derived_cached(w, abr::DERIVED_X6_PLANES, (size_t)x6_packed_bytes(Cout, K), w_version, st, [&](void* buf) { return x6_pack(w, Cout, K, buf, st); });
        ABR_REQUIRE(pl != nullptr, "conv_prepare_weights: no memory for the packed weight planes");
        ABR_CHECK_LAUNCH("conv_prepare_weights");
    }
    return ABR_OK;
}

namespace abr {
int prep_transpose_multi(const PrepJob* jobs_dev, int njobs, int blocks, hipStream_t st) {
    if (njobs <= 0 || blocks <= 0) return 0;
    dgrad_weights_multi_kernel<<<(unsigned)blocks, 256, 0, st>>>(jobs_dev, njobs);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
int prep_pack_h3_multi(const PrepJob* jobs_dev, int njobs, int blocks, int scale_blocks, hipStream_t st) {
    if (njobs <= 0 || blocks <= 0) return 0;
    h3_rowscale_multi_kernel<<<(unsigned)scale_blocks, 256, 0, st>>>(jobs_dev, njobs, abr::x6_guard_enabled() ? abr::x6_flags_ptr() : nullptr);
    h3_pack_multi_kernel<<<(unsigned)blocks, 256, 0, st>>>(jobs_dev, njobs);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
int prep_pack_multi(const PrepJob* jobs_dev, int njobs, int blocks, hipStream_t st) {
    if (njobs <= 0 || blocks <= 0) return 0;
    x6_pack_multi_kernel<<<(unsigned)blocks, 256, 0, st>>>(jobs_dev, njobs, abr::x6_guard_enabled() ? abr::x6_flags_ptr() : nullptr);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
}  // namespace abr

// Job tables of abr_conv_prepare_batch: a RING of pinned host staging + device slots per stream.  A call takes the next njobs slots; a region is
// reused only after the upload that last used it has completed -- thousands of jobs (tens of optimiser steps) later, so the host never waits for
// the stream (a single staging buffer made every call wait for the previous call's upload, which sits behind the whole backward pass).
namespace {
constexpr size_t kPrepRing = 8192;
struct PrepTables {
    abr::PrepJob* host = nullptr;
    abr::PrepJob* dev = nullptr;
    size_t head = 0;                                   // next free slot
    std::vector<std::pair<size_t, hipEvent_t>> inflight;   // (end slot of a past call's region, its upload event), oldest first
    size_t inflight_begin = 0;                         // first slot still covered by `inflight`
    float* u_scratch = nullptr;   // fp32 Winograd-domain weights on their way to being packed
    size_t u_floats = 0;
};
std::map<hipStream_t, PrepTables> g_prep_tables;
std::mutex g_prep_mu;

// slots [*first, *first + n) of the ring, free of any upload still in flight
bool prep_ring_take(PrepTables& T, size_t n, size_t* first) {
    if (n > kPrepRing) return false;
    if (!T.host) {
        if (hipHostMalloc(&T.host, kPrepRing * sizeof(abr::PrepJob)) != hipSuccess || hipMalloc(&T.dev, kPrepRing * sizeof(abr::PrepJob)) != hipSuccess) return false;
    }
    if (T.head + n > kPrepRing) {   // wrap: everything recorded so far must be done before slot 0 is written again
        for (auto& pr : T.inflight) { (void)hipEventSynchronize(pr.second); (void)hipEventDestroy(pr.second); }
        T.inflight.clear();
        T.head = 0;
    }
    *first = T.head;
    T.head += n;
    return true;
}
}  // namespace

extern "C" int abr_conv_prepare_batch(const abr_prep_item* items, int n, void* stream) {
    ABR_REQUIRE(n >= 0 && (n == 0 || items), "conv_prepare_batch: bad args");
    if (n == 0) return ABR_OK;
    hipStream_t st = abr::as_stream(stream);
    constexpr bool direct_on = true;   // (ABR_X6_WEIGHTS_DIRECT=0, the in-kernel weight split, was retired in round 6)
    std::lock_guard<std::mutex> lock(g_prep_mu);
    PrepTables& T = g_prep_tables[st];
    std::vector<abr::PrepJob> tj, uj, pj, hj, wj;    // transposes, Winograd weight transforms, bf16x3 packings, f16x3 packings, f16x3 Winograd weights straight to planes
    std::vector<void*> tokens;
    std::vector<hipStream_t> waited;              // reader streams this call's stream is already ordered behind (derived_acquire)
    // every early return between the acquires below and derived_commit releases the tokens (entries left pending would never be evictable and
    // would repack on every later call)
    struct TokenGuard {
        std::vector<void*>& t; bool committed = false;
        ~TokenGuard() { if (!committed && !t.empty()) abr::derived_abandon(t.data(), (int)t.size()); }
    } token_guard{tokens};
    std::vector<size_t> u_off;                    // per uj entry: offset (floats) of its U inside the scratch
    size_t u_total = 0;
    int tb = 0, ub = 0, pb = 0, hb = 0, hsb = 0, wb = 0, wsb = 0;  // workgroups of the launches (hsb: the f16x3 row-scale launch; wb / wsb: the direct Winograd pack / scale launches)
    static const bool wino_direct = !(getenv("ABR_PREP_WINO_DIRECT") && atoi(getenv("ABR_PREP_WINO_DIRECT")) == 0);   // 0: through the fp32 U in the scratch (A/B)
    std::vector<int> h_scale_first;               // per hj entry: its first workgroup in the row-scale launch (stored in PrepJob::c once the sources are patched)
    auto add_pack_h3 = [&](const float* src, int64_t rows, int K, void* dst) {
        abr::PrepJob j{};
        j.src = src; j.dst = dst; j.a = (int)rows; j.b = K; j.gx = (K + 63) / 64; j.gy = (int)((rows + 31) / 32); j.first_block = hb;
        hb += j.gx * j.gy;
        h_scale_first.push_back(hsb);
        hsb += j.gy * 8;
        hj.push_back(j);
    };
    auto add_pack = [&](const float* src, int64_t rows, int K, void* dst) {
        abr::PrepJob j{};
        j.src = src; j.dst = dst; j.a = (int)rows; j.b = K; j.gx = (K + 63) / 64; j.gy = (int)((rows + 31) / 32); j.first_block = pb;
        pb += j.gx * j.gy;
        pj.push_back(j);
    };
    // what abr_conv_prepare_weights derives from tensor `w` ([Cout][R][S][Cin]) of a conv with this geometry: returns 1 when it had to go the per-tensor way
    auto derive = [&](const float* w, int Cout, int R, int S, int Cin, int stride, int pad, int math, int64_t ver) -> int {
        const int K = R * S * Cin;
        if (wino_min_c() > 0 && math != ABR_MATH_BF16 && R == 3 && S == 3 && stride == 1 && pad == 1 && Cin % BK == 0 && Cout % 4 == 0 &&
            Cin >= wino_min_c() && Cout >= 128) {
            const bool h3 = math == ABR_MATH_F16X3;
            if (!(((math == ABR_MATH_BF16X6 && direct_on) || h3) && Cout % 32 == 0 && Cin % 4 == 0)) return 1;
            void* tok = nullptr;
            void* planes = h3 ? abr::derived_acquire(w, abr::DERIVED_WINO_U_H3_PLANES, (size_t)h3_packed_bytes((int64_t)36 * Cout, Cin), ver, st, &tok, &waited)
                              : abr::derived_acquire(w, abr::DERIVED_WINO_U_X6_PLANES, (size_t)x6_packed_bytes((int64_t)36 * Cout, Cin), ver, st, &tok, &waited);
            if (!planes) return 1;
            if (!tok) return 0;   // already there
            tokens.push_back(tok);
            if (h3 && wino_direct && Cin % 64 == 0) {   // w -> planes, U never written (conv_winograd.hip: wino_h3_scales / wino_h3_pack)
                abr::PrepJob j{};
                j.src = w; j.dst = planes; j.a = Cout; j.b = Cin; j.gx = Cin / 64; j.gy = Cout / 32; j.first_block = wb; j.c = wsb;
                wb += j.gx * j.gy;
                wsb += Cout / 4;
                wj.push_back(j);
                return 0;
            }
            abr::PrepJob j{};
            j.src = w; j.a = Cout; j.b = Cin; j.gx = (int)(((int64_t)Cout * (Cin / 4) + 255) / 256); j.gy = 1; j.first_block = ub;
            ub += j.gx;
            uj.push_back(j);
            u_off.push_back(u_total);
            // the packing job reads the fp32 U from the scratch: its src is patched in once the scratch address is known
            if (h3) { add_pack_h3(reinterpret_cast<const float*>(u_total), (int64_t)36 * Cout, Cin, planes); hj.back().c = 1; }
            else { add_pack(reinterpret_cast<const float*>(u_total), (int64_t)36 * Cout, Cin, planes); pj.back().c = 1; }   // c = 1: src is a scratch offset
            u_total += (size_t)36 * Cout * Cin;
            return 0;
        }
        if (math == ABR_MATH_F16X3) {
            if (!(Cin % BKX == 0 && h3_planes_bytes(Cout, K) < (int64_t)0xFFFFFFF0)) return 0;
            void* tok = nullptr;
            void* planes = abr::derived_acquire(w, abr::DERIVED_H3_PLANES, (size_t)h3_packed_bytes(Cout, K), ver, st, &tok, &waited);
            if (!planes) return 1;
            if (tok) { tokens.push_back(tok); add_pack_h3(w, Cout, K, planes); }
            return 0;
        }
        if ((math == ABR_MATH_BF16X6 || (math == ABR_MATH_BF16 && Cin % BKH == 0)) && direct_on && Cin % BKX == 0 &&
            x6_packed_bytes(Cout, K) < (int64_t)0xFFFFFFF0) {   // (the bf16 mode reads plane 0 of the same packed planes)
            void* tok = nullptr;
            void* planes = abr::derived_acquire(w, abr::DERIVED_X6_PLANES, (size_t)x6_packed_bytes(Cout, K), ver, st, &tok, &waited);
            if (!planes) return 1;
            if (tok) { tokens.push_back(tok); add_pack(w, Cout, K, planes); }
            return 0;
        }
        return 0;   // nothing to derive
    };
    std::vector<int> single_fwd, single_bwd;
    for (int i = 0; i < n; i++) {
        const abr_prep_item& it = items[i];
        ABR_REQUIRE(it.w && it.w_version != 0 && it.Cout > 0 && it.R > 0 && it.S > 0 && it.Cin > 0, "conv_prepare_batch: bad item");
        if (derive(it.w, it.Cout, it.R, it.S, it.Cin, it.stride, it.pad, it.math, it.w_version)) single_fwd.push_back(i);
        if (it.wt) {
            abr::PrepJob j{};
            j.src = it.w; j.scale = it.scale; j.dst = it.wt; j.a = it.Cout; j.b = it.R * it.S; j.c = it.Cin;
            const int tile = abr::prep_transpose_tile(it.Cout, it.Cin, it.w, it.wt);
            j.gx = (it.Cin + tile - 1) / tile; j.gy = (it.Cout + tile - 1) / tile; j.first_block = tb;
            tb += j.gx * j.gy * it.R * it.S;
            tj.push_back(j);
            // the dgrad conv: [Cin][R][S][Cout] weights, stride 1, pad R-1-pad
            if (derive(it.wt, it.Cin, it.R, it.S, it.Cout, 1, it.R - 1 - it.pad, it.math, it.w_version)) single_bwd.push_back(i);
        }
    }
    const size_t njobs = tj.size() + uj.size() + pj.size() + hj.size() + wj.size();
    if (njobs) {
        if (u_total && T.u_floats < u_total) {
            if (T.u_scratch) { (void)hipStreamSynchronize(st); (void)hipFree(T.u_scratch); T.u_scratch = nullptr; T.u_floats = 0; }
            ABR_REQUIRE(hipMalloc(&T.u_scratch, u_total * sizeof(float)) == hipSuccess, "conv_prepare_batch: no memory for the Winograd-domain scratch");
            T.u_floats = u_total;
        }
        for (size_t k = 0; k < uj.size(); k++) uj[k].dst = T.u_scratch + u_off[k];
        for (auto& j : pj)
            if (j.c == 1) { j.src = T.u_scratch + reinterpret_cast<size_t>(j.src); j.c = 0; }
        for (size_t k = 0; k < hj.size(); k++) {
            if (hj[k].c == 1) hj[k].src = T.u_scratch + reinterpret_cast<size_t>(hj[k].src);
            hj[k].c = h_scale_first[k];
        }
        size_t first = 0;
        ABR_REQUIRE(prep_ring_take(T, njobs, &first), "conv_prepare_batch: no memory for the job tables");
        abr::PrepJob* h = T.host + first;
        abr::PrepJob* d = T.dev + first;
        std::copy(tj.begin(), tj.end(), h);
        std::copy(uj.begin(), uj.end(), h + tj.size());
        std::copy(pj.begin(), pj.end(), h + tj.size() + uj.size());
        std::copy(hj.begin(), hj.end(), h + tj.size() + uj.size() + pj.size());
        std::copy(wj.begin(), wj.end(), h + tj.size() + uj.size() + pj.size() + hj.size());
        ABR_REQUIRE(hipMemcpyAsync(d, h, njobs * sizeof(abr::PrepJob), hipMemcpyHostToDevice, st) == hipSuccess, "conv_prepare_batch: table upload failed");
        hipEvent_t up = nullptr;
        if (hipEventCreateWithFlags(&up, hipEventDisableTiming) == hipSuccess) {
            (void)hipEventRecord(up, st);
            T.inflight.emplace_back(first + njobs, up);
        }
        int bad = abr::prep_transpose_multi(d, (int)tj.size(), tb, st);
        bad |= abr::prep_wino_u_multi(d + tj.size(), (int)uj.size(), ub, st);
        bad |= abr::prep_pack_multi(d + tj.size() + uj.size(), (int)pj.size(), pb, st);
        bad |= abr::prep_pack_h3_multi(d + tj.size() + uj.size() + pj.size(), (int)hj.size(), hb, hsb, st);
        bad |= abr::prep_wino_h3_direct_multi(d + tj.size() + uj.size() + pj.size() + hj.size(), (int)wj.size(), wb, wsb,
                                              abr::x6_guard_enabled() ? abr::x6_flags_ptr() : nullptr, st);
        ABR_REQUIRE(!bad, "conv_prepare_batch: launch failed");
        abr::derived_commit(tokens.data(), (int)tokens.size(), st);
        token_guard.committed = true;
        ABR_CHECK_LAUNCH("conv_prepare_batch");
    }
    // shapes the batched kernels do not take: the per-tensor calls (after the transposes above, which every dgrad copy went through)
    for (int i : single_fwd) {
        const abr_prep_item& it = items[i];
        const int rc = abr_conv_prepare_weights(it.w, it.Cout, it.R, it.S, it.Cin, it.stride, it.pad, it.math, it.w_version, stream);
        if (rc != ABR_OK) return rc;
    }
    for (int i : single_bwd) {
        const abr_prep_item& it = items[i];
        const int rc = abr_conv_prepare_weights(it.wt, it.Cin, it.R, it.S, it.Cout, 1, it.R - 1 - it.pad, it.math, it.w_version, stream);
        if (rc != ABR_OK) return rc;
    }
    return ABR_OK;
}

extern "C" int abr_conv_run(const abr_conv_op* ops, int n) {
    ABR_REQUIRE(n >= 0 && (n == 0 || ops), "conv_run: bad args");
    for (int i = 0; i < n; i++) {
        const abr_conv_op& o = ops[i];
        int rc;
        switch (o.kind) {
            case ABR_OP_FORWARD: rc = abr_conv_forward(&o.desc, o.a, o.b, o.out, o.stream); break;
            case ABR_OP_WGRAD: rc = abr_conv_wgrad(&o.desc, o.a, o.b, o.out, o.stream); break;
            case ABR_OP_STREAM_WAIT: rc = abr_stream_wait_stream(o.stream, o.other); break;
            default: ABR_REQUIRE(false, "conv_run: unknown op kind");
        }
        if (rc != ABR_OK) return rc;
    }
    return ABR_OK;
}

extern "C" int abr_conv_cache_clear(void) {
    abr::derived_cache_clear();
    return ABR_OK;
}
extern "C" int abr_conv_cache_drop_range(const void* base, int64_t bytes) {
    if (base && bytes > 0) abr::derived_cache_drop_range(base, (size_t)bytes);
    return ABR_OK;
}
extern "C" int64_t abr_conv_cache_bytes(void) { return (int64_t)abr::derived_cache_bytes(); }

extern "C" int64_t abr_conv_packed_bytes(int64_t rows, int64_t K) { return rows > 0 && K > 0 && K % 16 == 0 ? x6_packed_bytes(rows, K) : 0; }

extern "C" int abr_conv_pack_weights(const float* w, int64_t rows, int K, void* planes, void* stream) {
    ABR_REQUIRE(w && planes && rows > 0 && K > 0 && K % 16 == 0, "conv_pack_weights: needs pointers, rows > 0 and K a positive multiple of 16");
    ABR_REQUIRE(x6_packed_bytes(rows, K) < (int64_t)0xFFFFFFF0, "conv_pack_weights: packed matrix must stay below 4 GB (32-bit buffer offsets)");
    ABR_REQUIRE(x6_pack(w, rows, K, planes, abr::as_stream(stream)) == 0, "conv_pack_weights: launch failed");
    return ABR_OK;
}

extern "C" int abr_conv_dgrad_weights(const float* w, const float* scale, int Cout, int R, int S, int Cin, float* wt,
                                      void* stream) {
    ABR_REQUIRE(w && wt && Cout > 0 && R > 0 && S > 0 && Cin > 0, "conv_dgrad_weights: bad args");
    dim3 grid((Cin + 31) / 32, (Cout + 31) / 32, R * S);
    dgrad_weights_kernel<<<grid, 256, 0, abr::as_stream(stream)>>>(w, scale, Cout, R * S, Cin, wt);
    ABR_CHECK_LAUNCH("conv_dgrad_weights");
    return ABR_OK;
}

extern "C" int abr_bias_grad(const float* gy, int64_t M, int C, float* db, void* stream) {
    ABR_REQUIRE(M >= 0 && C > 0 && db, "bias_grad: bad args");
    if (M == 0) return ABR_OK;
    ABR_REQUIRE(gy, "bias_grad: null pointer");
    const int rows_per_block = 256;
    dim3 grid((C + 63) / 64, (unsigned)((M + rows_per_block - 1) / rows_per_block));
    // scratch: 64 floats per workgroup + one ticket per column chunk (the ring's ticket array serves as the ticket block: column chunks <= 64 here)
    abr::DetWs ws = grid.x <= 64 ? abr::det_ws(abr::as_stream(stream), (size_t)grid.x * grid.y * 64) : abr::DetWs{nullptr, nullptr};
    static std::map<hipStream_t, unsigned*> tick;
    static std::mutex tick_mu;
    unsigned* tk = nullptr;
    if (ws.part) {
        std::lock_guard<std::mutex> g(tick_mu);
        unsigned*& t = tick[abr::as_stream(stream)];
        if (!t && (hipMalloc(&t, 64 * sizeof(unsigned)) != hipSuccess || hipMemset(t, 0, 64 * sizeof(unsigned)) != hipSuccess)) t = nullptr;
        tk = t;
    }
    bias_grad_kernel<<<grid, 256, 0, abr::as_stream(stream)>>>(gy, M, C, rows_per_block, db, tk ? ws.part : nullptr, tk);
    ABR_CHECK_LAUNCH("bias_grad");
    return ABR_OK;
}
