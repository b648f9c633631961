// f16x3 lab (round 5, judge's item 1): the weights-direct main loop of conv_igemm_x6w_kernel<128,128,1,4> on a TWO-term fp16 split with
// THREE products per multiply-add instead of the exact three-term bf16 split with six.
//
//   x = s * (h0 + h1),  h0 = fp16(x / s),  h1 = fp16(x / s - h0)          s = a power of two that puts the tensor's (row's) amax in [2^14, 2^15)
//   x w = s_x s_w (h0 g0 + h0 g1 + h1 g0 + [h1 g1 dropped: <= 2^-22 |x w|])
//
// Each fp16 x fp16 product is exact in the fp32 accumulator of v_mfma_f32_32x32x16_f16 (same rate as the bf16 MFMA); the representation error of an
// operand is <= 2^-22 |x| for |x| >= 2^-18 amax (h1 normal) and <= 2^-40 amax below (h1 subnormal): random-signed, it averages out under the fp32
// accumulation error of the k-sum.  The split is TWO VALU per element (v_fma_mix: f16(x * 1/s) and f16(fma(x, 1/s, -h0)), both exact before the
// final rounding) against 5.5 for the bf16 split; two planes = 4 B per element, the bytes of the fp32 value itself.
//
// Rows of the table: the library's bf16x6 loop; f16x3 with the in-kernel split; f16x3 with A as pre-split fragment-ordered planes by LDS-DMA
// (what a producer epilogue / a Winograd transform could write at no extra traffic); knock-outs (MFMAs only).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -o hlab hlab.hip && ./hlab [M N K]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int BM = 128, BN = 128, BKX = 32, LDX = 40;

// ------------------------------------------------------------------------------------------------------------------------- packing
// fp32 matrix [rows][K] -> fragment-packed bf16x3 planes (library layout)
__global__ void pack_planes_x6(const float* __restrict__ B, u32x4* __restrict__ Bp, int N, int K) {
    const int KS = K / 16;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)(N / 32) * KS * 64) return;
    const int lane = (int)(idx % 64);
    const size_t c = idx / 64;
    const int ks = (int)(c % KS), nb = (int)(c / KS);
    const float* src = B + (size_t)(nb * 32 + (lane & 31)) * K + ks * 16 + (lane >> 5) * 8;
    __bf16 h[3][8];
    for (int e = 0; e < 8; e++) {
        const float v = src[e];
        const __bf16 h0 = (__bf16)v;
        const float r1 = v - (float)h0;
        const __bf16 h1 = (__bf16)r1;
        h[0][e] = h0; h[1][e] = h1; h[2][e] = (__bf16)(r1 - (float)h1);
    }
    for (int p = 0; p < 3; p++) Bp[(c * 3 + p) * 64 + lane] = *reinterpret_cast<const u32x4*>(h[p]);
}

// power of two s with amax / s in [2^14, 2^15) (amax = 0 -> 1)
__host__ __device__ inline float h3_scale(float amax) {
    if (!(amax > 0.f)) return 1.f;
    int e;
    frexpf(amax, &e);   // amax = f * 2^e, f in [0.5, 1)
    return ldexpf(1.f, e - 15);
}
__global__ void row_scales(const float* __restrict__ B, int K, float* __restrict__ sn) {   // one wave per row
    const float* r = B + (size_t)blockIdx.x * K;
    float m = 0.f;
    for (int k = threadIdx.x; k < K; k += 64) m = fmaxf(m, fabsf(r[k]));
    for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (threadIdx.x == 0) sn[blockIdx.x] = h3_scale(m);
}
// fp32 matrix [rows][K] -> fragment-packed f16 x 2 planes: chunk(rb, ks, pl) at (((rb * KS + ks) * 2 + pl) * 64 + lane) * 16; scale per row (sn) or one (s1)
__global__ void pack_planes_h2(const float* __restrict__ B, u32x4* __restrict__ Bp, int N, int K, const float* __restrict__ sn, float s1) {
    const int KS = K / 16;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)(N / 32) * KS * 64) return;
    const int lane = (int)(idx % 64);
    const size_t c = idx / 64;
    const int ks = (int)(c % KS), nb = (int)(c / KS);
    const int row = nb * 32 + (lane & 31);
    const float inv = 1.f / (sn ? sn[row] : s1);
    const float* src = B + (size_t)row * K + ks * 16 + (lane >> 5) * 8;
    _Float16 h[2][8];
    for (int e = 0; e < 8; e++) {
        const float v = src[e] * inv;
        const _Float16 h0 = (_Float16)v;
        h[0][e] = h0; h[1][e] = (_Float16)(v - (float)h0);
    }
    for (int p = 0; p < 2; p++) Bp[(c * 2 + p) * 64 + lane] = *reinterpret_cast<const u32x4*>(h[p]);
}

// ------------------------------------------------------------------------------------------------------------------------- helpers
__device__ __forceinline__ float sub1(float x, float y) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ void split_pair_x6(const float a, const float b, unsigned& o0, unsigned& o1, unsigned& o2) {
    const f32x2v f = {a, b};
    const bf16x2 h0 = __builtin_convertvector(f, bf16x2);
    const f32x2v h0f = __builtin_convertvector(h0, f32x2v);
    const f32x2v r1 = {sub1(a, h0f.x), sub1(b, h0f.y)};
    const bf16x2 h1 = __builtin_convertvector(r1, bf16x2);
    const f32x2v h1f = __builtin_convertvector(h1, f32x2v);
    const f32x2v r2 = {sub1(r1.x, h1f.x), sub1(r1.y, h1f.y)};
    const bf16x2 h2 = __builtin_convertvector(r2, bf16x2);
    o0 = *reinterpret_cast<const unsigned*>(&h0);
    o1 = *reinterpret_cast<const unsigned*>(&h1);
    o2 = *reinterpret_cast<const unsigned*>(&h2);
}
// two-term fp16 split of a pair: four v_fma_mix (h0 = f16(x * inv), h1 = f16(fma(x, inv, -h0)))
__device__ __forceinline__ void split_pair_h3(const float a, const float b, const float inv, unsigned& o0, unsigned& o1) {
    unsigned h0, h1;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(h0) : "v"(a), "v"(inv));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h0) : "v"(b), "v"(inv));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(h1) : "v"(a), "v"(inv), "v"(h0));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(h1) : "v"(b), "v"(inv), "v"(h0));
    o0 = h0; o1 = h1;
}

__device__ __forceinline__ int tile_of_block() {
    const unsigned nblk = gridDim.x, q_ = nblk / 8, r_ = nblk % 8, xcd = blockIdx.x % 8, pos = blockIdx.x / 8;
    return (int)((xcd < r_ ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_) + pos);
}
__device__ __forceinline__ void store_c(float* C, int N, int m_base, int n_base, int lane, const f32x16 (&acc)[4], float sc) {
    const int l31 = lane & 31, lh = lane >> 5;
    for (int i = 0; i < 4; i++)
        for (int r = 0; r < 16; r++) C[(size_t)(m_base + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * N + n_base + l31] = acc[i][r] * sc;
}

// ------------------------------------------------------------------------------------------------------------------------- bf16x6 library loop
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_x6(const float* __restrict__ A, const u32x4* __restrict__ Bp,
                                                                                        float* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16* As = reinterpret_cast<__bf16*>(smem);   // [3][BM][LDX]
    const int tiles_n = N / BN;
    const int tile = tile_of_block();
    const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kq = tid & 7, srow = ((tid >> 3) & ~5) | (((tid >> 3) & 1) << 2) | ((tid >> 5) & 1);
    const int KS = K / 16;
    const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, (unsigned)((size_t)M * K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(Bp), 0, (unsigned)((size_t)N * K * 6), 0x00020000);
    unsigned ao[4];
    for (int i = 0; i < 4; i++) ao[i] = ((m0 + srow + 32 * i) * K + kq * 4) * 4u;
    const unsigned bo = (unsigned)((((size_t)(n0 / 32 + wave) * KS) * 3 * 64 + lane) * 16);
    u32x4 ra[4];
    u32x4 fbr[2][3];
    auto load_a = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; i++) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(ra_, (int)ao[i], kt * BKX * 4, 0);
    };
    auto store_a = [&]() {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            __bf16* dst = As + (srow + 32 * i) * LDX + kq * 4;
            uint2 o0, o1, o2;
            split_pair_x6(__uint_as_float(ra[i].x), __uint_as_float(ra[i].y), o0.x, o1.x, o2.x);
            split_pair_x6(__uint_as_float(ra[i].z), __uint_as_float(ra[i].w), o0.y, o1.y, o2.y);
            *reinterpret_cast<uint2*>(dst) = o0;
            *reinterpret_cast<uint2*>(dst + BM * LDX) = o1;
            *reinterpret_cast<uint2*>(dst + 2 * BM * LDX) = o2;
        }
    };
    auto load_b = [&](int kt, int u) {
        const int ks = kt * 2 + u;
#pragma unroll
        for (int p = 0; p < 3; p++) fbr[u][p] = __builtin_amdgcn_raw_buffer_load_b128(rb_, (int)bo, (ks * 3 + p) * 1024, 0);
    };
    f32x16 acc[4];
    for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    const int l31 = lane & 31, lh = lane >> 5;
    const __bf16* af = As + l31 * LDX + lh * 8;
    const int nk = K / BKX;
    auto compute_tile = [&](int kt_next) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            bf16x8 fa[4][3], fb[3];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int pl = 0; pl < 3; pl++) fa[i][pl] = *reinterpret_cast<const bf16x8*>(af + pl * BM * LDX + i * 32 * LDX + u * 16);
#pragma unroll
            for (int pl = 0; pl < 3; pl++) fb[pl] = *reinterpret_cast<const bf16x8*>(&fbr[u][pl]);
            constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; t++)
#pragma unroll
                for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][pa[t]], fb[pb[t]], acc[i], 0, 0, 0);
            if (kt_next < nk) load_b(kt_next, u);
            if (u == 0) __builtin_amdgcn_sched_barrier(0);
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
    store_c(C, N, m0, n0 + wave * 32, lane, acc, 1.f);
}

// ------------------------------------------------------------------------------------------------------------------------- f16x3, in-kernel split
// KO: 0 = full; 1 = MFMAs only (no loads / split / LDS; results wrong)
// PF: B fragments prefetched PF tiles ahead (1 = the library's distance)
// DB: 0 = one LDS buffer, two barriers per k-tile (the library's loop); 1 = two LDS buffers, ONE barrier per k-tile; 2 = A fetched TWO k-tiles
// ahead (two register sets), one LDS buffer
template <int OCC, int KO, int PF, int DB = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void k_h3(const float* __restrict__ A, const u32x4* __restrict__ Bp,
                                                                                            float* __restrict__ C, int M, int N, int K,
                                                                                            const float* __restrict__ a_scale, const float* __restrict__ sn) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* As = reinterpret_cast<_Float16*>(smem);   // [2][BM][LDX]
    const int tiles_n = N / BN;
    const int tile = tile_of_block();
    const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kq = tid & 7, srow = ((tid >> 3) & ~5) | (((tid >> 3) & 1) << 2) | ((tid >> 5) & 1);
    const int KS = K / 16;
    const float sa = *a_scale, inv = 1.f / sa;
    const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, (unsigned)((size_t)M * K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(Bp), 0, (unsigned)((size_t)N * K * 4), 0x00020000);
    unsigned ao[4];
    for (int i = 0; i < 4; i++) ao[i] = ((m0 + srow + 32 * i) * K + kq * 4) * 4u;
    const unsigned bo = (unsigned)((((size_t)(n0 / 32 + wave) * KS) * 2 * 64 + lane) * 16);
    u32x4 ra[4], rb2[4];
    u32x4 fbr[PF][2][2];   // [tile slot][step][plane]
    int abuf = 0;          // DB == 1: the LDS buffer compute_tile reads
    auto load_into = [&](u32x4 (&r)[4], int kt) {
        if (KO) return;
#pragma unroll
        for (int i = 0; i < 4; i++) r[i] = __builtin_amdgcn_raw_buffer_load_b128(ra_, (int)ao[i], kt * BKX * 4, 0);
    };
    auto load_a = [&](int kt) { load_into(ra, kt); };
    auto store_from = [&](const u32x4 (&r)[4], int buf) {
        if (KO) return;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            _Float16* dst = As + buf * (2 * BM * LDX) + (srow + 32 * i) * LDX + kq * 4;
            uint2 o0, o1;
            split_pair_h3(__uint_as_float(r[i].x), __uint_as_float(r[i].y), inv, o0.x, o1.x);
            split_pair_h3(__uint_as_float(r[i].z), __uint_as_float(r[i].w), inv, o0.y, o1.y);
            *reinterpret_cast<uint2*>(dst) = o0;
            *reinterpret_cast<uint2*>(dst + BM * LDX) = o1;
        }
    };
    auto store_a = [&]() { store_from(ra, 0); };
    auto load_b = [&]<int SL>(int kt, int u) {
        if (KO) return;
        const int ks = kt * 2 + u;
#pragma unroll
        for (int p = 0; p < 2; p++) fbr[SL][u][p] = __builtin_amdgcn_raw_buffer_load_b128(rb_, (int)bo, (ks * 2 + p) * 1024, 0);
    };
    f32x16 acc[4];
    for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    const int l31 = lane & 31, lh = lane >> 5;
    const _Float16* af = As + l31 * LDX + lh * 8;
    const int nk = K / BKX;
    if (KO) {
        for (int s = 0; s < PF; s++) for (int u = 0; u < 2; u++) for (int p = 0; p < 2; p++) fbr[s][u][p] = u32x4{(unsigned)tid, 1u, 2u, 3u};
    }
    auto compute_tile = [&]<int SL>(int kt) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            f16x8 fa[4][2], fb[2];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int pl = 0; pl < 2; pl++) {
                    if (KO) { const u32x4 t = {(unsigned)(lane + i), (unsigned)pl, 5u, 7u}; fa[i][pl] = *reinterpret_cast<const f16x8*>(&t); }
                    else fa[i][pl] = *reinterpret_cast<const f16x8*>(af + (DB == 1 ? abuf * (2 * BM * LDX) : 0) + pl * BM * LDX + i * 32 * LDX + u * 16);
                }
#pragma unroll
            for (int pl = 0; pl < 2; pl++) fb[pl] = *reinterpret_cast<const f16x8*>(&fbr[SL][u][pl]);
            constexpr int pa[3] = {1, 0, 0}, pb[3] = {0, 1, 0};   // smallest first
#pragma unroll
            for (int t = 0; t < 3; t++)
#pragma unroll
                for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][pa[t]], fb[pb[t]], acc[i], 0, 0, 0);
            if (kt + PF < nk) load_b.template operator()<SL>(kt + PF, u);
            if (u == 0) __builtin_amdgcn_sched_barrier(0);
        }
    };
    load_b.template operator()<0>(0, 0);
    load_b.template operator()<0>(0, 1);
    if (PF == 2 && nk > 1) { load_b.template operator()<PF - 1>(1, 0); load_b.template operator()<PF - 1>(1, 1); }
    load_a(0);
    store_a();
    __syncthreads();
    auto body = [&]<int SL>(int kt) {   // tile kt with the fetch / split / store of tile kt + 1
        load_a(kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        compute_tile.template operator()<SL>(kt);
        if (!KO) __syncthreads();
        store_a();
        if (!KO) __syncthreads();
    };
    if constexpr (DB == 1) {           // two LDS buffers: tile kt + 1 is stored into the other buffer right behind tile kt's MFMAs, one barrier per tile
        for (int kt = 0; kt + 1 < nk; kt++) {
            load_a(kt + 1);
            __builtin_amdgcn_sched_barrier(0);
            compute_tile.template operator()<0>(kt);
            store_from(ra, abuf ^ 1);
            __syncthreads();
            abuf ^= 1;
        }
        compute_tile.template operator()<0>(nk - 1);
    } else if constexpr (DB == 2) {    // A two tiles ahead: ra carries tile kt + 1 (requested one tile ago), rb2 is requested for kt + 2 now
        if (nk > 1) load_into(rb2, 1);
        int kt = 0;
        while (true) {
            // LDS = tile kt; rb2 = tile kt + 1 in flight
            if (kt + 2 < nk) load_into(ra, kt + 2);
            __builtin_amdgcn_sched_barrier(0);
            compute_tile.template operator()<0>(kt);
            if (kt + 1 >= nk) break;
            __syncthreads();
            store_from(rb2, 0);
            __syncthreads();
            kt++;
            if (kt + 2 < nk) load_into(rb2, kt + 2);
            __builtin_amdgcn_sched_barrier(0);
            compute_tile.template operator()<0>(kt);
            if (kt + 1 >= nk) break;
            __syncthreads();
            store_from(ra, 0);
            __syncthreads();
            kt++;
        }
    } else if (PF == 1) {
        for (int kt = 0; kt + 1 < nk; kt++) body.template operator()<0>(kt);
        compute_tile.template operator()<0>(nk - 1);
    } else {   // two k-tiles per trip: the fragment slots are compile-time
        int kt = 0;
        for (; kt + 2 < nk; kt += 2) { body.template operator()<0>(kt); body.template operator()<PF - 1>(kt + 1); }
        if (kt + 1 < nk) { body.template operator()<0>(kt); compute_tile.template operator()<PF - 1>(kt + 1); }
        else compute_tile.template operator()<0>(kt);
    }
    store_c(C, N, m0, n0 + wave * 32, lane, acc, sa * sn[n0 + wave * 32 + (lane & 31)]);
}

// ------------------------------------------------------------------------------------------------------------------------- f16x3, A planes by LDS-DMA
__device__ __forceinline__ void dma16(unsigned lds_addr, const i32x4 rsrc, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ void bload(u32x4& dst, const i32x4 rsrc, unsigned voff, unsigned soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
template <int N> __device__ __forceinline__ void bwait(u32x4& a, u32x4& b) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}

// S = 16-k steps per k-tile (2: 32 k, 16 KB per buffer; 4: 64 k, 32 KB per buffer); two LDS buffers, one barrier per k-tile
template <int S, int OCC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void k_h3_dma(const u32x4* __restrict__ Ap, const u32x4* __restrict__ Bp,
                                                                                                float* __restrict__ C, int M, int N, int K,
                                                                                                const float* __restrict__ a_scale, const float* __restrict__ sn) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];   // [2][4 mb][S][2 pl][1 KB]
    constexpr int TB = 4 * S * 2 * 1024;   // bytes per buffer
    constexpr int WB = S * 2 * 1024;       // bytes per wave (row block) and tile: contiguous in HBM and in LDS
    const int tiles_n = N / BN;
    const int tile = tile_of_block();
    const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int KS = K / 16;
    const float sa = *a_scale;
    const i32x4 rap = {(int)(unsigned)(size_t)Ap, (int)(unsigned)((size_t)Ap >> 32), (int)(unsigned)((size_t)M * K * 4), 0x00020000};
    const i32x4 rbp = {(int)(unsigned)(size_t)Bp, (int)(unsigned)((size_t)Bp >> 32), (int)(unsigned)((size_t)N * K * 4), 0x00020000};
    const unsigned bo = (unsigned)((((size_t)(n0 / 32 + wave) * KS) * 2 * 64 + lane) * 16);
    const unsigned a_voff = (unsigned)((((size_t)(m0 / 32 + wave) * KS) * 2 * 64 + lane) * 16);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    u32x4 fbr[S][2];
    auto load_b = [&](int kt, int u) {
#pragma unroll
        for (int p = 0; p < 2; p++) bload(fbr[u][p], rbp, bo, (unsigned)(((kt * S + u) * 2 + p) * 1024));
    };
    auto dma_tile = [&](int kt) {
        const unsigned dst = lds_base + (unsigned)((kt & 1) * TB + wave * WB);
#pragma unroll
        for (int c = 0; c < S * 2; c++) dma16(dst + c * 1024, rap, a_voff, (unsigned)(kt * WB + c * 1024));
    };
    f32x16 acc[4];
    for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    const int nk = K / (16 * S);
    auto step = [&](int kt, int u) {
        const unsigned char* buf = lds + (kt & 1) * TB + lane * 16;
        f16x8 fa[4][2], fb[2];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int pl = 0; pl < 2; pl++) fa[i][pl] = *reinterpret_cast<const f16x8*>(buf + ((i * S + u) * 2 + pl) * 1024);
#pragma unroll
        for (int pl = 0; pl < 2; pl++) fb[pl] = *reinterpret_cast<const f16x8*>(&fbr[u][pl]);
        constexpr int pa[3] = {1, 0, 0}, pb[3] = {0, 1, 0};
#pragma unroll
        for (int t = 0; t < 3; t++)
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][pa[t]], fb[pb[t]], acc[i], 0, 0, 0);
    };
    // queue at the top of tile kt (oldest first): DMA(kt) x 2S, B(kt, 0..S-1) x 2 each  ->  vmcnt(2S) = DMA(kt) landed
    // after DMA(kt+1) x 2S is issued, the wait for B(kt, u) leaves 2(S-1) + 2S younger operations in flight (the B loads of the other steps: this
    // tile's later ones and the next tile's earlier ones, and the DMA)
    auto tile_body = [&]<bool LAST>(int kt) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * S) : "memory");
        __syncthreads();
        if (!LAST) dma_tile(kt + 1);
#pragma unroll
        for (int u = 0; u < S; u++) {
            if (!LAST) bwait<4 * S - 2>(fbr[u][0], fbr[u][1]);
            else {
                if (u == 0) bwait<2 * (S - 1)>(fbr[u][0], fbr[u][1]);
                else if (u == 1) bwait<(S > 2 ? 2 * (S - 2) : 0)>(fbr[u][0], fbr[u][1]);
                else if (u == 2) bwait<(S > 3 ? 2 * (S - 3) : 0)>(fbr[u][0], fbr[u][1]);
                else bwait<0>(fbr[u][0], fbr[u][1]);
            }
            step(kt, u);
            if (!LAST) load_b(kt + 1, u);
        }
    };
    dma_tile(0);
#pragma unroll
    for (int u = 0; u < S; u++) load_b(0, u);
    int kt = 0;
    for (; kt + 1 < nk; kt++) tile_body.template operator()<false>(kt);
    tile_body.template operator()<true>(kt);
    store_c(C, N, m0, n0 + wave * 32, lane, acc, sa * sn[n0 + wave * 32 + (lane & 31)]);
}

// ------------------------------------------------------------------------------------------------------------------------- host
static double check(const std::vector<float>& hA, const std::vector<float>& hB, const float* dC, int M, int N, int K, double* rms_out) {
    std::vector<float> hC((size_t)M * N);
    hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0.0, sq = 0.0;
    unsigned st = 777u;
    const int T = 2048;
    for (int t = 0; t < T; t++) {
        st = st * 1664525u + 1013904223u; const int m = (st >> 8) % M;
        st = st * 1664525u + 1013904223u; const int n = (st >> 8) % N;
        double s = 0.0, sa = 0.0;
        for (int k = 0; k < K; k++) { const double p = (double)hA[(size_t)m * K + k] * (double)hB[(size_t)n * K + k]; s += p; sa += fabs(p); }
        const double e = fabs((double)hC[(size_t)m * N + n] - s) / (sa * 5.9604644775390625e-8);   // units of 2^-24 * sum|a||b|
        worst = fmax(worst, e); sq += e * e;
    }
    *rms_out = sqrt(sq / T);
    return worst;
}

struct Ctx { const float* A; const u32x4 *Ap2, *Bp3, *Bp2; float* C; int M, N, K; const float* a_scale; const float* sn; };
static float time_it(const char* name, void (*launch)(Ctx*), Ctx* c, double flops, float base) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 10; i++) launch(c);
    hipEventRecord(a);
    for (int i = 0; i < 20; i++) launch(c);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 20;
    printf("  %-58s %8.4f ms  %6.1f TF-eq   %+7.4f ms vs bf16x6\n", name, ms, flops / ms * 1e-9, base > 0 ? ms - base : 0.f);
    return ms;
}
#define GRID(c) ((c->M / BM) * (c->N / BN))
static void l_x6(Ctx* c) { k_x6<<<GRID(c), 256, sizeof(__bf16) * 3 * BM * LDX>>>(c->A, c->Bp3, c->C, c->M, c->N, c->K); }
template <int OCC, int KO, int PF, int DB = 0> static void l_h3(Ctx* c) {
    if (DB == 1) hipFuncSetAttribute(reinterpret_cast<const void*>(k_h3<OCC, KO, PF, DB>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * 2 * BM * LDX);
    k_h3<OCC, KO, PF, DB><<<GRID(c), 256, (DB == 1 ? 2 : 1) * 2 * 2 * BM * LDX>>>(c->A, c->Bp2, c->C, c->M, c->N, c->K, c->a_scale, c->sn);
}
template <int S, int OCC> static void l_dma(Ctx* c) { k_h3_dma<S, OCC><<<GRID(c), 256, 2 * 4 * S * 2 * 1024>>>(c->Ap2, c->Bp2, c->C, c->M, c->N, c->K, c->a_scale, c->sn); }

// data: 0 = uniform(-1, 1) (the other labs' operands); 1 = N(0,1)-ish x 2^U(-6, 6) per element (wide range inside every reduction)
static void fill(std::vector<float>& v, unsigned seed, int kind) {
    unsigned st = seed;
    for (auto& x : v) {
        st = st * 1664525u + 1013904223u;
        float u = ((int)(st >> 8) - (1 << 23)) * (1.0f / (1 << 23));
        if (kind == 1) { st = st * 1664525u + 1013904223u; u = ldexpf(u, (int)((st >> 10) % 13) - 6); }
        x = u;
    }
}

static void run_shape(int M, int N, int K, int kind) {
    float *A, *B, *C, *sn, *as; u32x4 *Ap2, *Bp3, *Bp2;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * N * 4);
    hipMalloc(&Ap2, (size_t)M * K * 4); hipMalloc(&Bp3, (size_t)N * K * 6); hipMalloc(&Bp2, (size_t)N * K * 4); hipMalloc(&sn, N * 4); hipMalloc(&as, 4);
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    fill(hA, 12345u, kind); fill(hB, 999u, kind);
    hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    float amax = 0.f;
    for (float v : hA) amax = fmaxf(amax, fabsf(v));
    const float sa = h3_scale(amax);
    hipMemcpy(as, &sa, 4, hipMemcpyHostToDevice);
    pack_planes_x6<<<(unsigned)(((size_t)(N / 32) * (K / 16) * 64 + 255) / 256), 256>>>(B, Bp3, N, K);
    row_scales<<<N, 64>>>(B, K, sn);
    pack_planes_h2<<<(unsigned)(((size_t)(N / 32) * (K / 16) * 64 + 255) / 256), 256>>>(B, Bp2, N, K, sn, 0.f);
    pack_planes_h2<<<(unsigned)(((size_t)(M / 32) * (K / 16) * 64 + 255) / 256), 256>>>(A, Ap2, M, K, nullptr, sa);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_h3_dma<4, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_h3_dma<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    printf("GEMM %d x %d x %d (%s operands), %d workgroups; A scale 2^%d\n", M, N, K, kind ? "wide-range" : "uniform(-1,1)", (M / BM) * (N / BN), (int)log2f(sa));
    Ctx c{A, Ap2, Bp3, Bp2, C, M, N, K, as, sn};
    const double fl = 2.0 * M * N * K;
    struct V { const char* name; void (*fn)(Ctx*); };
    const V vs[] = {{"bf16x6 library loop (3 w/SIMD)", l_x6},
                    {"f16x3 in-kernel split, 3 w/SIMD, B 1 tile ahead", l_h3<3, 0, 1>},
                    {"f16x3 in-kernel split, 4 w/SIMD, B 1 tile ahead", l_h3<4, 0, 1>},
                    {"f16x3 split, two LDS buffers, one barrier per tile, 3 w", l_h3<3, 0, 1, 1>},
                    {"f16x3 split, A two k-tiles ahead, 3 w/SIMD", l_h3<3, 0, 1, 2>},
                    {"f16x3 split, A two k-tiles ahead, 2 w/SIMD", l_h3<2, 0, 1, 2>},
                    {"f16x3 A planes by LDS-DMA, 32-k tiles, 3 w/SIMD", l_dma<2, 3>},
                    {"f16x3 A planes by LDS-DMA, 32-k tiles, 4 w/SIMD", l_dma<2, 4>},
                    {"f16x3 A planes by LDS-DMA, 64-k tiles, 2 w/SIMD", l_dma<4, 2>},
                    {"f16x3 A planes by LDS-DMA, 64-k tiles, 3 w/SIMD", l_dma<4, 3>},
                    {"f16x3 MFMAs only (3 w/SIMD; results wrong)", l_h3<3, 1, 1>}};
    constexpr int NV = sizeof(vs) / sizeof(vs[0]);
    for (int v = 0; v < NV - 1; v++) {
        hipMemset(C, 0, (size_t)M * N * 4);
        vs[v].fn(&c);
        if (hipDeviceSynchronize() != hipSuccess) { printf("  %s: launch failed\n", vs[v].name); continue; }
        double rms;
        const double w = check(hA, hB, C, M, N, K, &rms);
        printf("  %-58s error vs float64: worst %.2f, rms %.2f  (units of 2^-24 sum|a||b|, 2048 outputs)\n", vs[v].name, w, rms);
    }
    time_it("(warm-up)", l_x6, &c, fl, 0.f);
    for (int r = 0; r < 2; r++) {
        const float b = time_it(vs[0].name, vs[0].fn, &c, fl, 0.f);
        for (int v = 1; v < NV; v++) time_it(vs[v].name, vs[v].fn, &c, fl, b);
    }
    hipFree(A); hipFree(B); hipFree(C); hipFree(Ap2); hipFree(Bp3); hipFree(Bp2); hipFree(sn); hipFree(as);
}

int main(int argc, char** argv) {
    if (argc > 3) { run_shape(atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), argc > 4 ? atoi(argv[4]) : 0); return 0; }
    run_shape(32768, 2048, 1024, 0);
    run_shape(32768, 2048, 1024, 1);
    run_shape(32768, 2048, 512, 0);
    run_shape(32768, 512, 2048, 0);
    run_shape(2048 * 36, 512, 512, 0);
    run_shape(9600, 1024, 256, 0);
    return 0;
}
