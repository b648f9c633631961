// f16x3 WEIGHT-GRADIENT lab (round 5): dW[n, k] = sum_m gy[m, n] * x[m, k], plain m-major operands, 128 x 128 output tile per workgroup, split-M.
//
//   k_lib     the library's loop (conv_wgrad_x6_kernel<3>): BOTH operands fetched by all waves, split, parked in LDS as two fp16 planes
//             ([octet][j][q] image), 32-row stages, single-buffered, two barriers per stage.  Per 16-row step a wave reads 8 KB of fragments and
//             the workgroup writes 16 KB of planes for 12 MFMAs: with three workgroups per CU the LDS is as busy as the matrix pipe.
//   k_direct  gy goes through LDS (shared by the two waves of a row of the 2 x 2 wave grid), x does NOT: every wave fetches the 64 x-columns of
//             its own sub-tile straight into the registers the MFMA reads (a lane fetches two columns of eight consecutive rows: after the
//             split that IS its fragment) -- half the LDS traffic, no LDS round trip for x, double-buffered gy planes, one barrier per stage.
//
//   hipcc --offload-arch=gfx950 -O3 -o whlab whlab.hip && ./whlab [M N K splits]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void h3_split2(const float a, const float b, const float inv_s, unsigned& o0, unsigned& o1) {
    unsigned h0, h1;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(h0) : "v"(a), "v"(inv_s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h0) : "v"(b), "v"(inv_s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(h1) : "v"(a), "v"(inv_s), "v"(h0));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(h1) : "v"(b), "v"(inv_s), "v"(h0));
    o0 = h0; o1 = h1;
}
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned n) {
    const unsigned per = n / 8u, rem = n % 8u, x = bid % 8u, i = bid / 8u;
    return x < rem ? x * (per + 1) + i : rem * (per + 1) + (x - rem) * per + i;
}
__device__ __forceinline__ f16x8 as_f16x8(const u32x4 v) { return __builtin_bit_cast(f16x8, v); }

enum { KO_MFMA = 1, KO_SPLIT = 2, KO_LDS = 4, KO_LOAD = 8 };
constexpr int MRX = 32;

// ------------------------------------------------------------------------------------------------ the library's loop
template <int OCC, int PF2 = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC))) void k_lib(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ ws,
                                                                                         int M, int N, int K, int splits, int mt_per_split, float inv_g, float inv_x) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int PL = (MRX / 8) * 128 * 4;
    unsigned* Gs = reinterpret_cast<unsigned*>(smem);
    unsigned* As = Gs + 2 * PL;
    const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tiles_n = N / 128, total_tiles = tiles_n * (K / 128);
    const int split = bid / total_tiles, tile = bid % total_tiles;
    const int n0 = (tile % tiles_n) * 128, k0 = (tile / tiles_n) * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const bool is_x = __builtin_amdgcn_readfirstlane(wave) >= 2;
    const int q = tid & 31, oct = (tid >> 5) & 3;
    const int mt0 = split * mt_per_split, mt1 = min(mt0 + mt_per_split, (M + MRX - 1) / MRX);
    const __amdgpu_buffer_rsrc_t rsrc = is_x ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (unsigned)((size_t)M * K * 4), 0x00020000)
                                             : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gy), 0, (unsigned)((size_t)M * N * 4), 0x00020000);
    const int col = (is_x ? k0 : n0) + q * 4, ld = is_x ? K : N;
    const unsigned voff = (unsigned)(oct * 8 * ld + col) * 4u;
    const float inv = is_x ? inv_x : inv_g;
    u32x4 rr[8], rb[8];
    auto load_into = [&](u32x4 (&r)[8], int mt) {
        const unsigned base = mt < mt1 ? voff + (unsigned)(mt * MRX * ld) * 4u : 0x80000000u;
#pragma unroll
        for (int i = 0; i < 8; i++) r[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(base + (unsigned)(i * ld) * 4u), 0, 0);
    };
    auto load_tile = [&](int mt) { load_into(rr, mt); };
    unsigned* const st_base = (is_x ? As : Gs) + (oct * 4 * 32 + q) * 4;
    auto store_from = [&](const u32x4 (&r)[8]) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            u32x4 o0, o1;
#pragma unroll
            for (int pp = 0; pp < 4; pp++) {
                unsigned a0, a1;
                h3_split2(__uint_as_float(r[2 * pp][j]), __uint_as_float(r[2 * pp + 1][j]), inv, a0, a1);
                o0[pp] = a0; o1[pp] = a1;
            }
            *reinterpret_cast<u32x4*>(st_base + j * 128) = o0;
            *reinterpret_cast<u32x4*>(st_base + j * 128 + PL) = o1;
        }
    };
    auto store_tile = [&]() { store_from(rr); };
    f32x16 acc[2][2];
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++)
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    const int l31 = lane & 31, lh = lane >> 5;
    const unsigned* const g_rd = Gs + ((lh * 4 + 2 * wm) * 32 + l31) * 4;
    const unsigned* const a_rd = As + ((lh * 4 + 2 * wn) * 32 + l31) * 4;
    auto compute_tile = [&]() {
#pragma unroll
        for (int s = 0; s < MRX / 16; s++) {
            u32x4 G[2][2], A[2][2];
#pragma unroll
            for (int pl = 0; pl < 2; pl++)
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    G[t][pl] = *reinterpret_cast<const u32x4*>(g_rd + pl * PL + (s * 8 + t) * 128);
                    A[t][pl] = *reinterpret_cast<const u32x4*>(a_rd + pl * PL + (s * 8 + t) * 128);
                }
            constexpr int pg[3] = {1, 0, 0}, pa[3] = {0, 1, 0};
#pragma unroll
            for (int t = 0; t < 3; t++)
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(G[i][pg[t]]), as_f16x8(A[j][pa[t]]), acc[i][j], 0, 0, 0);
        }
    };
    if constexpr (PF2) {
        // fetch TWO stages ahead: stage t + 2 is requested while stage t is multiplied; two register sets, the loop unrolled by two
        if (mt0 < mt1) {
            load_into(rr, mt0);
            load_into(rb, mt0 + 1);
            store_from(rr);
            __syncthreads();
            int mt = mt0;
            while (true) {
                // LDS holds stage mt; rb holds (in flight) stage mt + 1
                load_into(rr, mt + 2);
                __builtin_amdgcn_sched_barrier(0);
                compute_tile();
                if (mt + 1 >= mt1) break;
                __syncthreads();
                store_from(rb);
                __syncthreads();
                mt++;
                load_into(rb, mt + 2);
                __builtin_amdgcn_sched_barrier(0);
                compute_tile();
                if (mt + 1 >= mt1) break;
                __syncthreads();
                store_from(rr);
                __syncthreads();
                mt++;
            }
        }
    } else
    if (mt0 < mt1) {
        load_tile(mt0);
        store_tile();
        __syncthreads();
        for (int mt = mt0; mt + 1 < mt1; mt++) {
            load_tile(mt + 1);
            __builtin_amdgcn_sched_barrier(0);
            compute_tile();
            __syncthreads();
            store_tile();
            __syncthreads();
        }
        compute_tile();
    }
    float4* dst = reinterpret_cast<float4*>(reinterpret_cast<char*>(ws) + ((size_t)tile * splits + split) * 65536) + tid;
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const f32x16& a = acc[t >> 1][t & 1];
            dst[(t * 4 + c) * 256] = make_float4(a[4 * c], a[4 * c + 1], a[4 * c + 2], a[4 * c + 3]);
        }
}

// ------------------------------------------------------------------------------------------------ gy through LDS, x straight into registers
// LDS: [2 buffers][2 planes][4 octets][2 e][64 cp] chunks of 16 B (column = 2 cp + e): 8 KB per plane, 32 KB in all.
constexpr int DPL = 4 * 2 * 64 * 4;   // dwords per plane
constexpr int DBUF = 2 * DPL;         // dwords per buffer

template <int OCC, int KO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC))) void k_direct(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ ws,
                                                                                           int M, int N, int K, int splits, int mt_per_split, float inv_g, float inv_x) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned* const lds = reinterpret_cast<unsigned*>(smem);
    const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tiles_n = N / 128, total_tiles = tiles_n * (K / 128);
    const int split = bid / total_tiles, tile = bid % total_tiles;
    const int n0 = (tile % tiles_n) * 128, k0 = (tile / tiles_n) * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const int mt0 = split * mt_per_split, mt1 = min(mt0 + mt_per_split, (M + MRX - 1) / MRX);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gy), 0, (unsigned)((size_t)M * N * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (unsigned)((size_t)M * K * 4), 0x00020000);
    // gy loader: wave w = octet w of the stage, lane = column pair
    const unsigned g_voff = (unsigned)((wave * 8) * N + n0 + 2 * lane) * 4u;
    unsigned* const g_st = lds + (wave * 2 * 64 + lane) * 4;          // chunk (octet = wave, e, cp = lane) at + e * 256 dwords, plane at + DPL
    // x fragments of this wave: columns k0 + 64 wn + 2 l31 + e, rows 16 s + 8 lh + i of the stage
    const unsigned a_voff = (unsigned)((lh * 8) * K + k0 + 64 * wn + 2 * l31) * 4u;
    // gy fragment reads: chunk (octet 2 s + lh, e = i, cp = 32 wm + l31)
    const unsigned* const g_rd = lds + (lh * 2 * 64 + 32 * wm + l31) * 4;
    u32x2 gr[8], ar[2][8];
    u32x4 A[2][2][2];   // [k-step][sub-tile j = e][plane]
    auto load_g = [&](int mt) {
        const unsigned base = g_voff + (unsigned)(mt * MRX * N) * 4u;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if constexpr (KO & KO_LOAD) { gr[i].x = base + i; gr[i].y = base ^ i; }
            else gr[i] = __builtin_amdgcn_raw_buffer_load_b64(rg, (int)(base + (unsigned)(i * N) * 4u), 0, 0);
        }
    };
    auto load_a = [&](int mt) {
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const unsigned base = a_voff + (unsigned)((mt * MRX + s * 16) * K) * 4u;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if constexpr (KO & KO_LOAD) { ar[s][i].x = base + i; ar[s][i].y = base ^ i; }
                else ar[s][i] = __builtin_amdgcn_raw_buffer_load_b64(rx, (int)(base + (unsigned)(i * K) * 4u), 0, 0);
            }
        }
    };
    auto store_g = [&](int buf) {
#pragma unroll
        for (int e = 0; e < 2; e++) {
            u32x4 o0, o1;
#pragma unroll
            for (int pp = 0; pp < 4; pp++) {
                if constexpr (KO & KO_SPLIT) { o0[pp] = gr[2 * pp][e]; o1[pp] = gr[2 * pp + 1][e]; }
                else {
                    unsigned a0, a1;
                    h3_split2(__uint_as_float(gr[2 * pp][e]), __uint_as_float(gr[2 * pp + 1][e]), inv_g, a0, a1);
                    o0[pp] = a0; o1[pp] = a1;
                }
            }
            if constexpr (KO & KO_LDS) { asm volatile("" ::"v"(o0), "v"(o1)); }
            else {
                *reinterpret_cast<u32x4*>(g_st + buf * DBUF + e * 256) = o0;
                *reinterpret_cast<u32x4*>(g_st + buf * DBUF + e * 256 + DPL) = o1;
            }
        }
    };
    auto split_a = [&]() {
#pragma unroll
        for (int s = 0; s < 2; s++)
#pragma unroll
            for (int e = 0; e < 2; e++)
#pragma unroll
                for (int pp = 0; pp < 4; pp++) {
                    if constexpr (KO & KO_SPLIT) { A[s][e][0][pp] = ar[s][2 * pp][e]; A[s][e][1][pp] = ar[s][2 * pp + 1][e]; }
                    else {
                        unsigned a0, a1;
                        h3_split2(__uint_as_float(ar[s][2 * pp][e]), __uint_as_float(ar[s][2 * pp + 1][e]), inv_x, a0, a1);
                        A[s][e][0][pp] = a0; A[s][e][1][pp] = a1;
                    }
                }
    };
    f32x16 acc[2][2];
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++)
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    auto compute = [&](int buf) {
#pragma unroll
        for (int s = 0; s < 2; s++) {
            u32x4 G[2][2];
#pragma unroll
            for (int pl = 0; pl < 2; pl++)
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    if constexpr (KO & KO_LDS) G[e][pl] = A[s][e][pl];
                    else G[e][pl] = *reinterpret_cast<const u32x4*>(g_rd + buf * DBUF + pl * DPL + (s * 4 + e) * 256);
                }
            if constexpr (KO & KO_MFMA) {
                asm volatile("" ::"v"(G[0][0]), "v"(G[0][1]), "v"(G[1][0]), "v"(G[1][1]));
            } else {
                constexpr int pg[3] = {1, 0, 0}, pa[3] = {0, 1, 0};
#pragma unroll
                for (int t = 0; t < 3; t++)
#pragma unroll
                    for (int i = 0; i < 2; i++)
#pragma unroll
                        for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(G[i][pg[t]]), as_f16x8(A[s][j][pa[t]]), acc[i][j], 0, 0, 0);
            }
        }
    };
    if (mt0 < mt1) {
        load_g(mt0);
        load_a(mt0);
        store_g(0);
        split_a();
        __syncthreads();
        int buf = 0;
        for (int mt = mt0; mt + 1 < mt1; mt++) {
            load_g(mt + 1);
            load_a(mt + 1);
            __builtin_amdgcn_sched_barrier(0);
            compute(buf);
            store_g(buf ^ 1);
            split_a();
            __syncthreads();
            buf ^= 1;
        }
        compute(buf);
    }
    float4* dst = reinterpret_cast<float4*>(reinterpret_cast<char*>(ws) + ((size_t)tile * splits + split) * 65536) + tid;
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const f32x16& a = acc[t >> 1][t & 1];
            dst[(t * 4 + c) * 256] = make_float4(a[4 * c], a[4 * c + 1], a[4 * c + 2], a[4 * c + 3]);
        }
}


// ------------------------------------------------------------------------------------------------ 256 (gy columns) x 128 (x columns) tile
// Both operands through LDS as in the library loop, but a workgroup owns 256 x 128 of dW: a wave's sub-tile is 128 x 64 (8 accumulators), every
// x element is used by twice as many MFMAs: 43 flop per fetched byte instead of 32, six fragment reads per 24 MFMAs instead of four per 12.
// gy image [4 octets][4 j][64 q] (column = 4 q + j, b128 fetches), x image [4 octets][2 e][64 cp] (column = 2 cp + e, b64 fetches); 48 KB.
constexpr int BGP = 4 * 4 * 64 * 4;   // dwords per gy plane (16 KB)
constexpr int BXP = 4 * 2 * 64 * 4;   // dwords per x plane (8 KB)
template <int OCC, int KO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC))) void k_big(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ ws,
                                                                                        int M, int N, int K, int splits, int mt_per_split, float inv_g, float inv_x) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned* const Gs = reinterpret_cast<unsigned*>(smem);   // [2 planes][BGP]
    unsigned* const Xs = Gs + 2 * BGP;                         // [2 planes][BXP]
    const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tiles_n = N / 256, total_tiles = tiles_n * (K / 128);
    const int split = bid / total_tiles, tile = bid % total_tiles;
    const int n0 = (tile % tiles_n) * 256, k0 = (tile / tiles_n) * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const int mt0 = split * mt_per_split, mt1 = min(mt0 + mt_per_split, (M + MRX - 1) / MRX);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gy), 0, (unsigned)((size_t)M * N * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (unsigned)((size_t)M * K * 4), 0x00020000);
    // loaders: wave w = octet w of the stage; lane = gy column group (4 columns, b128) and x column pair (b64)
    const unsigned g_voff = (unsigned)((wave * 8) * N + n0 + 4 * lane) * 4u;
    const unsigned x_voff = (unsigned)((wave * 8) * K + k0 + 2 * lane) * 4u;
    unsigned* const g_st = Gs + (wave * 4 * 64 + lane) * 4;   // chunk (octet, j, q = lane) at + j * 256 dwords
    unsigned* const x_st = Xs + (wave * 2 * 64 + lane) * 4;   // chunk (octet, e, cp = lane) at + e * 256 dwords
    const unsigned* const g_rd = Gs + (lh * 4 * 64 + 32 * wm + l31) * 4;   // + (s * 8 + i) * 256: chunk (octet 2 s + lh, j = i, q = 32 wm + l31)
    const unsigned* const x_rd = Xs + (lh * 2 * 64 + 32 * wn + l31) * 4;   // + (s * 4 + e) * 256
    u32x4 gr[8];
    u32x2 xr[8];
    auto load_tile = [&](int mt) {
        const unsigned gb = g_voff + (unsigned)(mt * MRX * N) * 4u, xb = x_voff + (unsigned)(mt * MRX * K) * 4u;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if constexpr (KO & KO_LOAD) { gr[i] = u32x4{gb, gb + i, gb ^ i, gb}; xr[i] = u32x2{xb + i, xb ^ i}; }
            else {
                gr[i] = __builtin_amdgcn_raw_buffer_load_b128(rg, (int)(gb + (unsigned)(i * N) * 4u), 0, 0);
                xr[i] = __builtin_amdgcn_raw_buffer_load_b64(rx, (int)(xb + (unsigned)(i * K) * 4u), 0, 0);
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            u32x4 o0, o1;
#pragma unroll
            for (int pp = 0; pp < 4; pp++) {
                unsigned a0, a1;
                h3_split2(__uint_as_float(gr[2 * pp][j]), __uint_as_float(gr[2 * pp + 1][j]), inv_g, a0, a1);
                o0[pp] = a0; o1[pp] = a1;
            }
            *reinterpret_cast<u32x4*>(g_st + j * 256) = o0;
            *reinterpret_cast<u32x4*>(g_st + j * 256 + BGP) = o1;
        }
#pragma unroll
        for (int e = 0; e < 2; e++) {
            u32x4 o0, o1;
#pragma unroll
            for (int pp = 0; pp < 4; pp++) {
                unsigned a0, a1;
                h3_split2(__uint_as_float(xr[2 * pp][e]), __uint_as_float(xr[2 * pp + 1][e]), inv_x, a0, a1);
                o0[pp] = a0; o1[pp] = a1;
            }
            *reinterpret_cast<u32x4*>(x_st + e * 256) = o0;
            *reinterpret_cast<u32x4*>(x_st + e * 256 + BXP) = o1;
        }
    };
    f32x16 acc[4][2];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 2; j++)
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    auto compute_tile = [&]() {
#pragma unroll
        for (int s = 0; s < 2; s++) {
            u32x4 G[4][2], A[2][2];
#pragma unroll
            for (int pl = 0; pl < 2; pl++) {
#pragma unroll
                for (int i = 0; i < 4; i++) G[i][pl] = *reinterpret_cast<const u32x4*>(g_rd + pl * BGP + (s * 8 + i) * 256);
#pragma unroll
                for (int e = 0; e < 2; e++) A[e][pl] = *reinterpret_cast<const u32x4*>(x_rd + pl * BXP + (s * 4 + e) * 256);
            }
            constexpr int pg[3] = {1, 0, 0}, pa[3] = {0, 1, 0};
#pragma unroll
            for (int t = 0; t < 3; t++)
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(G[i][pg[t]]), as_f16x8(A[j][pa[t]]), acc[i][j], 0, 0, 0);
        }
    };
    if (mt0 < mt1) {
        load_tile(mt0);
        store_tile();
        __syncthreads();
        for (int mt = mt0; mt + 1 < mt1; mt++) {
            load_tile(mt + 1);
            __builtin_amdgcn_sched_barrier(0);
            compute_tile();
            __syncthreads();
            store_tile();
            __syncthreads();
        }
        compute_tile();
    }
    float4* dst = reinterpret_cast<float4*>(reinterpret_cast<char*>(ws) + ((size_t)tile * splits + split) * 131072) + tid;
#pragma unroll
    for (int t = 0; t < 8; t++)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const f32x16& a = acc[t >> 1][t & 1];
            dst[(t * 4 + c) * 256] = make_float4(a[4 * c], a[4 * c + 1], a[4 * c + 2], a[4 * c + 3]);
        }
}

// ------------------------------------------------------------------------------------------------ host
template <typename F>
static float time_it(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 10; i++) launch();
    hipDeviceSynchronize();
    const int reps = 20;
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

// element (n, k) of the tile from the parked partial tiles: thread-major layout, sub-tile (tm, tn) of wave (wm, wn), register r, lane l.
// map4 = the library image's interleave (n = 4 i + 2 wm + tm), else (n = 64 wm + 2 i + tm)
static double tile_elem(const std::vector<float>& ws, size_t tile, int splits, int n, int k, bool map4) {
    int wm, tm, i, wn, tn, j;
    if (map4) { i = n / 4; wm = (n % 4) / 2; tm = n % 2; j = k / 4; wn = (k % 4) / 2; tn = k % 2; }
    else { wm = n / 64; i = (n % 64) / 2; tm = n % 2; wn = k / 64; j = (k % 64) / 2; tn = k % 2; }
    // C layout of the 32x32 MFMA: col = lane & 31 (= j), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (= i)
    const int lane = j + 32 * ((i >> 2) & 1), r = (i & 3) + 4 * (i >> 3);
    const int tid = (wm * 2 + wn) * 64 + lane, t = tm * 2 + tn, c = r / 4, comp = r % 4;
    double s = 0;
    for (int sp = 0; sp < splits; sp++) s += ws[(tile * splits + sp) * 16384 + ((size_t)(t * 4 + c) * 256 + tid) * 4 + comp];
    return s;
}

// the same for k_big's 256 x 128 tile: n = 4 q + j (q = 32 wm + i, sub-tile tm = j), k = 2 cp + e (cp = 32 wn + l, tn = e)
static double big_elem(const std::vector<float>& ws, size_t tile, int splits, int n, int k) {
    const int q = n / 4, tm = n % 4, wm = q / 32, i = q % 32, cp = k / 2, tn = k % 2, wn = cp / 32, j = cp % 32;
    const int lane = j + 32 * ((i >> 2) & 1), r = (i & 3) + 4 * (i >> 3);
    const int tid = (wm * 2 + wn) * 64 + lane, t = tm * 2 + tn, c = r / 4, comp = r % 4;
    double s = 0;
    for (int sp = 0; sp < splits; sp++) s += ws[(tile * splits + sp) * 32768 + ((size_t)(t * 4 + c) * 256 + tid) * 4 + comp];
    return s;
}

static void run_shape(int M, int N, int K, int splits) {
    const int tiles = (N / 128) * (K / 128), m_tiles = (M + MRX - 1) / MRX, mtps = (m_tiles + splits - 1) / splits;
    printf("dW[%d x %d] over M = %d, %d splits = %d workgroups of %d stages\n", N, K, M, splits, tiles * splits, mtps);
    float *x, *gy, *ws;
    hipMalloc(&x, (size_t)M * K * 4); hipMalloc(&gy, (size_t)M * N * 4); hipMalloc(&ws, (size_t)tiles * splits * 65536 * 2);
    std::vector<float> hx((size_t)M * K), hg((size_t)M * N);
    srand(1);
    for (auto& v : hx) v = (float)rand() / RAND_MAX - 0.5f;
    for (auto& v : hg) v = ((float)rand() / RAND_MAX - 0.5f) * 3.f;
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(gy, hg.data(), hg.size() * 4, hipMemcpyHostToDevice);
    // amax 0.5 / 1.5 -> scales 2^-15 / 2^-14 (amax / s in [2^14, 2^15))
    const float s_x = ldexpf(1.f, -15), s_g = ldexpf(1.f, -14), inv_x = 1.f / s_x, inv_g = 1.f / s_g;
    const size_t lds_lib = sizeof(unsigned) * 4 * (MRX / 8) * 128 * 4, lds_dir = sizeof(unsigned) * 2 * DBUF;
    auto check = [&](const char* name, bool map4) {
        std::vector<float> hw((size_t)tiles * splits * 16384);
        hipMemcpy(hw.data(), ws, hw.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0;
        srand(7);
        for (int it = 0; it < 256; it++) {
            const int n = rand() % N, k = rand() % K;
            double ref = 0, sc = 0;
            for (int m = 0; m < M; m++) { ref += (double)hg[(size_t)m * N + n] * hx[(size_t)m * K + k]; sc += fabs((double)hg[(size_t)m * N + n] * hx[(size_t)m * K + k]); }
            const size_t tile = (size_t)(k / 128) * (N / 128) + n / 128;
            const double got = tile_elem(hw, tile, splits, n % 128, k % 128, map4) * s_x * s_g;
            worst = fmax(worst, fabs(got - ref) / sc);
        }
        printf("  %-44s error vs float64: worst %.2f units of 2^-24 sum|g||x| (256 outputs)\n", name, worst * 16777216.0);
    };
    const int btiles = (N / 256) * (K / 128);
    const size_t lds_big = sizeof(unsigned) * 2 * (BGP + BXP);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_big<2, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_big);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_big<2, KO_LOAD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_big);
    {
        hipMemset(ws, 0, (size_t)tiles * splits * 65536);
        k_big<2, 0><<<btiles * splits, 256, lds_big>>>(x, gy, ws, M, N, K, splits, mtps, inv_g, inv_x);
        std::vector<float> hw((size_t)btiles * splits * 32768);
        hipMemcpy(hw.data(), ws, hw.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0;
        srand(7);
        for (int it = 0; it < 256; it++) {
            const int n = rand() % N, k = rand() % K;
            double ref = 0, sc = 0;
            for (int m = 0; m < M; m++) { ref += (double)hg[(size_t)m * N + n] * hx[(size_t)m * K + k]; sc += fabs((double)hg[(size_t)m * N + n] * hx[(size_t)m * K + k]); }
            const size_t tile = (size_t)(k / 128) * (N / 256) + n / 256;
            worst = fmax(worst, fabs(big_elem(hw, tile, splits, n % 256, k % 128) * s_x * s_g - ref) / sc);
        }
        printf("  %-44s error vs float64: worst %.2f units of 2^-24 sum|g||x| (256 outputs)\n", "256 x 128 tile", worst * 16777216.0);
    }
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_lib<3, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_lib);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_lib<2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_lib);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_lib<3, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_lib);
    hipMemset(ws, 0, (size_t)tiles * splits * 65536);
    k_lib<2, 1><<<tiles * splits, 256, lds_lib>>>(x, gy, ws, M, N, K, splits, mtps, inv_g, inv_x);
    check("library loop, fetch two stages ahead", true);
    hipMemset(ws, 0, (size_t)tiles * splits * 65536);
    k_lib<3, 0><<<tiles * splits, 256, lds_lib>>>(x, gy, ws, M, N, K, splits, mtps, inv_g, inv_x);
    check("library loop", true);
    hipMemset(ws, 0, (size_t)tiles * splits * 65536);
    k_direct<2, 0><<<tiles * splits, 256, lds_dir>>>(x, gy, ws, M, N, K, splits, mtps, inv_g, inv_x);
    check("x direct, gy through LDS", false);
    const double gf = 2.0 * M * N * K * 1e-9;
    for (int round = 0; round < 2; round++) {
        float t;
        t = time_it([&] { k_lib<3, 0><<<tiles * splits, 256, lds_lib>>>(x, gy, ws, M, N, K, splits, mtps, inv_g, inv_x); });
        printf("  %-44s %8.4f ms  %6.1f TF-eq\n", "library loop (3 w / SIMD)", t, gf / t);
        t = time_it([&] { k_lib<2, 1><<<tiles * splits, 256, lds_lib>>>(x, gy, ws, M, N, K, splits, mtps, inv_g, inv_x); });
        printf("  %-44s %8.4f ms  %6.1f TF-eq\n", "library loop, fetch two stages ahead (2 w)", t, gf / t);
        t = time_it([&] { k_lib<3, 1><<<tiles * splits, 256, lds_lib>>>(x, gy, ws, M, N, K, splits, mtps, inv_g, inv_x); });
        printf("  %-44s %8.4f ms  %6.1f TF-eq\n", "library loop, fetch two stages ahead (3 w)", t, gf / t);
        t = time_it([&] { k_big<2, 0><<<btiles * splits, 256, lds_big>>>(x, gy, ws, M, N, K, splits, mtps, inv_g, inv_x); });
        printf("  %-44s %8.4f ms  %6.1f TF-eq\n", "256 x 128 tile (2 w / SIMD), same splits", t, gf / t);
        t = time_it([&] { k_big<2, 0><<<btiles * splits * 2, 256, lds_big>>>(x, gy, ws, M, N, K, splits * 2, (m_tiles + 2 * splits - 1) / (2 * splits), inv_g, inv_x); });
        printf("  %-44s %8.4f ms  %6.1f TF-eq\n", "256 x 128 tile, twice the splits", t, gf / t);
        t = time_it([&] { k_big<2, KO_LOAD><<<btiles * splits * 2, 256, lds_big>>>(x, gy, ws, M, N, K, splits * 2, (m_tiles + 2 * splits - 1) / (2 * splits), inv_g, inv_x); });
        printf("  %-44s %8.4f ms  %6.1f TF-eq\n", "256 x 128 tile - global loads", t, gf / t);
        t = time_it([&] { k_direct<2, 0><<<tiles * splits, 256, lds_dir>>>(x, gy, ws, M, N, K, splits, mtps, inv_g, inv_x); });
        printf("  %-44s %8.4f ms  %6.1f TF-eq\n", "x direct (2 w / SIMD)", t, gf / t);
        t = time_it([&] { k_direct<3, 0><<<tiles * splits, 256, lds_dir>>>(x, gy, ws, M, N, K, splits, mtps, inv_g, inv_x); });
        printf("  %-44s %8.4f ms  %6.1f TF-eq\n", "x direct (3 w / SIMD)", t, gf / t);
        t = time_it([&] { k_direct<2, KO_SPLIT><<<tiles * splits, 256, lds_dir>>>(x, gy, ws, M, N, K, splits, mtps, inv_g, inv_x); });
        printf("  %-44s %8.4f ms  %6.1f TF-eq\n", "x direct - split arithmetic", t, gf / t);
        t = time_it([&] { k_direct<2, KO_LDS><<<tiles * splits, 256, lds_dir>>>(x, gy, ws, M, N, K, splits, mtps, inv_g, inv_x); });
        printf("  %-44s %8.4f ms  %6.1f TF-eq\n", "x direct - LDS stores and reads", t, gf / t);
        t = time_it([&] { k_direct<2, KO_LOAD><<<tiles * splits, 256, lds_dir>>>(x, gy, ws, M, N, K, splits, mtps, inv_g, inv_x); });
        printf("  %-44s %8.4f ms  %6.1f TF-eq\n", "x direct - global loads", t, gf / t);
        t = time_it([&] { k_direct<2, KO_MFMA><<<tiles * splits, 256, lds_dir>>>(x, gy, ws, M, N, K, splits, mtps, inv_g, inv_x); });
        printf("  %-44s %8.4f ms  %6.1f TF-eq\n", "x direct - MFMAs", t, gf / t);
        t = time_it([&] { k_direct<2, KO_LOAD | KO_SPLIT | KO_LDS><<<tiles * splits, 256, lds_dir>>>(x, gy, ws, M, N, K, splits, mtps, inv_g, inv_x); });
        printf("  %-44s %8.4f ms  %6.1f TF-eq\n", "x direct: MFMAs only", t, gf / t);
    }
    hipFree(x); hipFree(gy); hipFree(ws);
}

int main(int argc, char** argv) {
    if (argc > 4) { run_shape(atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), atoi(argv[4])); return 0; }
    run_shape(4096, 256, 256, 2);        // correctness first (small)
    run_shape(36864, 2048, 512, 8);      // layer4 conv3 / conv1: 64 tiles x 8 = 512 workgroups of 144 stages
    run_shape(36864, 512, 2048, 8);
    run_shape(9600, 1024, 256, 16);      // layer3: 16 tiles x 16 = 256 workgroups of 19 stages
    run_shape(9600, 256, 1024, 16);
    return 0;
}
