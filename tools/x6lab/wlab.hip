// bf16x6 WEIGHT-GRADIENT main-loop lab: where does conv_wgrad_x6_kernel's time go?
//
//   dW[n, k] = sum_m gy[m, n] * x[m, k]        (plain operands, m-major; 128 x 128 output tile per workgroup, split-M)
//
// The kernel below is the library's loop (octet-major LDS image, 32-row stages, single-buffered, two barriers per stage) with KNOCK-OUT
// switches: each variant removes ONE ingredient (results are then wrong; only the time matters) so that the difference to the full kernel
// prices that ingredient in place -- fetch, split arithmetic, LDS stores, barriers, MFMAs, the parked partial tile.
//
//   hipcc --offload-arch=gfx950 -O3 -o wlab wlab.hip && ./wlab [M N K splits]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// x - y as ONE v_sub_f32 (the compiler would pack two of them into a v_pk_add_f32, which is slow beside MFMAs); -DPK_ADD restores the packed form
__device__ __forceinline__ float sub1(float x, float y) {
#ifdef PK_ADD
    return x - y;
#else
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
#endif
}
__device__ __forceinline__ void split_pair(const f32x2v f, unsigned& o0, unsigned& o1, unsigned& o2) {
    const bf16x2 h0 = __builtin_convertvector(f, bf16x2);
    const f32x2v h0f = __builtin_convertvector(h0, f32x2v);
    const f32x2v r1 = {sub1(f.x, h0f.x), sub1(f.y, h0f.y)};
    const bf16x2 h1 = __builtin_convertvector(r1, bf16x2);
    const f32x2v h1f = __builtin_convertvector(h1, f32x2v);
    const f32x2v r2 = {sub1(r1.x, h1f.x), sub1(r1.y, h1f.y)};
    const bf16x2 h2 = __builtin_convertvector(r2, bf16x2);
    o0 = *reinterpret_cast<const unsigned*>(&h0);
    o1 = *reinterpret_cast<const unsigned*>(&h1);
    o2 = *reinterpret_cast<const unsigned*>(&h2);
}

enum { KO_LOAD = 1, KO_SPLIT = 2, KO_STORE = 4, KO_BAR = 8, KO_MFMA = 16, KO_EPI = 32, KO_READ = 64, KO_EPIQ = 128, KO_EPINT = 256, OPT_PRIO = 512, OPT_PRIO_M = 1024 };
constexpr int MRX = 32;

__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned n) {
    const unsigned per = n / 8u, rem = n % 8u, x = bid % 8u, i = bid / 8u;
    return x < rem ? x * (per + 1) + i : rem * (per + 1) + (x - rem) * per + i;
}

template <int KO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void k_wgrad(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ ws,
                                                                                         int M, int N, int K, int splits, int mt_per_split) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int PL = (MRX / 8) * 128 * 4;
    unsigned* Gs = reinterpret_cast<unsigned*>(smem);
    unsigned* As = Gs + 3 * PL;
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
    u32x4 rr[8];
    auto load_tile = [&](int mt) {
        const unsigned base = voff + (unsigned)(mt * MRX * ld) * 4u;
#pragma unroll
        for (int i = 0; i < 8; i++) rr[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(base + (unsigned)(i * ld) * 4u), 0, 0);
    };
    unsigned* const st_base = (is_x ? As : Gs) + (oct * 4 * 32 + q) * 4;
    auto store_tile = [&]() {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            u32x4 o0, o1, o2;
#pragma unroll
            for (int pp = 0; pp < 4; pp++) {
                if constexpr (KO & KO_SPLIT) {
                    o0[pp] = rr[2 * pp][j]; o1[pp] = rr[2 * pp + 1][j]; o2[pp] = rr[2 * pp][j] ^ rr[2 * pp + 1][j];
                } else {
                    const f32x2v f = {__uint_as_float(rr[2 * pp][j]), __uint_as_float(rr[2 * pp + 1][j])};
                    unsigned a0, a1, a2;
                    split_pair(f, a0, a1, a2);
                    o0[pp] = a0; o1[pp] = a1; o2[pp] = a2;
                }
            }
            if constexpr (KO & KO_STORE) {
                asm volatile("" ::"v"(o0), "v"(o1), "v"(o2));
            } else {
                *reinterpret_cast<u32x4*>(st_base + j * 128) = o0;
                *reinterpret_cast<u32x4*>(st_base + j * 128 + PL) = o1;
                *reinterpret_cast<u32x4*>(st_base + j * 128 + 2 * PL) = o2;
            }
        }
    };
    f32x16 acc[2][2];
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++)
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    const int l31 = lane & 31, lh = lane >> 5;
    const unsigned* const g_rd = Gs + ((lh * 4 + 2 * wm) * 32 + l31) * 4;
    const unsigned* const a_rd = As + ((lh * 4 + 2 * wn) * 32 + l31) * 4;
    bf16x8 G[2][3], A[2][3];
    if constexpr (KO & KO_READ) {
        for (int pl = 0; pl < 3; pl++)
            for (int t = 0; t < 2; t++) { G[t][pl] = *reinterpret_cast<const bf16x8*>(g_rd + pl * PL + t * 128); A[t][pl] = *reinterpret_cast<const bf16x8*>(a_rd + pl * PL + t * 128); }
    }
    auto compute_tile = [&]() {
#pragma unroll
        for (int s = 0; s < MRX / 16; s++) {
            if constexpr (!(KO & KO_READ)) {
#pragma unroll
                for (int pl = 0; pl < 3; pl++)
#pragma unroll
                    for (int t = 0; t < 2; t++) {
                        G[t][pl] = *reinterpret_cast<const bf16x8*>(g_rd + pl * PL + (s * 8 + t) * 128);
                        A[t][pl] = *reinterpret_cast<const bf16x8*>(a_rd + pl * PL + (s * 8 + t) * 128);
                    }
            }
            if constexpr (KO & KO_MFMA) {
#pragma unroll
                for (int pl = 0; pl < 3; pl++)
#pragma unroll
                    for (int t = 0; t < 2; t++) asm volatile("" ::"v"(G[t][pl]), "v"(A[t][pl]));
            } else {
                constexpr int pg[6] = {2, 0, 1, 1, 0, 0}, pa[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int t = 0; t < 6; t++)
#pragma unroll
                    for (int i = 0; i < 2; i++)
#pragma unroll
                        for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(G[i][pg[t]], A[j][pa[t]], acc[i][j], 0, 0, 0);
            }
        }
    };
    auto bar = [&]() { if constexpr (!(KO & KO_BAR)) __syncthreads(); };
    if (mt0 < mt1) {
        load_tile(mt0);
        store_tile();
        __syncthreads();
        for (int mt = mt0; mt + 1 < mt1; mt++) {
            if constexpr (!(KO & KO_LOAD)) load_tile(mt + 1);
            __builtin_amdgcn_sched_barrier(0);
            compute_tile();
            bar();
            store_tile();
            bar();
        }
        compute_tile();
    }
    if constexpr (KO & KO_EPI) {
        float s = 0.f;
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 2; j++)
                for (int r = 0; r < 16; r++) s += acc[i][j][r];
        if (s == 123.456f) ws[tid] = s;
    } else {
        float4* dst = reinterpret_cast<float4*>(reinterpret_cast<char*>(ws) + ((size_t)tile * splits + split) * 65536) + tid;
        if constexpr (KO & KO_EPIQ) {   // a quarter of the tile, the rest folded in
            const f32x16 a = acc[0][0] + acc[0][1] + acc[1][0] + acc[1][1];
#pragma unroll
            for (int c = 0; c < 4; c++) dst[c * 256] = make_float4(a[4 * c], a[4 * c + 1], a[4 * c + 2], a[4 * c + 3]);
        } else {
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const f32x16& a = acc[t >> 1][t & 1];
                    typedef float f4 __attribute__((ext_vector_type(4)));
                    const f4 v = {a[4 * c], a[4 * c + 1], a[4 * c + 2], a[4 * c + 3]};
                    if constexpr (KO & KO_EPINT) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(dst + (t * 4 + c) * 256));
                    else *reinterpret_cast<f4*>(dst + (t * 4 + c) * 256) = v;
                }
        }
    }
}

// ---- producer / consumer waves: 512 threads; waves 0-3 only read fragments and issue MFMAs, waves 4-7 fetch (DEPTH stages ahead), split and
// store the planes; 16-row stages, three operand buffers (3 x 24 KB), one barrier per stage.  Chunks [octet][e][cp], column = 2 cp + e.
constexpr int SPL = 2 * 2 * 64 * 4;   // dwords per plane (4 KB)
constexpr int SBUF = 6 * SPL;         // one buffer: gy planes 0-2, x planes 0-2
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int DEPTH, int KO = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 3))) void k_spec(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ ws,
                                                                                           int M, int N, int K, int splits, int mt_per_split) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned* const lds = reinterpret_cast<unsigned*>(smem);
    const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tiles_n = N / 128, total_tiles = tiles_n * (K / 128);
    const int split = bid / total_tiles, tile = bid % total_tiles;
    const int n0 = (tile % tiles_n) * 128, k0 = (tile / tiles_n) * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt0 = split * mt_per_split * 2, mt1 = min(mt0 + mt_per_split * 2, (M + 15) / 16);
    const int T = mt1 - mt0;
    if (T <= 0) return;
    if (wave >= 4) {
        // ---------------- loader waves
        if constexpr (KO & OPT_PRIO) __builtin_amdgcn_s_setprio(3);
        const bool is_x = wave >= 6;
        const int oct = wave & 1;
        const __amdgpu_buffer_rsrc_t rsrc = is_x ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (unsigned)((size_t)M * K * 4), 0x00020000)
                                                 : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gy), 0, (unsigned)((size_t)M * N * 4), 0x00020000);
        const int col = (is_x ? k0 : n0) + lane * 2, ld = is_x ? K : N;
        const unsigned voff = (unsigned)col * 4u;
        u32x2 rr[DEPTH][8];
        auto fetch = [&](u32x2 (&r)[8], int t) {   // stage t of this workgroup (zeros past its slice)
            const unsigned base = t < T ? voff + (unsigned)(((mt0 + t) * 16 + oct * 8) * ld) * 4u : 0x80000000u;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if constexpr (KO & KO_LOAD) { r[i].x = base + i; r[i].y = base ^ i; }
                else r[i] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(base + (unsigned)(i * ld) * 4u), 0, 0);
            }
        };
        unsigned* const st_base = lds + (is_x ? 3 * SPL : 0) + (oct * 2 * 64 + lane) * 4;
        auto split_store = [&](const u32x2 (&r)[8], int boff) {
#pragma unroll
            for (int e = 0; e < 2; e++) {
                u32x4 o0, o1, o2;
#pragma unroll
                for (int pp = 0; pp < 4; pp++) {
                    if constexpr (KO & KO_SPLIT) {
                        o0[pp] = r[2 * pp][e]; o1[pp] = r[2 * pp + 1][e]; o2[pp] = r[2 * pp][e] ^ r[2 * pp + 1][e];
                    } else {
                    const f32x2v f = {__uint_as_float(r[2 * pp][e]), __uint_as_float(r[2 * pp + 1][e])};
                    unsigned a0, a1, a2;
                    split_pair(f, a0, a1, a2);
                    o0[pp] = a0; o1[pp] = a1; o2[pp] = a2;
                    }
                }
                unsigned* d = st_base + boff + e * 256;
                if constexpr (KO & KO_STORE) {
                    asm volatile("" ::"v"(o0), "v"(o1), "v"(o2));
                } else {
                *reinterpret_cast<u32x4*>(d) = o0;
                *reinterpret_cast<u32x4*>(d + SPL) = o1;
                *reinterpret_cast<u32x4*>(d + 2 * SPL) = o2;
                }
            }
        };
#pragma unroll
        for (int d = 0; d < DEPTH; d++) fetch(rr[d], d);
        // stages 0 and 1 -> buffers 0 and 1, then one stage per barrier interval: stage t + 2 is written while the MFMA waves work on stage t
        int boff = 0;
        auto advance = [&]() { boff = boff + SBUF == 3 * SBUF ? 0 : boff + SBUF; };
        int s = 0;   // next stage to split
        // (register sets are indexed statically: the loop below is unrolled by DEPTH)
        auto step = [&](u32x2 (&r)[8]) {
            split_store(r, boff);
            advance();
            fetch(r, s + DEPTH);
            s++;
        };
        // prologue: two stages
        static_assert(DEPTH >= 2, "");
        step(rr[0]);
        step(rr[1 % DEPTH]);
        __syncthreads();
        // main: at barrier interval t write stage t + 2
        int t = 0;
        while (true) {
#pragma unroll
            for (int d = 0; d < DEPTH; d++) {
                if (t >= T) break;
                step(rr[(d + 2) % DEPTH]);   // unconditional (zeros past the slice, into a buffer nobody reads): the number of loads in flight must not
                                             // depend on the path, or the compiler's wait counts fall back to "everything"
                __syncthreads();
                t++;
            }
            if (t >= T) break;
        }
        return;
    }
    // ---------------- MFMA waves
    if constexpr (KO & OPT_PRIO_M) __builtin_amdgcn_s_setprio(3);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const unsigned* const g_rd = lds + (lh * 2 * 64 + 32 * wm + l31) * 4;
    const unsigned* const a_rd = lds + 3 * SPL + (lh * 2 * 64 + 32 * wn + l31) * 4;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++)
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    __syncthreads();
    int boff = 0;
    for (int t = 0; t < T; t++) {
        bf16x8 G[2][3], A[2][3];
#pragma unroll
        for (int pl = 0; pl < 3; pl++)
#pragma unroll
            for (int u = 0; u < 2; u++) {
                G[u][pl] = *reinterpret_cast<const bf16x8*>(g_rd + boff + pl * SPL + u * 256);
                A[u][pl] = *reinterpret_cast<const bf16x8*>(a_rd + boff + pl * SPL + u * 256);
            }
        constexpr int pg[6] = {2, 0, 1, 1, 0, 0}, pa[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int u = 0; u < 6; u++)
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(G[i][pg[u]], A[j][pa[u]], acc[i][j], 0, 0, 0);
        boff = boff + SBUF == 3 * SBUF ? 0 : boff + SBUF;
        __syncthreads();
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

template <int DEPTH, int KO = 0>
static float run_spec(const char* name, const float* x, const float* gy, float* ws, int M, int N, int K, int splits, float base_ms) {
    const int tiles = (N / 128) * (K / 128), m_tiles = (M + MRX - 1) / MRX, mtps = (m_tiles + splits - 1) / splits;
    const size_t lds = sizeof(unsigned) * 3 * SBUF;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_spec<DEPTH, KO>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 10; i++) k_spec<DEPTH, KO><<<tiles * splits, 512, lds>>>(x, gy, ws, M, N, K, splits, mtps);
    hipDeviceSynchronize();
    const int reps = 20;
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) k_spec<DEPTH, KO><<<tiles * splits, 512, lds>>>(x, gy, ws, M, N, K, splits, mtps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("  %-34s %8.4f ms  %6.1f TF-eq   %+7.4f ms vs full\n", name, ms, 2.0 * M * N * K / ms * 1e-9, base_ms > 0 ? ms - base_ms : 0.f);
    return ms;
}

template <int KO>
static float run(const char* name, const float* x, const float* gy, float* ws, int M, int N, int K, int splits, float base_ms) {
    const int tiles = (N / 128) * (K / 128), m_tiles = (M + MRX - 1) / MRX, mtps = (m_tiles + splits - 1) / splits;
    const size_t lds = sizeof(unsigned) * 6 * (MRX / 8) * 128 * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad<KO>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 10; i++) k_wgrad<KO><<<tiles * splits, 256, lds>>>(x, gy, ws, M, N, K, splits, mtps);
    hipDeviceSynchronize();
    const int reps = 20;
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) k_wgrad<KO><<<tiles * splits, 256, lds>>>(x, gy, ws, M, N, K, splits, mtps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("  %-34s %8.4f ms  %6.1f TF-eq   %+7.4f ms vs full\n", name, ms, 2.0 * M * N * K / ms * 1e-9, base_ms > 0 ? ms - base_ms : 0.f);
    return ms;
}

static void run_shape(int M, int N, int K, int splits) {
    printf("dW[%d x %d] over M = %d, %d splits = %d workgroups of %d stages\n", N, K, M, splits, (N / 128) * (K / 128) * splits, (M + 31) / 32 / splits);
    float *x, *gy, *ws;
    hipMalloc(&x, (size_t)M * K * 4); hipMalloc(&gy, (size_t)M * N * 4); hipMalloc(&ws, (size_t)(N / 128) * (K / 128) * splits * 65536);
    std::vector<float> h((size_t)M * std::max(N, K));
    srand(1);
    for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(x, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(gy, h.data(), (size_t)M * N * 4, hipMemcpyHostToDevice);
    run<0>("(warm-up)", x, gy, ws, M, N, K, splits, 0.f);
    run<0>("(warm-up)", x, gy, ws, M, N, K, splits, 0.f);
    const float b = run<0>("full", x, gy, ws, M, N, K, splits, 0.f);
    run_spec<3>("producer / consumer waves, depth 3", x, gy, ws, M, N, K, splits, b);
    run_spec<4>("producer / consumer waves, depth 4", x, gy, ws, M, N, K, splits, b);
    run_spec<6>("producer / consumer waves, depth 6", x, gy, ws, M, N, K, splits, b);
    run_spec<3, OPT_PRIO>("p/c d3, loaders at priority 3", x, gy, ws, M, N, K, splits, b);
    run_spec<3, OPT_PRIO_M>("p/c d3, MFMA waves at priority 3", x, gy, ws, M, N, K, splits, b);
    run_spec<3, KO_LOAD>("p/c d3 - fetch", x, gy, ws, M, N, K, splits, b);
    run_spec<3, KO_SPLIT>("p/c d3 - split", x, gy, ws, M, N, K, splits, b);
    run_spec<3, KO_STORE>("p/c d3 - LDS stores", x, gy, ws, M, N, K, splits, b);
    run_spec<3, KO_LOAD | KO_SPLIT | KO_STORE>("p/c d3 idle loaders", x, gy, ws, M, N, K, splits, b);
    run<KO_LOAD>("- global fetch", x, gy, ws, M, N, K, splits, b);
    run<KO_SPLIT>("- split arithmetic", x, gy, ws, M, N, K, splits, b);
    run<KO_STORE>("- LDS stores", x, gy, ws, M, N, K, splits, b);
    run<KO_SPLIT | KO_STORE>("- split - LDS stores", x, gy, ws, M, N, K, splits, b);
    run<KO_READ>("- fragment reads", x, gy, ws, M, N, K, splits, b);
    run<KO_BAR>("- barriers", x, gy, ws, M, N, K, splits, b);
    run<KO_MFMA>("- MFMAs", x, gy, ws, M, N, K, splits, b);
    run<KO_EPI>("- parked partial tile", x, gy, ws, M, N, K, splits, b);
    run<KO_EPIQ>("quarter of the partial tile", x, gy, ws, M, N, K, splits, b);
    run<KO_EPINT>("partial tile, non-temporal stores", x, gy, ws, M, N, K, splits, b);
    run<KO_LOAD | KO_SPLIT | KO_STORE | KO_BAR | KO_READ>("MFMAs only", x, gy, ws, M, N, K, splits, b);
    run<KO_LOAD | KO_SPLIT | KO_STORE | KO_BAR>("MFMAs + fragment reads", x, gy, ws, M, N, K, splits, b);
    run<0>("full (again)", x, gy, ws, M, N, K, splits, b);
    run<KO_EPI>("- parked partial tile (again)", x, gy, ws, M, N, K, splits, b);
    run<0>("full (again)", x, gy, ws, M, N, K, splits, b);
    hipFree(x); hipFree(gy); hipFree(ws);
}

int main(int argc, char** argv) {
    if (argc > 4) { run_shape(atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), atoi(argv[4])); return 0; }
    run_shape(32768, 2048, 512, 8);     // layer4 conv3: 64 tiles x 8 = 512 workgroups of 128 stages
    run_shape(32768, 512, 2048, 8);
    run_shape(9600, 1024, 256, 16);     // layer3 conv3: 16 tiles x 16 = 256 workgroups of 19 stages
    run_shape(9600, 256, 2304 - 2304 % 128, 6);   // 36 tiles
    return 0;
}
