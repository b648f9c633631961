// bf16x6 main-loop experiment 6: the WEIGHT operand never touches LDS.
//
// The weight planes are split once (per optimiser step in the library) and stored in MFMA-FRAGMENT order:
//   chunk(nb, ks, p) = 64 lanes x 16 B, lane l = row nb*32 + (l & 31), k = ks*16 + (l >> 5)*8 .. +8 of plane p
//   address = (((nb * KS + ks) * 3 + p) * 64 + lane) * 16 B          (KS = K / 16)
// so a wave's B fragment is ONE fully coalesced 1 KB load straight into the registers the MFMA reads: no ds_write, no ds_read, no
// split arithmetic for B.  LDS carries only the A planes (30.7 KB per 128-row tile), which leaves room to double-buffer them (one
// barrier per k-tile instead of two).  Per 16-k step a wave issues 6 ds_read_b128 + 6 global 16 B loads for 24 MFMAs (today: 12 + 0).
//
//   hipcc --offload-arch=gfx950 -O3 -o lab6 lab6.hip && ./lab6
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int BM = 128, BN = 128, BKX = 32, LDX = 40;

__device__ __forceinline__ void split_store(const u32x4 v, __bf16* dst, int ps) {
    const f32x4v f = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
    const bf16x4 h0 = __builtin_convertvector(f, bf16x4);
    const f32x4v r1 = f - __builtin_convertvector(h0, f32x4v);
    const bf16x4 h1 = __builtin_convertvector(r1, bf16x4);
    const f32x4v r2 = r1 - __builtin_convertvector(h1, f32x4v);
    const bf16x4 h2 = __builtin_convertvector(r2, bf16x4);
    *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<const uint2*>(&h0);
    *reinterpret_cast<uint2*>(dst + ps) = *reinterpret_cast<const uint2*>(&h1);
    *reinterpret_cast<uint2*>(dst + 2 * ps) = *reinterpret_cast<const uint2*>(&h2);
}

__device__ __forceinline__ void mfma6(f32x16& acc, const bf16x8 (&a)[3], const bf16x8 (&b)[3]) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
}

__device__ __forceinline__ void store_c(float* C, int N, int m_base, int n_base, int lane, const f32x16 (&acc)[2][2]) {
    const int l31 = lane & 31, lh = lane >> 5;
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++)
            for (int r = 0; r < 16; r++)
                C[(size_t)(m_base + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * N + n_base + j * 32 + l31] = acc[i][j][r];
}

// ---- baseline: today's library loop (both operands fp32 -> split -> LDS planes, single-buffered, two tiles in flight)
__global__ __launch_bounds__(256) void k_base(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16* As = reinterpret_cast<__bf16*>(smem);
    __bf16* Bs = As + 3 * BM * LDX;
    const int tiles_n = N / BN;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int kq = tid & 7, srow = tid >> 3;
    const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, (unsigned)((size_t)M * K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B), 0, (unsigned)((size_t)N * K * 4), 0x00020000);
    unsigned ao[4], bo[4];
    for (int i = 0; i < 4; i++) { ao[i] = ((m0 + srow + 32 * i) * K + kq * 4) * 4u; bo[i] = ((n0 + srow + 32 * i) * K + kq * 4) * 4u; }
    u32x4 ra0[4], rb0[4], ra1[4], rb1[4];
    auto load_tile = [&](int kt, u32x4 (&ra)[4], u32x4 (&rb)[4]) {
#pragma unroll
        for (int i = 0; i < 4; i++) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(ra_, (int)ao[i], kt * BKX * 4, 0);
#pragma unroll
        for (int i = 0; i < 4; i++) rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rb_, (int)bo[i], kt * BKX * 4, 0);
    };
    auto store_tile = [&](u32x4 (&ra)[4], u32x4 (&rb)[4]) {
#pragma unroll
        for (int i = 0; i < 4; i++) split_store(ra[i], As + (srow + 32 * i) * LDX + kq * 4, BM * LDX);
#pragma unroll
        for (int i = 0; i < 4; i++) split_store(rb[i], Bs + (srow + 32 * i) * LDX + kq * 4, BN * LDX);
    };
    f32x16 acc[2][2];
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    const int l31 = lane & 31, lh = lane >> 5;
    const __bf16* af = As + (wm * 64 + l31) * LDX + lh * 8;
    const __bf16* bf = Bs + (wn * 64 + l31) * LDX + lh * 8;
    auto compute_tile = [&]() {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            bf16x8 fa[2][3], fb[2][3];
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int pl = 0; pl < 3; pl++) {
                    fa[i][pl] = *reinterpret_cast<const bf16x8*>(af + pl * BM * LDX + i * 32 * LDX + u * 16);
                    fb[i][pl] = *reinterpret_cast<const bf16x8*>(bf + pl * BN * LDX + i * 32 * LDX + u * 16);
                }
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) mfma6(acc[i][j], fa[i], fb[j]);
        }
    };
    const int nk = K / BKX;
    load_tile(0, ra0, rb0);
    load_tile(1, ra1, rb1);
    store_tile(ra0, rb0);
    __syncthreads();
    int kt = 0;
    for (; kt + 2 < nk; kt += 2) {
        load_tile(kt + 2, ra0, rb0);
        __builtin_amdgcn_sched_barrier(0);
        compute_tile();
        __syncthreads();
        store_tile(ra1, rb1);
        __syncthreads();
        if (kt + 3 < nk) load_tile(kt + 3, ra1, rb1);
        __builtin_amdgcn_sched_barrier(0);
        compute_tile();
        __syncthreads();
        store_tile(ra0, rb0);
        __syncthreads();
    }
    if (kt + 1 < nk) {
        compute_tile();
        __syncthreads();
        store_tile(ra1, rb1);
        __syncthreads();
    }
    compute_tile();
    store_c(C, N, m0 + wm * 64, n0 + wn * 64, lane, acc);
}

// ---- weights -> fragment-packed bf16x3 planes
__global__ void pack_b(const float* __restrict__ B, u32x4* __restrict__ Bp, int N, int K) {
    const int KS = K / 16;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // one (nb, ks, lane)
    const size_t total = (size_t)(N / 32) * KS * 64;
    if (idx >= total) return;
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
        const __bf16 h2 = (__bf16)(r1 - (float)h1);
        h[0][e] = h0; h[1][e] = h1; h[2][e] = h2;
    }
    for (int p = 0; p < 3; p++) Bp[(c * 3 + p) * 64 + lane] = *reinterpret_cast<const u32x4*>(h[p]);
}

// ---- B direct from global; A through LDS planes (single-buffered, one tile in flight); wave tile (TM*32) x 64, workgroup (2*TM*32) x 128
template <int TM, int SCHED, bool IL>
__device__ __forceinline__ void bd_body(const float* __restrict__ A, const u32x4* __restrict__ Bp, float* __restrict__ C, int M, int N, int K) {
    constexpr int BMv = 2 * TM * 32, NA = BMv / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16* As = reinterpret_cast<__bf16*>(smem);   // [3][BMv][LDX]
    const int tiles_n = N / BN;
    const unsigned nblk = gridDim.x, q_ = nblk / 8, r_ = nblk % 8, xcd = blockIdx.x % 8, pos = blockIdx.x / 8;
    const int tile = (int)((xcd < r_ ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_) + pos);
    const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
    const int m0 = tile_m * BMv, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int kq = tid & 7, srow = tid >> 3;
    const int KS = K / 16;
    const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, (unsigned)((size_t)M * K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(Bp), 0, (unsigned)((size_t)N * K * 6), 0x00020000);
    unsigned ao[NA];
    for (int i = 0; i < NA; i++) ao[i] = ((m0 + srow + 32 * i) * K + kq * 4) * 4u;
    unsigned bo[2];
    for (int j = 0; j < 2; j++) bo[j] = (unsigned)((((size_t)((n0 + wn * 64) / 32 + j) * KS) * 3 * 64 + lane) * 16);
    u32x4 ra[NA];
    u32x4 fbr[2][2][3];
    auto load_a = [&](int kt) {
#pragma unroll
        for (int i = 0; i < NA; i++) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(ra_, (int)ao[i], kt * BKX * 4, 0);
    };
    auto store_a = [&]() {
#pragma unroll
        for (int i = 0; i < NA; i++) split_store(ra[i], As + (srow + 32 * i) * LDX + kq * 4, BMv * LDX);
    };
    auto load_b = [&](int kt, int u) {
        const int ks = kt * 2 + u;
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int p = 0; p < 3; p++) fbr[u][j][p] = __builtin_amdgcn_raw_buffer_load_b128(rb_, (int)bo[j], (ks * 3 + p) * 1024, 0);
    };
    f32x16 acc[TM][2];
    for (int i = 0; i < TM; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    const int l31 = lane & 31, lh = lane >> 5;
    const __bf16* af = As + (wm * (TM * 32) + l31) * LDX + lh * 8;
    const int nk = K / BKX;
    auto compute_tile = [&](int kt_next) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            bf16x8 fa[TM][3], fb[2][3];
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int pl = 0; pl < 3; pl++) fa[i][pl] = *reinterpret_cast<const bf16x8*>(af + pl * BMv * LDX + i * 32 * LDX + u * 16);
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int pl = 0; pl < 3; pl++) fb[j][pl] = *reinterpret_cast<const bf16x8*>(&fbr[u][j][pl]);
            if (IL) {
                constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int t = 0; t < 6; t++)
#pragma unroll
                    for (int i = 0; i < TM; i++)
#pragma unroll
                        for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][pa[t]], fb[j][pb[t]], acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++) mfma6(acc[i][j], fa[i], fb[j]);
            }
            if (kt_next < nk) load_b(kt_next, u);
            if (SCHED == 2 && u == 0) __builtin_amdgcn_sched_barrier(0);
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
    const int m_base = m0 + wm * (TM * 32), n_base = n0 + wn * 64;
    for (int i = 0; i < TM; i++)
        for (int j = 0; j < 2; j++)
            for (int r = 0; r < 16; r++)
                C[(size_t)(m_base + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * N + n_base + j * 32 + l31] = acc[i][j][r];
}

#define BD_KERNEL(name, TM, SCHED, OCC, IL)                                                                                     \
    __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void name(                                  \
        const float* __restrict__ A, const u32x4* __restrict__ Bp, float* __restrict__ C, int M, int N, int K) {                 \
        bd_body<TM, SCHED, IL>(A, Bp, C, M, N, K);                                                                               \
    }
BD_KERNEL(k_t2_o3, 2, 2, 3, true)
BD_KERNEL(k_t2_o2, 2, 2, 2, true)
BD_KERNEL(k_t4_o2, 4, 2, 2, true)
BD_KERNEL(k_t4_o2_s0, 4, 0, 2, true)
BD_KERNEL(k_t4_o2_nil, 4, 2, 2, false)
BD_KERNEL(k_t3_o2, 3, 2, 2, true)

static double check(const std::vector<float>& hA, const std::vector<float>& hB, const float* dC, int M, int N, int K) {
    std::vector<float> hC((size_t)M * N);
    hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0.0;
    unsigned st = 777u;
    for (int t = 0; t < 512; t++) {
        st = st * 1664525u + 1013904223u; const int m = (st >> 8) % M;
        st = st * 1664525u + 1013904223u; const int n = (st >> 8) % N;
        double s = 0.0, sa = 0.0;
        for (int k = 0; k < K; k++) { const double p = (double)hA[(size_t)m * K + k] * (double)hB[(size_t)n * K + k]; s += p; sa += fabs(p); }
        worst = fmax(worst, fabs((double)hC[(size_t)m * N + n] - s) / (sa * 5.96e-8));   // in units of 2^-24 * sum|a||b|
    }
    return worst;
}

template <typename F> float time_it(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) f();
    hipEventRecord(a);
    for (int i = 0; i < 10; i++) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 10;
}

static void run_shape(int M, int N, int K) {
    float *A, *B, *C; u32x4* Bp;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * N * 4); hipMalloc(&Bp, (size_t)N * K * 6);
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    unsigned st = 12345u;
    for (auto& v : hA) { st = st * 1664525u + 1013904223u; v = ((int)(st >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }
    for (auto& v : hB) { st = st * 1664525u + 1013904223u; v = ((int)(st >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }
    hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    const size_t chunks = (size_t)(N / 32) * (K / 16) * 64;
    pack_b<<<(unsigned)((chunks + 255) / 256), 256>>>(B, Bp, N, K);
    dim3 grid((M / BM) * (N / BN));
    const double gf = 2.0 * M * N * K / 1e9;
    printf("GEMM %d x %d x %d (random operands), %u workgroups; ms per round (3 interleaved rounds), best TF-eq, error in 2^-24 sum|a||b|\n", M, N, K, grid.x);
    struct V { const char* name; void (*fn)(const float*, const u32x4*, float*, int, int, int); int tm; };
    const V vs[] = {{"B direct, wave 64x64, 3 w/SIMD (library x6w)", k_t2_o3, 2}, {"B direct, wave 64x64, 2 w/SIMD", k_t2_o2, 2},
                    {"B direct, wave 128x64 (WG 256x128), 2 w/SIMD", k_t4_o2, 4}, {"B direct, wave 128x64, 2 w/SIMD, sched 0", k_t4_o2_s0, 4},
                    {"B direct, wave 128x64, 2 w/SIMD, 6-chains", k_t4_o2_nil, 4}, {"B direct, wave 96x64 (WG 192x128), 2 w/SIMD", k_t3_o2, 3}};
    constexpr int NV = sizeof(vs) / sizeof(vs[0]);
    float t[NV + 1][3];
    double err[NV + 1];
    const size_t lds0 = sizeof(__bf16) * 3 * (BM + BN) * LDX;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_base), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds0);
    for (int r = 0; r < 3; r++) {
        t[0][r] = time_it([&] { k_base<<<grid, 256, lds0>>>(A, B, C, M, N, K); });
        if (r == 0) err[0] = check(hA, hB, C, M, N, K);
        for (int v = 0; v < NV; v++) {
            const size_t lds = sizeof(__bf16) * 3 * (2 * vs[v].tm * 32) * LDX;
            dim3 grid((M / (2 * vs[v].tm * 32)) * (N / BN));
            if (M % (2 * vs[v].tm * 32)) { t[v + 1][r] = 0.f; err[v + 1] = -1; continue; }
            hipFuncSetAttribute(reinterpret_cast<const void*>(vs[v].fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (r == 0) hipMemset(C, 0, (size_t)M * N * 4);
            t[v + 1][r] = time_it([&] { vs[v].fn<<<grid, 256, lds>>>(A, Bp, C, M, N, K); });
            if (r == 0) err[v + 1] = check(hA, hB, C, M, N, K);
        }
    }
    for (int v = 0; v <= NV; v++) {
        const float best = fminf(t[v][0], fminf(t[v][1], t[v][2]));
        printf("  %-52s %.3f %.3f %.3f ms   %.1f TF-eq   err %.2f\n", v ? vs[v - 1].name : "baseline (A, B split -> LDS; two tiles in flight)", t[v][0], t[v][1], t[v][2],
               gf / best, err[v]);
    }
    hipFree(A); hipFree(B); hipFree(C); hipFree(Bp);
}

int main(int argc, char** argv) {
    if (argc > 3) { run_shape(atoi(argv[1]), atoi(argv[2]), atoi(argv[3])); return 0; }
    run_shape(32768, 2048, 1024);
    run_shape(32768, 2048, 512);
    run_shape(32768, 512, 2048);
    run_shape(36864, 1024, 1024);
    return 0;
}
