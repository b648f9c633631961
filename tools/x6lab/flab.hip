// bf16x6 FORWARD main-loop knock-out lab: where does conv_igemm_x6w_kernel<128,128>'s time go?
//
// The kernel is the library's loop on a plain GEMM (weights pre-split, fragment-packed, loaded global -> MFMA registers; A rows fetched fp32,
// split in registers, parked in LDS as three bf16 planes; single A buffer, two barriers per 32-k tile; interleaved accumulators; 3 waves / SIMD)
// with switches that each remove ONE ingredient (results are then wrong; only the time matters).
//
//   hipcc --offload-arch=gfx950 -O3 -o flab flab.hip && ./flab [M N K]
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


// the library's split (conv_igemm.hip::x6_split4): on pairs, residual subtractions as single v_sub_f32 (-DPK_ADD: left to the compiler = v_pk_add_f32)
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float sub1(float x, float y) {
#ifdef PK_ADD
    return x - y;
#else
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
#endif
}
__device__ __forceinline__ void split_pair(const float a, const float b, unsigned& o0, unsigned& o1, unsigned& o2) {
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

enum { KO_ALOAD = 1, KO_SPLIT = 2, KO_STORE = 4, KO_BLOAD = 8, KO_BAR = 16, KO_READ = 32, KO_MFMA = 64, KO_EPI = 128 };

template <int KO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_fwd(const float* __restrict__ A, const u32x4* __restrict__ Bp, float* __restrict__ C,
                                                                                         int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16* As = reinterpret_cast<__bf16*>(smem);   // [3][BM][LDX]
    const int tiles_n = N / BN;
    const unsigned nblk = gridDim.x, q_ = nblk / 8, r_ = nblk % 8, xcd = blockIdx.x % 8, pos = blockIdx.x / 8;
    const int tile = (int)((xcd < r_ ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_) + pos);
#ifndef GN
    const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
#else
    // n-group-major order: GN n-tile columns at a time, all m-tiles of them, n fastest inside the group: an XCD's contiguous tile range stays on
    // FEW weight columns (their planes stay in its L2) and streams the activation rows past them
    const int tiles_m_ = M / BM;
    const int ng = tile / (tiles_m_ * GN), rem = tile % (tiles_m_ * GN);
    const int tile_m = rem / GN, tile_n = ng * GN + rem % GN;
#endif
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int kq = tid & 7, srow = tid >> 3;
    const int KS = K / 16;
    const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, (unsigned)((size_t)M * K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(Bp), 0, (unsigned)((size_t)N * K * 6), 0x00020000);
    unsigned ao[4];
    for (int i = 0; i < 4; i++) ao[i] = ((m0 + srow + 32 * i) * K + kq * 4) * 4u;
    unsigned bo[2];
    for (int j = 0; j < 2; j++) bo[j] = (unsigned)((((size_t)((n0 + wn * 64) / 32 + j) * KS) * 3 * 64 + lane) * 16);
    u32x4 ra[4];
    u32x4 fbr[2][2][3];
    auto load_a = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; i++) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(ra_, (int)ao[i], kt * BKX * 4, 0);
    };
    auto store_a = [&]() {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            __bf16* dst = As + (srow + 32 * i) * LDX + kq * 4;
            uint2 o0, o1, o2;
            if constexpr (KO & KO_SPLIT) {
                o0 = make_uint2(ra[i].x, ra[i].y); o1 = make_uint2(ra[i].z, ra[i].w); o2 = make_uint2(ra[i].x ^ ra[i].z, ra[i].y ^ ra[i].w);
            } else {
                split_pair(__uint_as_float(ra[i].x), __uint_as_float(ra[i].y), o0.x, o1.x, o2.x);
                split_pair(__uint_as_float(ra[i].z), __uint_as_float(ra[i].w), o0.y, o1.y, o2.y);
            }
            if constexpr (KO & KO_STORE) {
                asm volatile("" ::"v"(o0), "v"(o1), "v"(o2));
            } else {
                *reinterpret_cast<uint2*>(dst) = o0;
                *reinterpret_cast<uint2*>(dst + BM * LDX) = o1;
                *reinterpret_cast<uint2*>(dst + 2 * BM * LDX) = o2;
            }
        }
    };
    auto load_b = [&](int kt, int u) {
        const int ks = kt * 2 + u;
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int p = 0; p < 3; p++) fbr[u][j][p] = __builtin_amdgcn_raw_buffer_load_b128(rb_, (int)bo[j], (ks * 3 + p) * 1024, 0);
    };
    f32x16 acc[2][2];
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    const int l31 = lane & 31, lh = lane >> 5;
    const __bf16* af = As + (wm * 64 + l31) * LDX + lh * 8;
    const int nk = K / BKX;
    bf16x8 fa[2][3];
    auto read_a = [&](int u) {
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int pl = 0; pl < 3; pl++) fa[i][pl] = *reinterpret_cast<const bf16x8*>(af + pl * BM * LDX + i * 32 * LDX + u * 16);
    };
    auto compute_tile = [&](int kt_next) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            bf16x8 fb[2][3];
            if constexpr (!(KO & KO_READ)) read_a(u);
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int pl = 0; pl < 3; pl++) fb[j][pl] = *reinterpret_cast<const bf16x8*>(&fbr[u][j][pl]);
            if constexpr (KO & KO_MFMA) {
#pragma unroll
                for (int pl = 0; pl < 3; pl++)
#pragma unroll
                    for (int i = 0; i < 2; i++) asm volatile("" ::"v"(fa[i][pl]), "v"(fb[i][pl]));
            } else {
                constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int t = 0; t < 6; t++)
#pragma unroll
                    for (int i = 0; i < 2; i++)
#pragma unroll
                        for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][pa[t]], fb[j][pb[t]], acc[i][j], 0, 0, 0);
            }
            if constexpr (!(KO & KO_BLOAD)) { if (kt_next < nk) load_b(kt_next, u); }
            if (u == 0) __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto bar = [&]() { if constexpr (!(KO & KO_BAR)) __syncthreads(); };
    load_b(0, 0);
    load_b(0, 1);
    load_a(0);
    store_a();
    __syncthreads();
    if constexpr (KO & KO_READ) read_a(0);
    for (int kt = 0; kt + 1 < nk; kt++) {
        if constexpr (!(KO & KO_ALOAD)) load_a(kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        compute_tile(kt + 1);
        bar();
        store_a();
        bar();
    }
    compute_tile(nk);
    if constexpr (KO & KO_EPI) {
        float s = 0.f;
        for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) s += acc[i][j][r];
        if (s == 123.456f) C[tid] = s;
    } else {
        store_c(C, N, m0 + wm * 64, n0 + wn * 64, lane, acc);
    }
}

template <int KO>
static float run(const char* name, const float* A, const u32x4* Bp, float* C, int M, int N, int K, float base) {
    const size_t lds = sizeof(__bf16) * 3 * BM * LDX;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_fwd<KO>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid((M / BM) * (N / BN));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 10; i++) k_fwd<KO><<<grid, 256, lds>>>(A, Bp, C, M, N, K);
    hipEventRecord(a);
    for (int i = 0; i < 20; i++) k_fwd<KO><<<grid, 256, lds>>>(A, Bp, C, M, N, K);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 20;
    printf("  %-36s %8.4f ms  %6.1f TF-eq   %+7.4f ms vs full\n", name, ms, 2.0 * M * N * K / ms * 1e-9, base > 0 ? ms - base : 0.f);
    return ms;
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
    printf("GEMM %d x %d x %d (random operands), %d workgroups\n", M, N, K, (M / BM) * (N / BN));
    run<0>("(warm-up)", A, Bp, C, M, N, K, 0.f);
    run<0>("(warm-up)", A, Bp, C, M, N, K, 0.f);
    const float b = run<0>("full", A, Bp, C, M, N, K, 0.f);
    run<KO_ALOAD>("- A fetch", A, Bp, C, M, N, K, b);
    run<KO_SPLIT>("- split arithmetic", A, Bp, C, M, N, K, b);
    run<KO_STORE>("- LDS stores", A, Bp, C, M, N, K, b);
    run<KO_SPLIT | KO_STORE>("- split - LDS stores", A, Bp, C, M, N, K, b);
    run<KO_BLOAD>("- B fragment loads", A, Bp, C, M, N, K, b);
    run<KO_READ>("- A fragment reads", A, Bp, C, M, N, K, b);
    run<KO_BAR>("- barriers", A, Bp, C, M, N, K, b);
    run<KO_MFMA>("- MFMAs", A, Bp, C, M, N, K, b);
    run<KO_EPI>("- C stores", A, Bp, C, M, N, K, b);
    run<KO_ALOAD | KO_SPLIT | KO_STORE | KO_BAR>("MFMAs + A reads + B loads", A, Bp, C, M, N, K, b);
    run<KO_ALOAD | KO_SPLIT | KO_STORE | KO_BAR | KO_READ>("MFMAs + B loads", A, Bp, C, M, N, K, b);
    run<KO_ALOAD | KO_SPLIT | KO_STORE | KO_BAR | KO_BLOAD>("MFMAs + A reads", A, Bp, C, M, N, K, b);
    run<KO_ALOAD | KO_SPLIT | KO_STORE | KO_BAR | KO_READ | KO_BLOAD>("MFMAs only", A, Bp, C, M, N, K, b);
    run<0>("full (again)", A, Bp, C, M, N, K, b);
    hipFree(A); hipFree(B); hipFree(C); hipFree(Bp);
}

int main(int argc, char** argv) {
    if (argc > 3) { run_shape(atoi(argv[1]), atoi(argv[2]), atoi(argv[3])); return 0; }
    run_shape(32768, 2048, 1024);
    run_shape(32768, 2048, 512);
    run_shape(9600, 1024, 256);
    return 0;
}
