// bf16x6 main loop, version 5 (= version 2 with a templated wave tile TM x TN of 32x32 blocks; 4x2 -> 256x128 workgroup tile, one wave per SIMD):
// version 2: k-stage 16, double-buffered operand LDS (73.7 KB -> still two workgroups per CU), four fp32
// stages in flight in a register ring, and the split + LDS store of stage kt+1 placed in the SAME basic block as the MFMAs of
// stage kt so that the VALU / DS work issues in the shadow of the matrix pipe.  Plain TN GEMM on fp32 sources, random operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int BKS = 16, LDS_ = 24;   // pitch 24 bf16 = 48 B
template <int TM, int TN, int INTERLEAVE>
__global__ __launch_bounds__(256) void k2(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
    constexpr int BM = 2 * TM * 32, BN = 2 * TN * 32, NA = BM / 64, NB = BN / 64, NR = NA + NB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16* As = reinterpret_cast<__bf16*>(smem);            // [2][3][BM][LDS_]
    __bf16* Bs = As + 2 * 3 * BM * LDS_;                     // [2][3][BN][LDS_]
    const int tiles_n = N / BN;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int kq = tid & 3, srow = tid >> 2;                 // rows srow, srow + 64
    const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, (unsigned)((size_t)M * K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B), 0, (unsigned)((size_t)N * K * 4), 0x00020000);
    unsigned ao[NA], bo[NB];
    for (int i = 0; i < NA; i++) ao[i] = ((m0 + srow + 64 * i) * K + kq * 4) * 4u;
    for (int i = 0; i < NB; i++) bo[i] = ((n0 + srow + 64 * i) * K + kq * 4) * 4u;
    u32x4 ring[4][NR];
    auto load_stage = [&](int kt, u32x4 (&r)[NR], bool valid) {
        const unsigned oob = valid ? 0u : 0x80000000u;
#pragma unroll
        for (int i = 0; i < NA; i++) r[i] = __builtin_amdgcn_raw_buffer_load_b128(ra_, (int)(ao[i] | oob), kt * BKS * 4, 0);
#pragma unroll
        for (int i = 0; i < NB; i++) r[NA + i] = __builtin_amdgcn_raw_buffer_load_b128(rb_, (int)(bo[i] | oob), kt * BKS * 4, 0);
    };
    auto split_store = [](const u32x4 v, __bf16* dst, int ps) {
        const f32x4v f = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
        const bf16x4 h0 = __builtin_convertvector(f, bf16x4);
        const f32x4v r1 = f - __builtin_convertvector(h0, f32x4v);
        const bf16x4 h1 = __builtin_convertvector(r1, bf16x4);
        const f32x4v r2 = r1 - __builtin_convertvector(h1, f32x4v);
        const bf16x4 h2 = __builtin_convertvector(r2, bf16x4);
        *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<const uint2*>(&h0);
        *reinterpret_cast<uint2*>(dst + ps) = *reinterpret_cast<const uint2*>(&h1);
        *reinterpret_cast<uint2*>(dst + 2 * ps) = *reinterpret_cast<const uint2*>(&h2);
    };
    auto store_stage = [&](int buf, u32x4 (&r)[NR]) {
        __bf16* a = As + buf * 3 * BM * LDS_;
        __bf16* b = Bs + buf * 3 * BN * LDS_;
#pragma unroll
        for (int i = 0; i < NA; i++) split_store(r[i], a + (srow + 64 * i) * LDS_ + kq * 4, BM * LDS_);
#pragma unroll
        for (int i = 0; i < NB; i++) split_store(r[NA + i], b + (srow + 64 * i) * LDS_ + kq * 4, BN * LDS_);
    };
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; i++) for (int j = 0; j < TN; j++) for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    const int l31 = lane & 31, lh = lane >> 5;
    const __bf16* af = As + (wm * TM * 32 + l31) * LDS_ + lh * 8;
    const __bf16* bf = Bs + (wn * TN * 32 + l31) * LDS_ + lh * 8;
    auto compute_stage = [&](int buf) {
        bf16x8 fa[TM][3], fb[TN][3];
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int pl = 0; pl < 3; pl++) fa[i][pl] = *reinterpret_cast<const bf16x8*>(af + (buf * 3 + pl) * BM * LDS_ + i * 32 * LDS_);
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int pl = 0; pl < 3; pl++) fb[j][pl] = *reinterpret_cast<const bf16x8*>(bf + (buf * 3 + pl) * BN * LDS_ + j * 32 * LDS_);
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], acc[i][j], 0, 0, 0);
            }
    };
    const long long c0 = clock64(), w0 = wall_clock64();
    const int nk = K / BKS;   // multiple of 4 here
    load_stage(0, ring[0], true); load_stage(1, ring[1], true); load_stage(2, ring[2], true); load_stage(3, ring[3], true);
    store_stage(0, ring[0]);
    __syncthreads();
    // iteration kt: LDS[kt & 1] holds stage kt; split stage kt+1 (set (kt+1)&3) into the other buffer; refill set kt&3 with stage kt+4
#define ITER(KT, S_SPLIT, S_LOAD)                                                  \
    {                                                                              \
        compute_stage((KT) & 1);                                                   \
        if ((KT) + 1 < nk) store_stage(((KT) + 1) & 1, ring[S_SPLIT]);             \
        load_stage((KT) + 4, ring[S_LOAD], (KT) + 4 < nk);                         \
        if (INTERLEAVE == 1) {                                                     \
            _Pragma("unroll") for (int q = 0; q < TM * TN * 6; q++) {                       \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  /* 1 MFMA   */ \
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);  /* 4 VALU   */ \
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  /* 1 DS wr  */ \
            }                                                                      \
        } else if (INTERLEAVE == 2) {                                              \
            __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);      /* the 4 VMEM reads first */ \
            _Pragma("unroll") for (int q = 0; q < TM * TN * 6; q++) {                       \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 \
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);                 \
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                 \
            }                                                                      \
        } else if (INTERLEAVE == 3) {                                              \
            _Pragma("unroll") for (int q = 0; q < TM * TN * 6; q++) {                       \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 \
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                 \
            }                                                                      \
        } else if (INTERLEAVE == 4) {                                              \
            _Pragma("unroll") for (int q = 0; q < TM * TN * 3; q++) {                       \
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  /* DS read */  \
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                 \
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);                 \
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                 \
            }                                                                      \
        }                                                                          \
        __syncthreads();                                                           \
    }
    for (int kt = 0; kt < nk; kt += 4) {
        ITER(kt, 1, 0)
        ITER(kt + 1, 2, 1)
        ITER(kt + 2, 3, 2)
        ITER(kt + 3, 0, 3)
    }
    float s = 0.f;
    for (int i = 0; i < TM; i++) for (int j = 0; j < TN; j++) for (int r = 0; r < 16; r++) s += acc[i][j][r];
    C[(size_t)blockIdx.x * 256 + tid] = s;
    if (tid == 0 && blockIdx.x == gridDim.x / 2) {   // shader cycles per 100 MHz wall tick over this workgroup's life -> clock in MHz
        const long long c1 = clock64(), w1 = wall_clock64();
        C[(size_t)gridDim.x * 256] = (float)(c1 - c0) / (float)(w1 - w0) * 100.0f;
    }
}
template <int TM, int TN, int IL> void run(const char* name, const float* A, const float* B, float* C, int M, int N, int K) {
    constexpr int BM = 2 * TM * 32, BN = 2 * TN * 32;
    const size_t lds = sizeof(__bf16) * 2 * 3 * (BM + BN) * LDS_;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k2<TM, TN, IL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid((M / BM) * (N / BN));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) k2<TM, TN, IL><<<grid, 256, lds>>>(A, B, C, M, N, K);
    hipEventRecord(a);
    for (int i = 0; i < 10; i++) k2<TM, TN, IL><<<grid, 256, lds>>>(A, B, C, M, N, K);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
    float mhz = 0.f; hipMemcpy(&mhz, C + (size_t)grid.x * 256, 4, hipMemcpyDeviceToHost);
    printf("lds %3zu KB  %-40s %.3f ms  %.1f TF fp32-equivalent   shader clock %.0f MHz\n", lds / 1024, name, ms, 2.0 * M * N * K / ms / 1e9, mhz);
}
int main() {
    const int M = 32768, N = 2048, K = 1024;
    float *A, *B, *C;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&C, (size_t)(M / 128) * (N / 128) * 256 * 4 + 64);
    std::vector<float> h((size_t)M * K);
    unsigned st = 12345u;
    for (auto& v : h) { st = st * 1664525u + 1013904223u; v = ((int)(st >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }
    hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    run<2, 2, 1>("wave tile 64x64  (wg 128x128)", A, B, C, M, N, K);
    run<4, 2, 1>("wave tile 128x64 (wg 256x128)", A, B, C, M, N, K);
    run<4, 2, 0>("wave tile 128x64, compiler schedule", A, B, C, M, N, K);
    run<2, 4, 1>("wave tile 64x128 (wg 128x256)", A, B, C, M, N, K);
    run<2, 2, 1>("wave tile 64x64  (again)", A, B, C, M, N, K);
    return 0;
}
