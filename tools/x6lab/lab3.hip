// bf16x6 main loop, version 3: BOTH operands arrive pre-split (three bf16 planes each, made once by a separate pass), so the loop has
// no split arithmetic at all: 16 B plane chunks go global -> registers -> LDS.  k-stage 32, double-buffered, ring of 3 stages.
// Upper bound for the "split once" design.  Plain TN GEMM, random operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void split3(const float4* x, size_t n4, uint2* planes) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = x[i];
        const f32x4v f = {v.x, v.y, v.z, v.w};
        const bf16x4 h0 = __builtin_convertvector(f, bf16x4);
        const f32x4v r1 = f - __builtin_convertvector(h0, f32x4v);
        const bf16x4 h1 = __builtin_convertvector(r1, bf16x4);
        const f32x4v r2 = r1 - __builtin_convertvector(h1, f32x4v);
        const bf16x4 h2 = __builtin_convertvector(r2, bf16x4);
        planes[i] = *reinterpret_cast<const uint2*>(&h0); planes[n4 + i] = *reinterpret_cast<const uint2*>(&h1); planes[2 * n4 + i] = *reinterpret_cast<const uint2*>(&h2);
    }
}
constexpr int BM = 128, BN = 128;
template <int BK, int NBUF>
__global__ __launch_bounds__(256) void k3(const __bf16* __restrict__ Ap, const __bf16* __restrict__ Bp, float* __restrict__ C, int M, int N, int K) {
    constexpr int LD = BK + 8;
    constexpr int CPR = BK / 8;                 // 16 B chunks per row per plane
    constexpr int NCH = 3 * BM * CPR / 256;     // chunks per thread per operand per stage
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16* As = reinterpret_cast<__bf16*>(smem);            // [NBUF][3][BM][LD]
    __bf16* Bs = As + NBUF * 3 * BM * LD;
    const int tiles_n = N / BN;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const unsigned abytes = (unsigned)((size_t)M * K * 2), bbytes = (unsigned)((size_t)N * K * 2);
    const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(Ap), 0, 3u * abytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(Bp), 0, 3u * bbytes, 0x00020000);
    unsigned ao[NCH], bo[NCH], lo[NCH];
    for (int i = 0; i < NCH; i++) {
        const int c = tid + 256 * i, pl = c / (BM * CPR), rem = c % (BM * CPR), row = rem / CPR, kc = rem % CPR;
        ao[i] = pl * abytes + ((m0 + row) * K + kc * 8) * 2u;
        bo[i] = pl * bbytes + ((n0 + row) * K + kc * 8) * 2u;
        lo[i] = (pl * BM + row) * LD + kc * 8;
    }
    u32x4 ring[3][2 * NCH];
    auto load_stage = [&](int kt, u32x4 (&r)[2 * NCH], bool valid) {
        const unsigned oob = valid ? 0u : 0x80000000u;
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            r[i] = __builtin_amdgcn_raw_buffer_load_b128(ra_, (int)(ao[i] | oob), kt * BK * 2, 0);
            r[NCH + i] = __builtin_amdgcn_raw_buffer_load_b128(rb_, (int)(bo[i] | oob), kt * BK * 2, 0);
        }
    };
    auto store_stage = [&](int buf, u32x4 (&r)[2 * NCH]) {
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            *reinterpret_cast<u32x4*>(As + buf * 3 * BM * LD + lo[i]) = r[i];
            *reinterpret_cast<u32x4*>(Bs + buf * 3 * BN * LD + lo[i]) = r[NCH + i];
        }
    };
    f32x16 acc[2][2];
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    const int l31 = lane & 31, lh = lane >> 5;
    const __bf16* af = As + (wm * 64 + l31) * LD + lh * 8;
    const __bf16* bf = Bs + (wn * 64 + l31) * LD + lh * 8;
    auto compute_stage = [&](int buf) {
#pragma unroll
        for (int u = 0; u < BK / 16; u++) {
            bf16x8 fa[2][3], fb[2][3];
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int pl = 0; pl < 3; pl++) {
                    fa[i][pl] = *reinterpret_cast<const bf16x8*>(af + (buf * 3 + pl) * BM * LD + i * 32 * LD + u * 16);
                    fb[i][pl] = *reinterpret_cast<const bf16x8*>(bf + (buf * 3 + pl) * BN * LD + i * 32 * LD + u * 16);
                }
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], acc[i][j], 0, 0, 0);
                }
        }
    };
    const int nk = K / BK;   // multiple of 3 not required: guards below
    load_stage(0, ring[0], true); load_stage(1, ring[1], 1 < nk); load_stage(2, ring[2], 2 < nk);
    store_stage(0, ring[0]);
    __syncthreads();
#define ITER(KT, S_NEXT, S_LOAD)                                                      \
    if ((KT) < nk) {                                                                  \
        if (NBUF == 2) {                                                              \
            compute_stage((KT) & 1);                                                  \
            if ((KT) + 1 < nk) store_stage(((KT) + 1) & 1, ring[S_NEXT]);             \
            load_stage((KT) + 3, ring[S_LOAD], (KT) + 3 < nk);                        \
            _Pragma("unroll") for (int q = 0; q < 24 * (BK / 16); q++) {              \
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                    \
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                    \
            }                                                                         \
            __syncthreads();                                                          \
        } else {                                                                      \
            compute_stage(0);                                                         \
            __syncthreads();                                                          \
            if ((KT) + 1 < nk) store_stage(0, ring[S_NEXT]);                          \
            load_stage((KT) + 3, ring[S_LOAD], (KT) + 3 < nk);                        \
            __syncthreads();                                                          \
        }                                                                             \
    }
    for (int kt = 0; kt < nk; kt += 3) {
        ITER(kt, 1, 0)
        ITER(kt + 1, 2, 1)
        ITER(kt + 2, 0, 2)
    }
    float s = 0.f;
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) s += acc[i][j][r];
    C[(size_t)blockIdx.x * 256 + tid] = s;
}
template <int BK, int NBUF> void run(const char* name, const __bf16* A, const __bf16* B, float* C, int M, int N, int K) {
    const size_t lds = sizeof(__bf16) * NBUF * 3 * (BM + BN) * (BK + 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k3<BK, NBUF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid((M / BM) * (N / BN));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) k3<BK, NBUF><<<grid, 256, lds>>>(A, B, C, M, N, K);
    hipEventRecord(a);
    for (int i = 0; i < 10; i++) k3<BK, NBUF><<<grid, 256, lds>>>(A, B, C, M, N, K);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
    printf("%-44s lds %3zu KB  %.3f ms  %.1f TF fp32-equivalent\n", name, lds / 1024, ms, 2.0 * M * N * K / ms / 1e9);
}
int main() {
    const int M = 32768, N = 2048, K = 1024;
    float *A, *B, *C; __bf16 *Ap, *Bp;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&C, (size_t)(M / BM) * (N / BN) * 256 * 4);
    hipMalloc(&Ap, (size_t)M * K * 6); hipMalloc(&Bp, (size_t)N * K * 6);
    std::vector<float> h((size_t)M * K);
    unsigned st = 12345u;
    for (auto& v : h) { st = st * 1664525u + 1013904223u; v = ((int)(st >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }
    hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    split3<<<4096, 256>>>((const float4*)A, (size_t)M * K / 4, (uint2*)Ap);
    hipEventRecord(a);
    split3<<<4096, 256>>>((const float4*)A, (size_t)M * K / 4, (uint2*)Ap);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("split pass over A (%d x %d): %.3f ms\n", M, K, ms);
    split3<<<4096, 256>>>((const float4*)B, (size_t)N * K / 4, (uint2*)Bp);
    run<32, 1>("planes, k32, single-buffered", Ap, Bp, C, M, N, K);
    run<16, 2>("planes, k16, double-buffered + interleave", Ap, Bp, C, M, N, K);
    run<32, 2>("planes, k32, double-buffered + interleave", Ap, Bp, C, M, N, K);
    return 0;
}
