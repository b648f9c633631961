// bf16x6 main loop, version 4: operands pre-split into bf16 planes in HBM and brought into LDS by LDS-DMA (global_load_lds_dwordx4:
// no VGPR staging, no ds_write).  The DMA writes lane-linear (base + 16 B x lane), so the tile is unpadded [row][2 chunks] and the
// bank swizzle is applied through WHICH global chunk each lane fetches (chunk ^ ((row >> 3) & 1)) and again on the fragment read.
// k-stage 16, NBUF-deep LDS ring, plain TN GEMM, random operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
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
constexpr int BM = 128, BN = 128, BK = 16;
constexpr int PLANE = BM * BK;            // bf16 elements of one plane tile (4 KB)
constexpr int STAGE = 6 * PLANE;          // A planes then B planes (24 KB)
template <int NBUF>
__global__ __launch_bounds__(256) void k4(const __bf16* __restrict__ Ap, const __bf16* __restrict__ Bp, float* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(1024))) __bf16 lds[];   // [NBUF][6][128 rows][2 chunks][8]
    const int tiles_n = N / BN;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    // DMA assignment: a stage is 24 wave-instructions of 1 KB (= 32 rows x 2 chunks of one plane); wave w issues pieces w, w+4, ...
    // piece p -> operand/plane pl = p / 4 (0..2 A, 3..5 B), row block rb = p % 4; lane -> row = rb*32 + lane/2, slot = lane & 1,
    // global chunk = slot ^ ((row >> 3) & 1)
    const __bf16* src[6];
    for (int j = 0; j < 6; j++) {
        const int p = wave + 4 * j, pl = p >> 2, rb = p & 3;
        const int row = rb * 32 + (lane >> 1), ch = (lane & 1) ^ ((row >> 3) & 1);
        src[j] = pl < 3 ? Ap + (size_t)pl * M * K + (size_t)(m0 + row) * K + ch * 8
                        : Bp + (size_t)(pl - 3) * N * K + (size_t)(n0 + row) * K + ch * 8;
    }
    auto dma_stage = [&](int kt, int buf) {
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const int p = wave + 4 * j;
            __bf16* dst = lds + buf * STAGE + (p >> 2) * PLANE + (p & 3) * 32 * BK;   // wave-uniform
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + kt * BK),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
    f32x16 acc[2][2];
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    const int l31 = lane & 31, lh = lane >> 5;
    // fragment: row = base + l31 (+32 i), k-chunk lh -> slot = lh ^ ((row >> 3) & 1); (row + 32 i) >> 3 & 1 == (row >> 3) & 1
    const int ra = wm * 64 + l31, rb_ = wn * 64 + l31;
    const int offa = ra * BK + ((lh ^ ((ra >> 3) & 1)) * 8), offb = 3 * PLANE + rb_ * BK + ((lh ^ ((rb_ >> 3) & 1)) * 8);
    auto compute_stage = [&](int buf) {
        const __bf16* s = lds + buf * STAGE;
        bf16x8 fa[2][3], fb[2][3];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int pl = 0; pl < 3; pl++) {
                fa[i][pl] = *reinterpret_cast<const bf16x8*>(s + offa + pl * PLANE + i * 32 * BK);
                fb[i][pl] = *reinterpret_cast<const bf16x8*>(s + offb + pl * PLANE + i * 32 * BK);
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
    };
    const int nk = K / BK;
    // prologue: NBUF-1 stages in flight
#pragma unroll
    for (int s = 0; s < NBUF - 1; s++) dma_stage(s, s);
    for (int kt = 0; kt < nk; kt++) {
        // stage kt must have landed: all but the (NBUF-2) most recent stages' DMAs complete (6 per stage per wave)
        if (NBUF == 2) __builtin_amdgcn_s_waitcnt(0x0070 | 0x3F00 | 0xC000 | 0);          // vmcnt(0) (lgkm/exp untouched)
        else if (NBUF == 3) __builtin_amdgcn_s_waitcnt(0x0070 | 0x3F00 | 0xC000 | 6);     // vmcnt(6)
        else __builtin_amdgcn_s_waitcnt(0x0070 | 0x3F00 | 0xC000 | 12);                   // vmcnt(12)
        __syncthreads();                                   // everyone's pieces of stage kt are in; everyone is done reading stage kt-1
        if (kt + NBUF - 1 < nk) dma_stage(kt + NBUF - 1, (kt + NBUF - 1) % NBUF);   // refill the buffer that stage kt-1 just vacated
        else {   // keep the vmcnt accounting uniform at the tail: issue dummy loads? no -- just wait for everything
            __builtin_amdgcn_s_waitcnt(0x0070 | 0x3F00 | 0xC000 | 0);
            __syncthreads();
        }
        compute_stage(kt % NBUF);
    }
    float s = 0.f;
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) s += acc[i][j][r];
    C[(size_t)blockIdx.x * 256 + tid] = s;
}
// reference check of ONE output element via the same planes (validates the swizzle): compare sum of acc with host? keep simple:
template <int NBUF> void run(const char* name, const __bf16* A, const __bf16* B, float* C, int M, int N, int K, std::vector<float>* keep) {
    const size_t lds = sizeof(__bf16) * NBUF * STAGE;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k4<NBUF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid((M / BM) * (N / BN));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) k4<NBUF><<<grid, 256, lds>>>(A, B, C, M, N, K);
    hipEventRecord(a);
    for (int i = 0; i < 10; i++) k4<NBUF><<<grid, 256, lds>>>(A, B, C, M, N, K);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
    std::vector<float> h(256); hipMemcpy(h.data(), C, 1024, hipMemcpyDeviceToHost);
    double chk = 0; for (float v : h) chk += v;
    printf("%-28s lds %3zu KB  %.3f ms  %.1f TF fp32-equivalent   checksum(block 0) %.6e\n", name, lds / 1024, ms, 2.0 * M * N * K / ms / 1e9, chk);
    if (keep) *keep = h;
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
    split3<<<4096, 256>>>((const float4*)A, (size_t)M * K / 4, (uint2*)Ap);
    split3<<<4096, 256>>>((const float4*)B, (size_t)N * K / 4, (uint2*)Bp);
    // host reference for block 0's checksum: sum over the 128x128 tile of A[0:128] . B[0:128]^T (double)
    double ref = 0;
    for (int i = 0; i < 128; i++) for (int j = 0; j < 128; j++) { double d = 0; for (int k = 0; k < K; k++) d += (double)h[(size_t)i * K + k] * (double)h[(size_t)j * K + k]; ref += d; }
    printf("host reference checksum(block 0) %.6e\n", ref);
    run<2>("LDS-DMA planes, 2 buffers", Ap, Bp, C, M, N, K, nullptr);
    run<3>("LDS-DMA planes, 3 buffers", Ap, Bp, C, M, N, K, nullptr);
    run<4>("LDS-DMA planes, 4 buffers", Ap, Bp, C, M, N, K, nullptr);
    return 0;
}
