// Sustained matrix-pipe rate on this chip, operands held in registers (no memory traffic): constant vs random operand bits.
// Data-dependent switching power moves the sustained clock, so "peak" for a real GEMM is below the datasheet number.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <bool RANDOM>
__global__ __launch_bounds__(256) void f32_loop(float* out, int iters) {
    const unsigned t = blockIdx.x * 256 + threadIdx.x;
    float a[4], b[4];
    for (int i = 0; i < 4; i++) {
        a[i] = RANDOM ? (float)(int)(hash(t * 8 + i) >> 8) * (1.0f / (1 << 23)) - 1.0f : 1.0f;
        b[i] = RANDOM ? (float)(int)(hash(t * 8 + 4 + i) >> 8) * (1.0f / (1 << 23)) - 1.0f : 0.5f;
    }
    f32x16 acc[4];
    for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(i + u) & 3], b[u], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
    out[t] = s;
}
template <bool RANDOM>
__global__ __launch_bounds__(256) void bf16_loop(float* out, int iters) {
    const unsigned t = blockIdx.x * 256 + threadIdx.x;
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; i++)
        for (int e = 0; e < 8; e++) {
            a[i][e] = (__bf16)(RANDOM ? (float)(int)(hash(t * 64 + i * 8 + e) >> 8) * (1.0f / (1 << 23)) - 1.0f : 1.0f);
            b[i][e] = (__bf16)(RANDOM ? (float)(int)(hash(t * 64 + 32 + i * 8 + e) >> 8) * (1.0f / (1 << 23)) - 1.0f : 0.5f);
        }
    f32x16 acc[4];
    for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + u) & 3], b[u], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
    out[t] = s;
}
template <typename K> void run(const char* name, K kern, float* out, int blocks, int iters, double flops_per_mfma) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    kern<<<blocks, 256>>>(out, iters / 8);
    hipDeviceSynchronize();
    hipEventRecord(a);
    kern<<<blocks, 256>>>(out, iters);       // ~ tens of milliseconds: long enough for the clock to settle
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double fl = (double)blocks * 4 /*waves*/ * iters * 16 * flops_per_mfma;
    printf("%-44s %8.2f ms  %8.1f TFLOP/s\n", name, ms, fl / ms / 1e9);
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * 2;   // 8 waves per CU
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    printf("%s, %d CUs, clock %d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
    run("v_mfma_f32_32x32x2_f32   constant operands", f32_loop<false>, out, blocks, 40000, 2.0 * 32 * 32 * 2);
    run("v_mfma_f32_32x32x2_f32   random operands", f32_loop<true>, out, blocks, 40000, 2.0 * 32 * 32 * 2);
    run("v_mfma_f32_32x32x16_bf16 constant operands", bf16_loop<false>, out, blocks, 80000, 2.0 * 32 * 32 * 16);
    run("v_mfma_f32_32x32x16_bf16 random operands", bf16_loop<true>, out, blocks, 80000, 2.0 * 32 * 32 * 16);
    return 0;
}
