// Winograd F(4x4, 3x3) transforms for the stride-1, pad-1 3x3 convolutions of the path (RPN head conv, layer2-4 conv2 and
// their dgrads): Y = A^T [ (G g G^T) (.) (B^T d B) ] A per 6x6 input tile / 4x4 output tile (Lavin & Gray, "Fast Algorithms for
// Convolutional Neural Networks", 2015: the minimal-filtering matrices below are theirs).  The 36 element-wise products over the
// channel axis are 36 independent GEMMs [tiles x Cin] x [Cin x Cout] -- 4x fewer multiply-adds than the direct implicit GEMM --
// and run on the fp32 MFMA kernel of conv_igemm.hip in batched mode; this file holds the three HBM-bound transform kernels.
//
//   V[p][t][c]  = (B^T d B)[p]      p = 6*i + j, t = tile (image, tile row, tile col), c = input channel     (input transform)
//   U[p][n][c]  = (G g G^T)[p]      n = output channel                                                      (weight transform)
//   M[p][t][n]  = sum_c V[p][t][c] * U[p][n][c]                                                             (batched GEMM)
//   out[b,y,x,n] = epilogue((A^T M A)[...])                                                                 (output transform)
//
// Arithmetic is fp32 throughout; the result differs from the direct convolution by re-association only (<= 2.4e-5 of the largest
// output at Cin = 1024, bounded by tests/test_gpu_ops.py against a float64 reference; ~2e-6 for the direct kernel), inside the 1e-4
// the path is held to.  The weight gradient takes the same route: dU[p] = sum_tiles (A dY A^T)[p]^T V[p], dW += scale * G^T dU G.  Reference semantics replaced: the same cuDNN conv + FrozenBN + ReLU
// (+ ReLU mask in backward) as conv_igemm.hip.
#include <map>

#include "common.h"

namespace {

// The transforms work on TWO adjacent channels per thread (8 B loads / stores, half the memory instructions of a scalar version;
// four channels would need > 200 VGPRs for the 6x6 tile).
struct f2 {
    float x, y;
};
__device__ __forceinline__ f2 operator+(f2 a, f2 b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ f2 operator-(f2 a, f2 b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ f2 operator-(f2 a) { return {-a.x, -a.y}; }
__device__ __forceinline__ f2 operator*(float s, f2 a) { return {s * a.x, s * a.y}; }
__device__ __forceinline__ f2 ld2(const float* p) { const float2 v = *reinterpret_cast<const float2*>(p); return {v.x, v.y}; }
__device__ __forceinline__ void st2(float* p, f2 v) { *reinterpret_cast<float2*>(p) = make_float2(v.x, v.y); }

// B^T (6x6), applied to columns then rows
template <typename T>
__device__ __forceinline__ void bt6(const T d0, const T d1, const T d2, const T d3, const T d4, const T d5, T& o0, T& o1, T& o2, T& o3, T& o4,
                                    T& o5) {
    o0 = 4.f * d0 - 5.f * d2 + d4;
    o1 = d3 + d4 - 4.f * (d1 + d2);
    o2 = 4.f * d1 - 4.f * d2 - d3 + d4;
    o3 = 2.f * (d3 - d1) - d2 + d4;
    o4 = 2.f * d1 - d2 - 2.f * d3 + d4;
    o5 = 4.f * d1 - 5.f * d3 + d5;
}

// grid-stride over (tile, channel pair); consecutive threads = consecutive channel pairs (coalesced 512 B per wave)
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, int B, int H, int W, int C, int th_n, int tw_n,
                                                         float* __restrict__ V) {
    const int64_t T = (int64_t)B * th_n * tw_n;
    const int C2 = C / 2;
    const int64_t total = T * C2;
    const int64_t ps = T * C;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = 2 * (int)(idx % C2);
        const int64_t t = idx / C2;
        const int tw = (int)(t % tw_n), th = (int)((t / tw_n) % th_n), b = (int)(t / ((int64_t)tw_n * th_n));
        const int y0 = 4 * th - 1, x0 = 4 * tw - 1;
        f2 tcol[6][6];  // B^T d, built column by column so that only one input column is live at a time
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const int xx = x0 + j;
            const bool xin = (unsigned)xx < (unsigned)W;
            f2 d[6];
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const int y = y0 + i;
                d[i] = (xin && (unsigned)y < (unsigned)H) ? ld2(x + (((int64_t)b * H + y) * W + xx) * C + c) : f2{0.f, 0.f};
            }
            bt6(d[0], d[1], d[2], d[3], d[4], d[5], tcol[0][j], tcol[1][j], tcol[2][j], tcol[3][j], tcol[4][j], tcol[5][j]);
        }
#pragma unroll
        for (int i = 0; i < 6; i++) {  // (B^T d) B : the same combination along the row
            f2 v0, v1, v2, v3, v4, v5;
            bt6(tcol[i][0], tcol[i][1], tcol[i][2], tcol[i][3], tcol[i][4], tcol[i][5], v0, v1, v2, v3, v4, v5);
            float* o = V + ((int64_t)(6 * i) * T + t) * C + c;
            st2(o, v0); st2(o + ps, v1); st2(o + 2 * ps, v2); st2(o + 3 * ps, v3); st2(o + 4 * ps, v4); st2(o + 5 * ps, v5);
        }
    }
}

// G (6x3) on a 3-vector
__device__ __forceinline__ void g6(const float g0, const float g1, const float g2, float& o0, float& o1, float& o2, float& o3, float& o4,
                                   float& o5) {
    o0 = 0.25f * g0;
    o1 = (-1.f / 6.f) * (g0 + g1 + g2);
    o2 = (-1.f / 6.f) * (g0 - g1 + g2);
    o3 = (1.f / 24.f) * g0 + (1.f / 12.f) * g1 + (1.f / 6.f) * g2;
    o4 = (1.f / 24.f) * g0 - (1.f / 12.f) * g1 + (1.f / 6.f) * g2;
    o5 = g2;
}

// w [N][3][3][C] (OHWI) -> U [36][N][C]
__global__ __launch_bounds__(256) void wino_weight_kernel(const float* __restrict__ w, int N, int C, float* __restrict__ U) {
    const int64_t total = (int64_t)N * C;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        const int64_t n = idx / C;
        float g[3][3];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int s = 0; s < 3; s++) g[r][s] = w[((n * 3 + r) * 3 + s) * C + c];
        float t[6][3];  // G g
#pragma unroll
        for (int s = 0; s < 3; s++) g6(g[0][s], g[1][s], g[2][s], t[0][s], t[1][s], t[2][s], t[3][s], t[4][s], t[5][s]);
#pragma unroll
        for (int i = 0; i < 6; i++) {
            float u0, u1, u2, u3, u4, u5;
            g6(t[i][0], t[i][1], t[i][2], u0, u1, u2, u3, u4, u5);
            float* o = U + ((int64_t)(6 * i) * N + n) * C + c;
            const int64_t ps = (int64_t)N * C;
            o[0] = u0; o[ps] = u1; o[2 * ps] = u2; o[3 * ps] = u3; o[4 * ps] = u4; o[5 * ps] = u5;
        }
    }
}

// A^T (4x6) on a 6-vector
template <typename T>
__device__ __forceinline__ void at4(const T m0, const T m1, const T m2, const T m3, const T m4, const T m5, T& o0, T& o1, T& o2, T& o3) {
    const T s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
    o0 = m0 + s12 + s34;
    o1 = d12 + 2.f * d34;
    o2 = s12 + 4.f * s34;
    o3 = d12 + 8.f * d34 + m5;
}

// M [36][T][N] -> out [B,H,W,N] with the conv epilogue (scale, bias, ReLU, ReLU mask of the producer); two channels per thread
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ Mm, int B, int H, int W, int N, int th_n, int tw_n,
                                                          const float* __restrict__ scale, const float* __restrict__ bias, int relu,
                                                          const float* __restrict__ mask, float* __restrict__ out) {
    const int64_t T = (int64_t)B * th_n * tw_n;
    const int N2 = N / 2;
    const int64_t total = T * N2;
    const int64_t ps = T * N;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int n = 2 * (int)(idx % N2);
        const int64_t t = idx / N2;
        const int tw = (int)(t % tw_n), th = (int)((t / tw_n) % th_n), b = (int)(t / ((int64_t)tw_n * th_n));
        const float* m = Mm + t * N + n;
        f2 tcol[4][6];  // A^T M
#pragma unroll
        for (int j = 0; j < 6; j++)
            at4(ld2(m + (0 + j) * ps), ld2(m + (6 + j) * ps), ld2(m + (12 + j) * ps), ld2(m + (18 + j) * ps), ld2(m + (24 + j) * ps),
                ld2(m + (30 + j) * ps), tcol[0][j], tcol[1][j], tcol[2][j], tcol[3][j]);
        const f2 sc = scale ? ld2(scale + n) : f2{1.f, 1.f}, bi = bias ? ld2(bias + n) : f2{0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            f2 y[4];
            at4(tcol[i][0], tcol[i][1], tcol[i][2], tcol[i][3], tcol[i][4], tcol[i][5], y[0], y[1], y[2], y[3]);
            const int oy = 4 * th + i;
            if (oy >= H) continue;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int ox = 4 * tw + j;
                if (ox >= W) continue;
                const int64_t o = (((int64_t)b * H + oy) * W + ox) * N + n;
                f2 v = {y[j].x * sc.x + bi.x, y[j].y * sc.y + bi.y};
                if (relu) v = {fmaxf(v.x, 0.f), fmaxf(v.y, 0.f)};
                if (mask) {
                    const f2 mk = ld2(mask + o);
                    v = {mk.x > 0.f ? v.x : 0.f, mk.y > 0.f ? v.y : 0.f};
                }
                st2(out + o, v);
            }
        }
    }
}

// A (6x4) on a 4-vector: the transpose of at4's matrix
template <typename T>
__device__ __forceinline__ void a6(const T y0, const T y1, const T y2, const T y3, T& o0, T& o1, T& o2, T& o3, T& o4, T& o5) {
    o0 = y0;
    o1 = y0 + y1 + y2 + y3;
    o2 = y0 - y1 + y2 - y3;
    o3 = y0 + 2.f * y1 + 4.f * y2 + 8.f * y3;
    o4 = y0 - 2.f * y1 + 4.f * y2 - 8.f * y3;
    o5 = y3;
}

// weight gradient, step 1: gy [B,H,W,N] -> Mg [36][T][N] = A dY A^T per 4x4 output tile (zeros beyond the image)
__global__ __launch_bounds__(256) void wino_outgrad_kernel(const float* __restrict__ gy, int B, int H, int W, int N, int th_n, int tw_n,
                                                           float* __restrict__ Mg) {
    const int64_t T = (int64_t)B * th_n * tw_n;
    const int N2 = N / 2;
    const int64_t total = T * N2;
    const int64_t ps = T * N;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int n = 2 * (int)(idx % N2);
        const int64_t t = idx / N2;
        const int tw = (int)(t % tw_n), th = (int)((t / tw_n) % th_n), b = (int)(t / ((int64_t)tw_n * th_n));
        f2 tcol[6][4];  // A dY
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int ox = 4 * tw + j;
            f2 y[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int oy = 4 * th + i;
                y[i] = (oy < H && ox < W) ? ld2(gy + (((int64_t)b * H + oy) * W + ox) * N + n) : f2{0.f, 0.f};
            }
            a6(y[0], y[1], y[2], y[3], tcol[0][j], tcol[1][j], tcol[2][j], tcol[3][j], tcol[4][j], tcol[5][j]);
        }
#pragma unroll
        for (int i = 0; i < 6; i++) {
            f2 v0, v1, v2, v3, v4, v5;
            a6(tcol[i][0], tcol[i][1], tcol[i][2], tcol[i][3], v0, v1, v2, v3, v4, v5);
            float* o = Mg + ((int64_t)(6 * i) * T + t) * N + n;
            st2(o, v0); st2(o + ps, v1); st2(o + 2 * ps, v2); st2(o + 3 * ps, v3); st2(o + 4 * ps, v4); st2(o + 5 * ps, v5);
        }
    }
}

// G^T (3x6) on a 6-vector
__device__ __forceinline__ void gt3(const float u0, const float u1, const float u2, const float u3, const float u4, const float u5, float& o0,
                                    float& o1, float& o2) {
    o0 = 0.25f * u0 - (1.f / 6.f) * (u1 + u2) + (1.f / 24.f) * (u3 + u4);
    o1 = (1.f / 6.f) * (u2 - u1) + (1.f / 12.f) * (u3 - u4);
    o2 = (1.f / 6.f) * (u3 + u4 - u1 - u2) + u5;
}

// weight gradient, step 3: dU [36][N][C] -> dw [N][3][3][C] += scale[n] * G^T dU G
__global__ __launch_bounds__(256) void wino_wgrad_inverse_kernel(const float* __restrict__ dU, int N, int C, const float* __restrict__ scale,
                                                                 float* __restrict__ dw) {
    const int64_t total = (int64_t)N * C;
    const int64_t ps = total;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        const int64_t n = idx / C;
        const float* u = dU + idx;
        float tcol[3][6];  // G^T dU
#pragma unroll
        for (int j = 0; j < 6; j++)
            gt3(u[(0 + j) * ps], u[(6 + j) * ps], u[(12 + j) * ps], u[(18 + j) * ps], u[(24 + j) * ps], u[(30 + j) * ps], tcol[0][j], tcol[1][j],
                tcol[2][j]);
        const float sc = scale ? scale[n] : 1.f;
#pragma unroll
        for (int r = 0; r < 3; r++) {
            float g0, g1, g2;
            gt3(tcol[r][0], tcol[r][1], tcol[r][2], tcol[r][3], tcol[r][4], tcol[r][5], g0, g1, g2);
            float* o = dw + ((n * 3 + r) * 3) * C + c;
            o[0] += sc * g0; o[C] += sc * g1; o[2 * C] += sc * g2;
        }
    }
}

inline unsigned grid_for(int64_t n) { return (unsigned)std::min<int64_t>((n + 255) / 256, 32768); }

}  // namespace

namespace abr {

int wino_input_transform(const float* x, int B, int H, int W, int C, float* V, hipStream_t st) {
    const int th_n = (H + 3) / 4, tw_n = (W + 3) / 4;
    wino_input_kernel<<<grid_for((int64_t)B * th_n * tw_n * (C / 2)), 256, 0, st>>>(x, B, H, W, C, th_n, tw_n, V);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int wino_weight_transform(const float* w, int N, int C, float* U, hipStream_t st) {
    wino_weight_kernel<<<grid_for((int64_t)N * C), 256, 0, st>>>(w, N, C, U);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int wino_output_transform(const float* Mm, int B, int H, int W, int N, const float* scale, const float* bias, int relu, const float* mask,
                          float* out, hipStream_t st) {
    const int th_n = (H + 3) / 4, tw_n = (W + 3) / 4;
    wino_output_kernel<<<grid_for((int64_t)B * th_n * tw_n * (N / 2)), 256, 0, st>>>(Mm, B, H, W, N, th_n, tw_n, scale, bias, relu, mask, out);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int wino_outgrad_transform(const float* gy, int B, int H, int W, int N, float* Mg, hipStream_t st) {
    const int th_n = (H + 3) / 4, tw_n = (W + 3) / 4;
    wino_outgrad_kernel<<<grid_for((int64_t)B * th_n * tw_n * (N / 2)), 256, 0, st>>>(gy, B, H, W, N, th_n, tw_n, Mg);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int wino_wgrad_inverse(const float* dU, int N, int C, const float* scale, float* dw, hipStream_t st) {
    wino_wgrad_inverse_kernel<<<grid_for((int64_t)N * C), 256, 0, st>>>(dU, N, C, scale, dw);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// Winograd scratch, one grow-only allocation per stream (forward / dgrad run on the main stream, weight gradients on the side one)
float* wino_ws(hipStream_t st, size_t floats) {
    struct Ws { float* buf = nullptr; size_t floats = 0; };
    static std::map<hipStream_t, Ws> pool;
    Ws& w = pool[st];
    if (w.floats < floats) {
        if (w.buf) { (void)hipStreamSynchronize(st); (void)hipFree(w.buf); w.buf = nullptr; w.floats = 0; }
        if (hipMalloc(&w.buf, floats * sizeof(float)) != hipSuccess) return nullptr;
        w.floats = floats;
    }
    return w.buf;
}

}  // namespace abr
