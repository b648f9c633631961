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

#include <algorithm>
#include <map>
#include <functional>
#include <mutex>
#include <utility>
#include <type_traits>
#include <vector>

#include "common.h"

namespace {

// The transforms work on V adjacent channels per thread: V = 4 (16 B loads / stores per lane, the width at which the memory pipe
// reaches its peak -- cdna_hip_programming.md) whenever the channel count allows, V = 2 otherwise.  A 6x6 tile of float4 is 144
// live VGPRs: two waves per SIMD, each with 36 independent 16 B loads in flight, which is plenty for a streaming kernel.
template <int V>
struct vf {
    float v[V];
};
template <int V>
__device__ __forceinline__ vf<V> operator+(vf<V> a, vf<V> b) {
    vf<V> r;
#pragma unroll
    for (int i = 0; i < V; i++) r.v[i] = a.v[i] + b.v[i];
    return r;
}
template <int V>
__device__ __forceinline__ vf<V> operator-(vf<V> a, vf<V> b) {
    vf<V> r;
#pragma unroll
    for (int i = 0; i < V; i++) r.v[i] = a.v[i] - b.v[i];
    return r;
}
template <int V>
__device__ __forceinline__ vf<V> operator-(vf<V> a) {
    vf<V> r;
#pragma unroll
    for (int i = 0; i < V; i++) r.v[i] = -a.v[i];
    return r;
}
template <int V>
__device__ __forceinline__ vf<V> operator*(float s, vf<V> a) {
    vf<V> r;
#pragma unroll
    for (int i = 0; i < V; i++) r.v[i] = s * a.v[i];
    return r;
}
template <int V>
__device__ __forceinline__ vf<V> vzero() {
    vf<V> r;
#pragma unroll
    for (int i = 0; i < V; i++) r.v[i] = 0.f;
    return r;
}
template <int V>
__device__ __forceinline__ vf<V> vld(const float* p);
template <>
__device__ __forceinline__ vf<2> vld<2>(const float* p) { const float2 t = *reinterpret_cast<const float2*>(p); return {{t.x, t.y}}; }
template <>
__device__ __forceinline__ vf<4> vld<4>(const float* p) { const float4 t = *reinterpret_cast<const float4*>(p); return {{t.x, t.y, t.z, t.w}}; }
template <int V>
__device__ __forceinline__ vf<V> vld_buf(__amdgpu_buffer_rsrc_t r, unsigned byte_off);
template <>
__device__ __forceinline__ vf<2> vld_buf<2>(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_off, 0, 0);
    return {{__uint_as_float(t.x), __uint_as_float(t.y)}};
}
template <>
__device__ __forceinline__ vf<4> vld_buf<4>(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
    return {{__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w)}};
}
__device__ __forceinline__ void vst(float* p, vf<2> a) { *reinterpret_cast<float2*>(p) = make_float2(a.v[0], a.v[1]); }
__device__ __forceinline__ void vst(float* p, vf<4> a) { *reinterpret_cast<float4*>(p) = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]); }

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

// grid-stride over (tile, channel group); consecutive threads = consecutive channel groups (coalesced 1 KB per wave at V = 4)
template <int V>
__device__ __forceinline__ unsigned vabs_max(unsigned m, vf<V> a) {
#pragma unroll
    for (int i = 0; i < V; i++) m = max(m, __float_as_uint(a.v[i]) & 0x7FFFFFFFu);
    return m;
}

// amax / epoch (optional): max |V| goes to that amax word (one atomic per workgroup): the f16x3 GEMMs scale their A operand by it
template <int V>
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, int B, int H, int W, int C, int th_n, int tw_n,
                                                         float* __restrict__ Vo, unsigned long long* amax, unsigned epoch) {
    unsigned am = 0;
    typedef vf<V> f2;
    const int64_t T = (int64_t)B * th_n * tw_n;
    const int C2 = C / V;
    const int64_t total = T * C2;
    const int64_t ps = T * C;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (unsigned)((int64_t)B * H * W * C * 4), 0x00020000);   // (< 2 GB: checked by the caller)
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = V * (int)(idx % C2);
        const int64_t t = idx / C2;
        const int tw = (int)(t % tw_n), th = (int)((t / tw_n) % th_n), b = (int)(t / ((int64_t)tw_n * th_n));
        const int y0 = 4 * th - 1, x0 = 4 * tw - 1;
        f2 tcol[6][6];  // B^T d, built column by column so that only one input column is live at a time
        // the 36 taps come through BUFFER loads: a 32-bit offset against a range-checked descriptor, halo taps get an out-of-range offset
        // and read as zeros in hardware -- no exec-masked block and no 64-bit address chain per tap (as in the conv kernels' gathers)
        const int base = (b * H * W) * C + c;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const int xx = x0 + j;
            const bool xin = (unsigned)xx < (unsigned)W;
            f2 d[6];
#pragma unroll
            for (int i = 0; i < 6; i++) {
                const int y = y0 + i;
                const bool in = xin & ((unsigned)y < (unsigned)H);
                d[i] = vld_buf<V>(rx, in ? (unsigned)(base + (y * W + xx) * C) * 4u : 0x80000000u);
            }
            bt6(d[0], d[1], d[2], d[3], d[4], d[5], tcol[0][j], tcol[1][j], tcol[2][j], tcol[3][j], tcol[4][j], tcol[5][j]);
        }
#pragma unroll
        for (int i = 0; i < 6; i++) {  // (B^T d) B : the same combination along the row
            f2 v0, v1, v2, v3, v4, v5;
            bt6(tcol[i][0], tcol[i][1], tcol[i][2], tcol[i][3], tcol[i][4], tcol[i][5], v0, v1, v2, v3, v4, v5);
            float* o = Vo + ((int64_t)(6 * i) * T + t) * C + c;
            vst(o, v0); vst(o + ps, v1); vst(o + 2 * ps, v2); vst(o + 3 * ps, v3); vst(o + 4 * ps, v4); vst(o + 5 * ps, v5);
            if (amax) am = vabs_max(vabs_max(vabs_max(vabs_max(vabs_max(vabs_max(am, v0), v1), v2), v3), v4), v5);
        }
    }
    if (amax) abr::h3_amax_emit(amax, epoch, am);
}

// U = G g G^T, one row I of it (6 values), with EVERY rounding spelled out (contraction off, explicit fma): the same U has to come out of every
// kernel that forms it -- the fp32 transforms (wino_weight_*_kernel) and the kernels that go from w straight to packed f16x3 planes
// (wino_h3_*_multi_kernel) -- and left to itself the compiler fuses (1/24) a -+ (1/12) b + (1/6) c and the sums of products differently from one
// kernel to the next (seen in round 6: planes differing in rows 24..35 between two kernels built from one expression).  The expressions are the
// ones rounds 2-5's build evaluated (read off its code), so that every result of those rounds is reproduced bit for bit:
//   G row 0: 1/4 a          rows 1, 2: -1/6 ((a +- b) + c)          rows 3, 4: fma(c, 1/6, fma(a, 1/24, +-(b / 12)))          row 5: c
// along r (t = G g), then the same along q (u = t G^T), except that where t is itself a product k s (rows 0..2) the sums of u1 / u2 take the
// second and third terms unrounded: fma(s2, k, fma(+-s1, k, t0)).
template <int I, int V>
__device__ __forceinline__ void wino_u_row(const vf<V> (&g)[3][3], vf<V> (&u)[6]) {
#pragma clang fp contract(off)
    constexpr float k6 = -1.f / 6.f, c24 = 1.f / 24.f, c12 = 1.f / 12.f, c6 = 1.f / 6.f;
    constexpr float k = I == 0 ? 0.25f : k6;
#pragma unroll
    for (int e = 0; e < V; e++) {
        float t[3], s[3];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const float a = g[0][q].v[e], b = g[1][q].v[e], c = g[2][q].v[e];
            if constexpr (I == 0) { s[q] = a; t[q] = 0.25f * a; }
            else if constexpr (I == 1) { s[q] = (a + b) + c; t[q] = k6 * s[q]; }
            else if constexpr (I == 2) { s[q] = (a - b) + c; t[q] = k6 * s[q]; }
            else if constexpr (I == 3) { s[q] = 0.f; t[q] = __builtin_fmaf(c, c6, __builtin_fmaf(a, c24, b * c12)); }
            else if constexpr (I == 4) { s[q] = 0.f; t[q] = __builtin_fmaf(c, c6, __builtin_fmaf(a, c24, -(b * c12))); }
            else { s[q] = 0.f; t[q] = c; }
        }
        u[0].v[e] = 0.25f * t[0];
        if constexpr (I < 3) {
            u[1].v[e] = k6 * __builtin_fmaf(s[2], k, __builtin_fmaf(s[1], k, t[0]));
            u[2].v[e] = k6 * __builtin_fmaf(s[2], k, __builtin_fmaf(-s[1], k, t[0]));
        } else {
            u[1].v[e] = k6 * ((t[0] + t[1]) + t[2]);
            u[2].v[e] = k6 * ((t[0] - t[1]) + t[2]);
        }
        const float m = t[1] * c12;
        u[3].v[e] = __builtin_fmaf(t[2], c6, __builtin_fmaf(t[0], c24, m));
        u[4].v[e] = __builtin_fmaf(t[2], c6, __builtin_fmaf(t[0], c24, -m));
        u[5].v[e] = t[2];
    }
}
// f(integral_constant<int, 0>) ... f(integral_constant<int, 5>)
template <typename F>
__device__ __forceinline__ void for_rows6(F&& f) {
    f(std::integral_constant<int, 0>{}); f(std::integral_constant<int, 1>{}); f(std::integral_constant<int, 2>{});
    f(std::integral_constant<int, 3>{}); f(std::integral_constant<int, 4>{}); f(std::integral_constant<int, 5>{});
}

// w [N][3][3][C] (OHWI) -> U [36][N][C]; V input channels per thread
template <int V>
__global__ __launch_bounds__(256) void wino_weight_kernel(const float* __restrict__ w, int N, int C, float* __restrict__ U) {
    typedef vf<V> fv;
    const int Cv = C / V;
    const int64_t total = (int64_t)N * Cv;
    const int64_t ps = (int64_t)N * C;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = V * (int)(idx % Cv);
        const int64_t n = idx / Cv;
        fv g[3][3];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int q = 0; q < 3; q++) g[r][q] = vld<V>(w + ((n * 3 + r) * 3 + q) * C + c);
        for_rows6([&](auto I) {
            constexpr int i = decltype(I)::value;
            fv u[6];
            wino_u_row<i>(g, u);
            float* o = U + ((int64_t)(6 * i) * N + n) * C + c;
            vst(o, u[0]); vst(o + ps, u[1]); vst(o + 2 * ps, u[2]); vst(o + 3 * ps, u[3]); vst(o + 4 * ps, u[4]); vst(o + 5 * ps, u[5]);
        });
    }
}

// the same for a table of weight tensors (abr_conv_prepare_batch): workgroup -> (job, 256 (n, c4) items of it)
__global__ __launch_bounds__(256) void wino_weight_multi_kernel(const abr::PrepJob* __restrict__ jobs, int njobs) {
    typedef vf<4> fv;
    const int j = abr::prep_find_job(jobs, njobs, blockIdx.x);
    const abr::PrepJob jb = jobs[j];
    const float* w = jb.src;
    float* U = reinterpret_cast<float*>(jb.dst);
    const int N = jb.a, C = jb.b, Cv = C / 4;
    const int64_t total = (int64_t)N * Cv, ps = (int64_t)N * C;
    const int64_t idx = (int64_t)(blockIdx.x - jb.first_block) * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = 4 * (int)(idx % Cv);
    const int64_t n = idx / Cv;
    fv g[3][3];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int q = 0; q < 3; q++) g[r][q] = vld<4>(w + ((n * 3 + r) * 3 + q) * C + c);
    for_rows6([&](auto I) {
        constexpr int i = decltype(I)::value;
        fv u[6];
        wino_u_row<i>(g, u);
        float* o = U + ((int64_t)(6 * i) * N + n) * C + c;
        vst(o, u[0]); vst(o + ps, u[1]); vst(o + 2 * ps, u[2]); vst(o + 3 * ps, u[3]); vst(o + 4 * ps, u[4]); vst(o + 5 * ps, u[5]);
    });
}

// ---- f16x3: Winograd-domain weights straight into the packed planes (round 6) ---------------------------------------------------------------
// abr_conv_prepare_batch used to go  w -> U (fp32, 4x the weights' size, written) -> row amax (U read) -> pack (U read again, planes written):
// 3.4 GB of HBM traffic per optimiser step for the step's 20.6 M Winograd-domain weights and their dgrad copies, on a stream of its own but NOT
// hidden -- with the preparation knocked out the step is 0.86 ms shorter (MEASUREMENTS.md).  The two kernels below never materialise U: pass A
// computes every U row's amax from w (one wave per output channel), pass B recomputes U from a w tile staged in LDS and writes the two fp16 planes
// in MFMA-fragment order directly.  w (1/4 of U) is read twice, the planes are written once.  Same G g G^T arithmetic (wino_u_row), same row scales,
// same split as wino_weight_multi_kernel + h3_rowscale_multi_kernel + h3_pack_multi_kernel: bit-identical planes
// (tests/test_gpu_prep_batch.py compares against the self-contained path).  jobs[j]: src = w [N][3][3][C], dst = planes of the [36 N][C] matrix with
// its 36 N row scales behind them, a = N (% 32 == 0), b = C (% 64 == 0), c = first workgroup in the scale launch (N / 4 workgroups),
// first_block / gx = C / 64 / gy = N / 32 in the pack launch.
__device__ __forceinline__ int prep_find_job_by_c(const abr::PrepJob* jobs, int njobs, int block) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].c <= block) lo = mid; else hi = mid - 1;
    }
    return lo;
}
__device__ __forceinline__ size_t h3_planes_bytes_of(int rows, int K) { return (size_t)((rows + 31) / 32 * 32) * (size_t)K * 4; }

// pass A: the scale of every U row.  One WAVE per output channel n: its lanes walk the input channels (4 per lane, 256 per trip), keep the 36
// running maxima of |U| in registers, reduce them across the wave once and lanes 0..35 write the rows' scales (h3_scales of the amax bits) behind the
// planes -- no atomics, no zeroed scratch.  .c = the job's first workgroup in this launch (4 output channels per workgroup).
__global__ __launch_bounds__(256, 3) void wino_h3_scales_multi_kernel(const abr::PrepJob* __restrict__ jobs, int njobs, unsigned* flags) {
    typedef vf<4> fv;
    const int j = prep_find_job_by_c(jobs, njobs, blockIdx.x);
    const abr::PrepJob jb = jobs[j];
    const float* w = jb.src;
    const int N = jb.a, C = jb.b;
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)(blockIdx.x - jb.c) * 4 + (threadIdx.x >> 6);    // N % 32 == 0: always inside
    float* scales = reinterpret_cast<float*>(reinterpret_cast<char*>(jb.dst) + h3_planes_bytes_of(36 * N, C));
    unsigned mx[36];
#pragma unroll
    for (int k = 0; k < 36; k++) mx[k] = 0u;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    for (int c = 4 * lane; c < C; c += 256) {
        fv g[3][3];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int q = 0; q < 3; q++) g[r][q] = vld<4>(w + ((n * 3 + r) * 3 + q) * C + c);
        for_rows6([&](auto I) {
            constexpr int i = decltype(I)::value;
            fv u[6];
            wino_u_row<i>(g, u);
#pragma unroll
            for (int jj = 0; jj < 6; jj++) mx[6 * i + jj] = vabs_max(mx[6 * i + jj], u[jj]);
        });
    }
    // across the wave through LDS: [k][lane] written by everyone, row k read back by lane k (16 x 16 B)
    __shared__ __attribute__((aligned(16))) unsigned red_[4][36][64];
    unsigned (*red)[64] = red_[threadIdx.x >> 6];
#pragma unroll
    for (int k = 0; k < 36; k++) red[k][lane] = mx[k];
    __builtin_amdgcn_wave_barrier();
    unsigned mine = 0u;
    if (lane < 36) {
        const uint4* rowp = reinterpret_cast<const uint4*>(red[lane]);
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const uint4 v = rowp[q];
            mine = max(max(mine, max(v.x, v.y)), max(v.z, v.w));
        }
    }
    if (lane < 36) {
        float sc, inv;
        abr::h3_scales(mine, sc, inv);
        scales[(size_t)lane * N + n] = sc;
        if (flags && (mine >> 23) >= 255u) atomicOr(flags, ABR_X6_FLAG_NONFINITE);
    }
}

constexpr int kWhPitch = 9 * 64 + 4;                                  // floats per row of the staged w tile (+4: conflict-free ds_read_b128 over 32 rows)
constexpr size_t kWhLds = sizeof(float) * 32 * kWhPitch;              // 74 240 B: two workgroups per CU

__global__ __launch_bounds__(256) void wino_h3_pack_multi_kernel(const abr::PrepJob* __restrict__ jobs, int njobs) {
    typedef vf<4> fv;
    extern __shared__ __attribute__((aligned(16))) float wt_[];
    const int j = abr::prep_find_job(jobs, njobs, blockIdx.x);
    const abr::PrepJob jb = jobs[j];
    const float* w = jb.src;
    const int N = jb.a, C = jb.b;
    const int lb = blockIdx.x - jb.first_block;
    const int nblk = lb / jb.gx, kc = lb % jb.gx;                     // 32 output channels x 64 input channels
    const int n0 = nblk * 32, k0 = kc * 64;
    const int tid = threadIdx.x;
    // the w tile [32 rows][9 taps][64 channels], coalesced (256 B per row and tap)
#pragma unroll
    for (int it = 0; it < 18; it++) {
        const int f = tid + 256 * it, rt = f >> 4, c4 = (f & 15) * 4;
        const int row = rt / 9, tap = rt - row * 9;
        *reinterpret_cast<float4*>(wt_ + row * kWhPitch + tap * 64 + c4) =
            *reinterpret_cast<const float4*>(w + (((int64_t)(n0 + row)) * 9 + tap) * C + k0 + c4);
    }
    __syncthreads();
    // lane = (row, 8 consecutive channels): exactly the element set of its 16 B in a packed chunk (h3_pack_chunk's mapping)
    const int wave = tid >> 6, lane = tid & 63, row = lane & 31, k8 = wave * 16 + (lane >> 5) * 8;
    const float* src = wt_ + row * kWhPitch + k8;
    fv gA[3][3], gB[3][3];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const float4 a = *reinterpret_cast<const float4*>(src + (r * 3 + q) * 64), b = *reinterpret_cast<const float4*>(src + (r * 3 + q) * 64 + 4);
            gA[r][q] = {{a.x, a.y, a.z, a.w}};
            gB[r][q] = {{b.x, b.y, b.z, b.w}};
        }
    uint4* planes = reinterpret_cast<uint4*>(jb.dst);
    const float* scales = reinterpret_cast<const float*>(reinterpret_cast<const char*>(jb.dst) + h3_planes_bytes_of(36 * N, C));   // pass A's
    const int ks = kc * 4 + wave, KS = C / 16;
    auto emit = [&](int xi, const fv& uA, const fv& uB, float sc) {
        const float inv = 1.f / sc;   // (a power of two: exact)
        unsigned h0[4], h1[4];
        abr::h3_split2(uA.v[0], uA.v[1], inv, h0[0], h1[0]);
        abr::h3_split2(uA.v[2], uA.v[3], inv, h0[1], h1[1]);
        abr::h3_split2(uB.v[0], uB.v[1], inv, h0[2], h1[2]);
        abr::h3_split2(uB.v[2], uB.v[3], inv, h0[3], h1[3]);
        const size_t nb = ((size_t)xi * N + n0) / 32;
        const size_t ch = (nb * KS + ks) * 2;
        planes[ch * 64 + lane] = make_uint4(h0[0], h0[1], h0[2], h0[3]);
        planes[(ch + 1) * 64 + lane] = make_uint4(h1[0], h1[1], h1[2], h1[3]);
    };
    auto row_i = [&](auto I) {
        constexpr int i = decltype(I)::value;
        float sc[6];
#pragma unroll
        for (int jj = 0; jj < 6; jj++) sc[jj] = scales[(size_t)(6 * i + jj) * N + n0 + row];
        fv uA[6], uB[6];
        wino_u_row<i>(gA, uA);
        wino_u_row<i>(gB, uB);
#pragma unroll
        for (int jj = 0; jj < 6; jj++) emit(6 * i + jj, uA[jj], uB[jj], sc[jj]);
    };
    for_rows6(row_i);
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

// M [36][T][N] -> out [B,H,W,N] with the conv epilogue (scale, bias, ReLU, ReLU mask of the producer); V channels per thread
template <int V>
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ Mm, int B, int H, int W, int N, int th_n, int tw_n,
                                                          const float* __restrict__ scale, const float* __restrict__ bias, int relu,
                                                          const float* __restrict__ mask, float* __restrict__ out, unsigned long long* amax, unsigned epoch) {
    typedef vf<V> f2;
    unsigned am = 0;
    const int64_t T = (int64_t)B * th_n * tw_n;
    const int N2 = N / V;
    const int64_t total = T * N2;
    const int64_t ps = T * N;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int n = V * (int)(idx % N2);
        const int64_t t = idx / N2;
        const int tw = (int)(t % tw_n), th = (int)((t / tw_n) % th_n), b = (int)(t / ((int64_t)tw_n * th_n));
        const float* m = Mm + t * N + n;
        f2 tcol[4][6];  // A^T M
#pragma unroll
        for (int j = 0; j < 6; j++)
            at4(vld<V>(m + (0 + j) * ps), vld<V>(m + (6 + j) * ps), vld<V>(m + (12 + j) * ps), vld<V>(m + (18 + j) * ps), vld<V>(m + (24 + j) * ps),
                vld<V>(m + (30 + j) * ps), tcol[0][j], tcol[1][j], tcol[2][j], tcol[3][j]);
        f2 sc, bi;
#pragma unroll
        for (int e = 0; e < V; e++) { sc.v[e] = scale ? scale[n + e] : 1.f; bi.v[e] = bias ? bias[n + e] : 0.f; }
        // the ReLU mask of the producer (dgrad): all 16 mask vectors of the tile are requested BEFORE the first is used (pixels past the
        // image read the tile's first pixel, which always exists, and are not stored) -- inside the store loop each mask load was followed
        // by its own wait, 16 dependent round trips per tile
        f2 mk[4][4];
        if (mask) {
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int oy = 4 * th + i, ox = 4 * tw + j;
                    const bool in = oy < H && ox < W;
                    mk[i][j] = vld<V>(mask + (((int64_t)b * H + (in ? oy : 4 * th)) * W + (in ? ox : 4 * tw)) * N + n);
                }
        }
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
                f2 v;
#pragma unroll
                for (int e = 0; e < V; e++) {
                    v.v[e] = y[j].v[e] * sc.v[e] + bi.v[e];
                    if (relu) v.v[e] = abr::relu_f(v.v[e]);
                }
                if (mask) {
#pragma unroll
                    for (int e = 0; e < V; e++) v.v[e] = mk[i][j].v[e] > 0.f ? v.v[e] : 0.f;
                }
                vst(out + o, v);
                if (amax) am = vabs_max(am, v);
            }
        }
    }
    if (amax) abr::h3_amax_emit(amax, epoch, am);
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
template <int V>
__global__ __launch_bounds__(256) void wino_outgrad_kernel(const float* __restrict__ gy, int B, int H, int W, int N, int th_n, int tw_n,
                                                           float* __restrict__ Mg, unsigned long long* amax, unsigned epoch) {
    typedef vf<V> f2;
    unsigned am = 0;
    const int64_t T = (int64_t)B * th_n * tw_n;
    const int N2 = N / V;
    const int64_t total = T * N2;
    const int64_t ps = T * N;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int n = V * (int)(idx % N2);
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
                y[i] = (oy < H && ox < W) ? vld<V>(gy + (((int64_t)b * H + oy) * W + ox) * N + n) : vzero<V>();
            }
            a6(y[0], y[1], y[2], y[3], tcol[0][j], tcol[1][j], tcol[2][j], tcol[3][j], tcol[4][j], tcol[5][j]);
        }
#pragma unroll
        for (int i = 0; i < 6; i++) {
            f2 v0, v1, v2, v3, v4, v5;
            a6(tcol[i][0], tcol[i][1], tcol[i][2], tcol[i][3], v0, v1, v2, v3, v4, v5);
            float* o = Mg + ((int64_t)(6 * i) * T + t) * N + n;
            vst(o, v0); vst(o + ps, v1); vst(o + 2 * ps, v2); vst(o + 3 * ps, v3); vst(o + 4 * ps, v4); vst(o + 5 * ps, v5);
            if (amax) am = vabs_max(vabs_max(vabs_max(vabs_max(vabs_max(vabs_max(am, v0), v1), v2), v3), v4), v5);
        }
    }
    if (amax) abr::h3_amax_emit(amax, epoch, am);
}

// G^T (3x6) on a 6-vector
template <typename T>
__device__ __forceinline__ void gt3(const T u0, const T u1, const T u2, const T u3, const T u4, const T u5, T& o0, T& o1, T& o2) {
    o0 = 0.25f * u0 - (1.f / 6.f) * (u1 + u2) + (1.f / 24.f) * (u3 + u4);
    o1 = (1.f / 6.f) * (u2 - u1) + (1.f / 12.f) * (u3 - u4);
    o2 = (1.f / 6.f) * (u3 + u4 - u1 - u2) + u5;
}

// weight gradient, step 3: dU [36][N][C] -> dw [N][3][3][C] += scale[n] * G^T dU G; V input channels per thread
template <int V>
__global__ __launch_bounds__(256) void wino_wgrad_inverse_kernel(const float* __restrict__ dU, int N, int C, const float* __restrict__ scale,
                                                                 float* __restrict__ dw) {
    typedef vf<V> fv;
    const int Cv = C / V;
    const int64_t total = (int64_t)N * Cv;
    const int64_t ps = (int64_t)N * C;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = V * (int)(idx % Cv);
        const int64_t n = idx / Cv;
        const float* u = dU + n * C + c;
        fv tcol[3][6];  // G^T dU
#pragma unroll
        for (int j = 0; j < 6; j++)
            gt3(vld<V>(u + (0 + j) * ps), vld<V>(u + (6 + j) * ps), vld<V>(u + (12 + j) * ps), vld<V>(u + (18 + j) * ps), vld<V>(u + (24 + j) * ps),
                vld<V>(u + (30 + j) * ps), tcol[0][j], tcol[1][j], tcol[2][j]);
        const float sc = scale ? scale[n] : 1.f;
        fv old[3][3];   // dw += : the nine current values are requested together, ahead of the arithmetic (not one read-modify-write at a time)
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int q = 0; q < 3; q++) old[r][q] = vld<V>(dw + ((n * 3 + r) * 3 + q) * C + c);
#pragma unroll
        for (int r = 0; r < 3; r++) {
            fv g0, g1, g2;
            gt3(tcol[r][0], tcol[r][1], tcol[r][2], tcol[r][3], tcol[r][4], tcol[r][5], g0, g1, g2);
            float* o = dw + ((n * 3 + r) * 3) * C + c;
            vst(o, old[r][0] + sc * g0); vst(o + C, old[r][1] + sc * g1); vst(o + 2 * C, old[r][2] + sc * g2);
        }
    }
}

inline unsigned grid_for(int64_t n) { return (unsigned)std::min<int64_t>((n + 255) / 256, 32768); }
inline unsigned grid_for(int64_t n, bool) { return grid_for(n); }   // (a smaller grid for the launches that feed an amax word changed nothing: 18.62 vs 18.56 ms)

}  // namespace

namespace abr {

// V = 4 needs C % 4 == 0 (16 B alignment of every row of every tensor involved: all of them have C as their innermost pitch) and is
// only taken when it still leaves >= 4 workgroups per CU: the small layers (layer3: 640 tiles x 256 channels) are bound by launch
// latency / parallelism, not by access width, and run better with twice the threads
static inline bool wide(int64_t tiles, int C) {
    static const int force = getenv("ABR_WINO_VEC") ? atoi(getenv("ABR_WINO_VEC")) : 0;
    if (force == 2) return false;
    if (force == 4) return C % 4 == 0;
    return C % 4 == 0 && tiles * (C / 4) >= (int64_t)256 * 4 * 256;
}

int wino_input_transform(const float* x, int B, int H, int W, int C, float* V, hipStream_t st, const AmaxRef* amax) {
    const int th_n = (H + 3) / 4, tw_n = (W + 3) / 4;
    unsigned long long* aw = amax ? amax->word : nullptr;
    const unsigned ae = amax ? amax->epoch : 0u;
    // (measured: the input transform is never faster with 16 B accesses -- 0.95 ms / step at V = 2 against 1.07 mixed and 1.13 at V = 4)
    static const bool in4 = getenv("ABR_WINO_VEC") && atoi(getenv("ABR_WINO_VEC")) == 4;
    if (in4 && C % 4 == 0) wino_input_kernel<4><<<grid_for((int64_t)B * th_n * tw_n * (C / 4), aw != nullptr), 256, 0, st>>>(x, B, H, W, C, th_n, tw_n, V, aw, ae);
    else wino_input_kernel<2><<<grid_for((int64_t)B * th_n * tw_n * (C / 2), aw != nullptr), 256, 0, st>>>(x, B, H, W, C, th_n, tw_n, V, aw, ae);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int wino_weight_transform(const float* w, int N, int C, float* U, hipStream_t st) {
    if (C % 4 == 0) wino_weight_kernel<4><<<grid_for((int64_t)N * (C / 4)), 256, 0, st>>>(w, N, C, U);
    else wino_weight_kernel<2><<<grid_for((int64_t)N * (C / 2)), 256, 0, st>>>(w, N, C, U);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int prep_wino_h3_direct_multi(const PrepJob* jobs_dev, int njobs, int pack_blocks, int scale_blocks, unsigned* flags, hipStream_t st) {
    if (njobs <= 0 || pack_blocks <= 0) return 0;
    static std::once_flag attr_once;   // (several host threads may prepare weights: one attribute call, seen by all)
    std::call_once(attr_once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino_h3_pack_multi_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWhLds);
    });
    wino_h3_scales_multi_kernel<<<(unsigned)scale_blocks, 256, 0, st>>>(jobs_dev, njobs, flags);
    wino_h3_pack_multi_kernel<<<(unsigned)pack_blocks, 256, kWhLds, st>>>(jobs_dev, njobs);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int prep_wino_u_multi(const PrepJob* jobs_dev, int njobs, int blocks, hipStream_t st) {
    if (njobs <= 0 || blocks <= 0) return 0;
    wino_weight_multi_kernel<<<(unsigned)blocks, 256, 0, st>>>(jobs_dev, njobs);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int wino_output_transform(const float* Mm, int B, int H, int W, int N, const float* scale, const float* bias, int relu, const float* mask,
                          float* out, hipStream_t st, const AmaxRef* amax) {
    const int th_n = (H + 3) / 4, tw_n = (W + 3) / 4;
    unsigned long long* aw = amax ? amax->word : nullptr;
    const unsigned ae = amax ? amax->epoch : 0u;
    if (wide((int64_t)B * th_n * tw_n, N))
        wino_output_kernel<4><<<grid_for((int64_t)B * th_n * tw_n * (N / 4), aw != nullptr), 256, 0, st>>>(Mm, B, H, W, N, th_n, tw_n, scale, bias, relu, mask, out, aw, ae);
    else
        wino_output_kernel<2><<<grid_for((int64_t)B * th_n * tw_n * (N / 2), aw != nullptr), 256, 0, st>>>(Mm, B, H, W, N, th_n, tw_n, scale, bias, relu, mask, out, aw, ae);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int wino_outgrad_transform(const float* gy, int B, int H, int W, int N, float* Mg, hipStream_t st, const AmaxRef* amax) {
    const int th_n = (H + 3) / 4, tw_n = (W + 3) / 4;
    unsigned long long* aw = amax ? amax->word : nullptr;
    const unsigned ae = amax ? amax->epoch : 0u;
    if (wide((int64_t)B * th_n * tw_n, N)) wino_outgrad_kernel<4><<<grid_for((int64_t)B * th_n * tw_n * (N / 4), aw != nullptr), 256, 0, st>>>(gy, B, H, W, N, th_n, tw_n, Mg, aw, ae);
    else wino_outgrad_kernel<2><<<grid_for((int64_t)B * th_n * tw_n * (N / 2), aw != nullptr), 256, 0, st>>>(gy, B, H, W, N, th_n, tw_n, Mg, aw, ae);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int wino_wgrad_inverse(const float* dU, int N, int C, const float* scale, float* dw, hipStream_t st) {
    if (C % 4 == 0) wino_wgrad_inverse_kernel<4><<<grid_for((int64_t)N * (C / 4)), 256, 0, st>>>(dU, N, C, scale, dw);
    else wino_wgrad_inverse_kernel<2><<<grid_for((int64_t)N * (C / 2)), 256, 0, st>>>(dU, N, C, scale, dw);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// Winograd scratch, one grow-only allocation per stream (forward / dgrad run on the main stream, weight gradients on the side one)
float* wino_ws(hipStream_t st, size_t floats) {
    struct Ws { float* buf = nullptr; size_t floats = 0; };
    static std::map<hipStream_t, Ws> pool;
    static std::mutex mu;   // host threads may run Winograd convs for different streams concurrently
    std::lock_guard<std::mutex> g(mu);
    Ws& w = pool[st];
    if (w.floats < floats) {
        if (w.buf) { (void)hipStreamSynchronize(st); (void)hipFree(w.buf); w.buf = nullptr; w.floats = 0; }
        if (hipMalloc(&w.buf, floats * sizeof(float)) != hipSuccess) return nullptr;
        w.floats = floats;
    }
    return w.buf;
}

// Data DERIVED from a weight tensor, kept per (weight address, kind) under abr_conv_desc::w_version: the Winograd-domain weights U (fp32,
// for the fp32 MFMA kernels), the fragment-packed bf16x3 planes of a weight matrix, and the packed planes of U (both for the bf16x6
// weights-direct kernel).  One entry per (address, kind); a new version refills the same buffer.  Ordering: a hit from a stream other than
// the filling one waits for the fill's event; every stream that has read an entry is remembered, and a REFILL (new version) first makes
// its stream wait for an event recorded on each of those streams at refill time -- everything they had queued, the readers of the old
// bytes included, precedes it -- so no reader of version v can see version v+1's bytes whatever stream refills.  Bounded: least-recently-
// used entries are dropped when the cache exceeds ABR_WINO_CACHE_MB (default 4096: two full models need ~1.2 GB), and abr_conv_cache_clear() drops everything (the host
// calls it when a model's parameter storage is rebuilt or released).  One mutex guards the map.  Returns nullptr when there is no memory
// or the fill failed (callers then derive into scratch / split in-kernel).
namespace {
struct DEntry {
    void* buf = nullptr; size_t bytes = 0; int64_t version = 0; hipStream_t stream = nullptr; hipEvent_t filled = nullptr;
    std::vector<hipStream_t> readers;   // streams other than `stream` that have been handed this buffer since the last fill
    uint64_t last_use = 0;
    int64_t pending = 0;                // derived_acquire handed out a token for this version; derived_commit has not run yet
};
std::map<std::pair<const void*, int>, DEntry> g_dcache;
std::mutex g_dcache_mu;
uint64_t g_dcache_clock = 0;
size_t g_dcache_bytes = 0;

void dcache_drop(DEntry& e) {   // (mutex held) wait for every stream that may still read or write the buffer, then free it
    if (e.stream) (void)hipStreamSynchronize(e.stream);
    for (hipStream_t r : e.readers) (void)hipStreamSynchronize(r);
    if (e.buf) { (void)hipFree(e.buf); g_dcache_bytes -= e.bytes; }
    if (e.filled) (void)hipEventDestroy(e.filled);
    e = DEntry();
}
size_t dcache_limit() {
    static const size_t mb = getenv("ABR_WINO_CACHE_MB") ? (size_t)atoll(getenv("ABR_WINO_CACHE_MB")) : 4096;
    return mb << 20;
}
}  // namespace

// (mutex held) the entry for (w, kind) with a buffer of `bytes`; *needs_fill = it does not hold `version`; a refill is ordered behind the readers
// `waited` (a batch of acquires on one host thread, abr_conv_prepare_batch): streams `st` was already ordered behind DURING this batch -- one event per
// reader stream and batch instead of one per reader stream and entry (~150 event create / record / wait / destroy rounds per training step)
static DEntry* dcache_acquire_locked(const void* w, int kind, size_t bytes, int64_t version, hipStream_t st, bool* needs_fill, std::vector<hipStream_t>* waited = nullptr) {
    DEntry& e = g_dcache[std::make_pair(w, kind)];
    e.last_use = ++g_dcache_clock;
    if (e.buf && e.bytes != bytes) dcache_drop(e);   // the address now holds a different weight tensor
    if (!e.buf) {
        while (g_dcache_bytes + bytes > dcache_limit()) {   // evict least-recently-used entries (never this one, never one with a fill pending)
            auto victim = g_dcache.end();
            for (auto it = g_dcache.begin(); it != g_dcache.end(); ++it)
                if (it->second.buf && &it->second != &e && !it->second.pending && (victim == g_dcache.end() || it->second.last_use < victim->second.last_use)) victim = it;
            if (victim == g_dcache.end()) break;
            dcache_drop(victim->second);
            g_dcache.erase(victim);
        }
        if (hipMalloc(&e.buf, bytes) != hipSuccess) { e.buf = nullptr; return nullptr; }
        e.bytes = bytes;
        g_dcache_bytes += bytes;
        e.last_use = g_dcache_clock;
        if (!e.filled) (void)hipEventCreateWithFlags(&e.filled, hipEventDisableTiming);
    }
    if (e.version == version && !e.pending) {
        if (st != e.stream && std::find(e.readers.begin(), e.readers.end(), st) == e.readers.end()) {
            // first use of this fill on stream st: wait for it once -- st's later calls are ordered behind this wait by the stream itself
            // (`readers` is cleared by every refill)
            (void)hipStreamWaitEvent(st, e.filled, 0);
            e.readers.push_back(st);
        }
        *needs_fill = false;
        return &e;
    }
    // refill in place: order it behind everything the other streams that used this buffer (its previous filler included) have queued
    std::vector<hipStream_t> users = e.readers;
    if (e.stream && e.stream != st) users.push_back(e.stream);
    for (hipStream_t r : users) {
        if (r == st) continue;
        if (waited) {
            if (std::find(waited->begin(), waited->end(), r) != waited->end()) continue;
            waited->push_back(r);
        }
        hipEvent_t ev = nullptr;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipStreamSynchronize(r); continue; }
        (void)hipEventRecord(ev, r);
        (void)hipStreamWaitEvent(st, ev, 0);
        (void)hipEventDestroy(ev);   // (released by the runtime once it has completed)
    }
    e.readers.clear();
    e.version = 0;
    *needs_fill = true;
    return &e;
}

void* derived_cached(const void* w, int kind, size_t bytes, int64_t version, hipStream_t st, const std::function<int(void*)>& fill) {
    std::lock_guard<std::mutex> lock(g_dcache_mu);
    bool needs_fill = false;
    DEntry* e = dcache_acquire_locked(w, kind, bytes, version, st, &needs_fill);
    if (!e) return nullptr;
    if (!needs_fill) return e->buf;
    if (fill(e->buf)) return nullptr;
    (void)hipEventRecord(e->filled, st);
    e->version = version;
    e->stream = st;
    e->pending = 0;
    return e->buf;
}

void* derived_acquire(const void* w, int kind, size_t bytes, int64_t version, hipStream_t st, void** token, std::vector<hipStream_t>* waited) {
    std::lock_guard<std::mutex> lock(g_dcache_mu);
    *token = nullptr;
    bool needs_fill = false;
    DEntry* e = dcache_acquire_locked(w, kind, bytes, version, st, &needs_fill, waited);
    if (!e) return nullptr;
    if (needs_fill) {
        e->pending = version;
        *token = e;
    }
    return e->buf;
}

void derived_commit(void* const* tokens, int n, hipStream_t st) {
    std::lock_guard<std::mutex> lock(g_dcache_mu);
    for (int i = 0; i < n; i++) {
        DEntry* e = reinterpret_cast<DEntry*>(tokens[i]);
        if (!e) continue;
        (void)hipEventRecord(e->filled, st);
        e->version = e->pending;
        e->stream = st;
        e->pending = 0;
    }
}

void derived_abandon(void* const* tokens, int n) {
    std::lock_guard<std::mutex> lock(g_dcache_mu);
    for (int i = 0; i < n; i++) {
        DEntry* e = reinterpret_cast<DEntry*>(tokens[i]);
        if (!e) continue;
        e->pending = 0;
        e->version = 0;   // the buffer holds no complete version: the next lookup refills it
    }
}

float* wino_u_cached(const float* w, int N, int C, int64_t version, hipStream_t st) {
    return reinterpret_cast<float*>(derived_cached(w, DERIVED_WINO_U, (size_t)36 * N * C * sizeof(float), version, st,
                                                   [&](void* buf) { return wino_weight_transform(w, N, C, reinterpret_cast<float*>(buf), st); }));
}

// Entries with a fill token outstanding (derived_acquire .. derived_commit / derived_abandon on another thread) are left alone: their
// DEntry* is in that thread's hands.  They become droppable once committed.
void derived_cache_clear() {
    std::lock_guard<std::mutex> lock(g_dcache_mu);
    for (auto it = g_dcache.begin(); it != g_dcache.end();) {
        if (it->second.pending) { ++it; continue; }
        dcache_drop(it->second);
        it = g_dcache.erase(it);
    }
}
// drop what was derived from tensors inside [base, base + bytes): a model's parameter storage that is going away.  Other models' entries
// (and their in-flight conv calls on other host threads) are not touched.
void derived_cache_drop_range(const void* base, size_t bytes) {
    std::lock_guard<std::mutex> lock(g_dcache_mu);
    const char* lo = reinterpret_cast<const char*>(base);
    const char* hi = lo + bytes;
    for (auto it = g_dcache.begin(); it != g_dcache.end();) {
        const char* k = reinterpret_cast<const char*>(it->first.first);
        if (k < lo || k >= hi || it->second.pending) { ++it; continue; }
        dcache_drop(it->second);
        it = g_dcache.erase(it);
    }
}
size_t derived_cache_bytes() {
    std::lock_guard<std::mutex> lock(g_dcache_mu);
    return g_dcache_bytes;
}

}  // namespace abr
