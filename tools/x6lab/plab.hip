// bf16x6 FORWARD main loop with the ACTIVATION operand pre-split too (round 4, judge's item 1): does it pay in the weights-direct structure?
//
// Today (conv_igemm_x6w_kernel<128,128,1,4>): a workgroup fetches its 128 x 32 fp32 A tile, splits it into three bf16 planes in registers
// (~1.8 VALU per MFMA) and parks them in LDS (12 ds_write_b64 per thread and k-tile); every n-tile column repeats that for the same rows.
// Here the producer of A has written the exact split ONCE, in MFMA-fragment order (the layout x6_pack_kernel gives the weights):
//     chunk(mb, ks, pl) = 64 lanes x 16 B, lane l = row mb*32 + (l & 31), k = ks*16 + (l >> 5)*8 .. +8 of plane pl
//     byte address      = (((mb * K/16 + ks) * 3 + pl) * 64 + l) * 16
// and the consumer brings the 24 chunks of a k-tile into LDS by LDS-DMA (buffer_load_dwordx4 ... lds: 1 KB per wave-instruction, no VGPR
// staging, no split, no ds_write); a wave's A fragment is then a contiguous (conflict-free) ds_read_b128 at lane * 16.  Two LDS buffers of
// 24 KB, ONE barrier per k-tile.  Results must equal the in-kernel split's bit for bit (the split is exact, the product order unchanged).
//
//   hipcc --offload-arch=gfx950 -O3 -o plab plab.hip && ./plab [M N K]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int BM = 128, BN = 128, BKX = 32, LDX = 40;

// fp32 matrix [rows][K] -> fragment-packed bf16x3 planes
__global__ void pack_planes(const float* __restrict__ B, u32x4* __restrict__ Bp, int N, int K) {
    const int KS = K / 16;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
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

__device__ __forceinline__ float sub1(float x, float y) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
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

__device__ __forceinline__ void store_c(float* C, int N, int m_base, int n_base, int lane, const f32x16 (&acc)[4]) {
    const int l31 = lane & 31, lh = lane >> 5;
    for (int i = 0; i < 4; i++)
        for (int r = 0; r < 16; r++) C[(size_t)(m_base + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * N + n_base + l31] = acc[i][r];
}

__device__ __forceinline__ int tile_of_block() {
    const unsigned nblk = gridDim.x, q_ = nblk / 8, r_ = nblk % 8, xcd = blockIdx.x % 8, pos = blockIdx.x / 8;
    return (int)((xcd < r_ ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_) + pos);
}

// ---------------------------------------------------------------------------------------------------------------- the library's loop
// (conv_igemm_x6w_kernel<128,128,1,4,PLAIN>: four waves of 128 x 32, conflict-free store rows)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_split(const float* __restrict__ A, const u32x4* __restrict__ Bp,
                                                                                           float* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16* As = reinterpret_cast<__bf16*>(smem);   // [3][BM][LDX]
    const int tiles_n = N / BN;
    const int tile = tile_of_block();
    const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kq = tid & 7, srow = ((tid >> 3) & ~5) | (((tid >> 3) & 1) << 2) | ((tid >> 5) & 1);
    const int KS = K / 16;
    const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, (unsigned)((size_t)M * K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(Bp), 0, (unsigned)((size_t)N * K * 6), 0x00020000);
    unsigned ao[4];
    for (int i = 0; i < 4; i++) ao[i] = ((m0 + srow + 32 * i) * K + kq * 4) * 4u;
    const unsigned bo = (unsigned)((((size_t)(n0 / 32 + wave) * KS) * 3 * 64 + lane) * 16);
    u32x4 ra[4];
    u32x4 fbr[2][3];
    auto load_a = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; i++) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(ra_, (int)ao[i], kt * BKX * 4, 0);
    };
    auto store_a = [&]() {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            __bf16* dst = As + (srow + 32 * i) * LDX + kq * 4;
            uint2 o0, o1, o2;
            split_pair(__uint_as_float(ra[i].x), __uint_as_float(ra[i].y), o0.x, o1.x, o2.x);
            split_pair(__uint_as_float(ra[i].z), __uint_as_float(ra[i].w), o0.y, o1.y, o2.y);
            *reinterpret_cast<uint2*>(dst) = o0;
            *reinterpret_cast<uint2*>(dst + BM * LDX) = o1;
            *reinterpret_cast<uint2*>(dst + 2 * BM * LDX) = o2;
        }
    };
    auto load_b = [&](int kt, int u) {
        const int ks = kt * 2 + u;
#pragma unroll
        for (int p = 0; p < 3; p++) fbr[u][p] = __builtin_amdgcn_raw_buffer_load_b128(rb_, (int)bo, (ks * 3 + p) * 1024, 0);
    };
    f32x16 acc[4];
    for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    const int l31 = lane & 31, lh = lane >> 5;
    const __bf16* af = As + l31 * LDX + lh * 8;
    const int nk = K / BKX;
    auto compute_tile = [&](int kt_next) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            bf16x8 fa[4][3], fb[3];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int pl = 0; pl < 3; pl++) fa[i][pl] = *reinterpret_cast<const bf16x8*>(af + pl * BM * LDX + i * 32 * LDX + u * 16);
#pragma unroll
            for (int pl = 0; pl < 3; pl++) fb[pl] = *reinterpret_cast<const bf16x8*>(&fbr[u][pl]);
            constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; t++)
#pragma unroll
                for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][pa[t]], fb[pb[t]], acc[i], 0, 0, 0);
            if (kt_next < nk) load_b(kt_next, u);
            if (u == 0) __builtin_amdgcn_sched_barrier(0);
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
    store_c(C, N, m0, n0 + wave * 32, lane, acc);
}

// ---------------------------------------------------------------------------------------------------------------- A planes by LDS-DMA
// The compiler does not count asm memory operations: the DMA pieces of tile kt+1 are issued right BEHIND the point where the compiler has
// waited for the B fragments of step 0 (so its own vmcnt(N) never has to drain them early), and waited for explicitly before the barrier.
__device__ __forceinline__ void dma16(unsigned lds_addr, const i32x4 rsrc, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

template <int NBUF, int ISSUE>   // ISSUE 0: behind step 0's MFMAs; 1: at the top of the tile (before step 0)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_dma(const u32x4* __restrict__ Ap, const u32x4* __restrict__ Bp,
                                                                                         float* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];   // [NBUF][4 mb][2 u][3 pl][1 KB]
    const int tiles_n = N / BN;
    const int tile = tile_of_block();
    const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int KS = K / 16;
    const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(Bp), 0, (unsigned)((size_t)N * K * 6), 0x00020000);
    const i32x4 rap = {(int)(unsigned)(size_t)Ap, (int)(unsigned)((size_t)Ap >> 32), (int)(unsigned)((size_t)M * K * 6), 0x00020000};
    const unsigned bo = (unsigned)((((size_t)(n0 / 32 + wave) * KS) * 3 * 64 + lane) * 16);
    // wave w brings row block w of the tile: 6 chunks (2 steps x 3 planes) = 6 KB contiguous in HBM and in LDS
    const unsigned a_voff = (unsigned)((((size_t)(m0 / 32 + wave) * KS) * 3 * 64 + lane) * 16);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    u32x4 fbr[2][3];
    auto load_b = [&](int kt, int u) {
        const int ks = kt * 2 + u;
#pragma unroll
        for (int p = 0; p < 3; p++) fbr[u][p] = __builtin_amdgcn_raw_buffer_load_b128(rb_, (int)bo, (ks * 3 + p) * 1024, 0);
    };
    auto dma_tile = [&](int kt) {
        const unsigned dst = lds_base + (unsigned)((kt % NBUF) * 24576 + wave * 6144);
#pragma unroll
        for (int c = 0; c < 6; c++) dma16(dst + c * 1024, rap, a_voff, (unsigned)(kt * 6144 + c * 1024));
    };
    f32x16 acc[4];
    for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    const int nk = K / BKX;
    auto compute_tile = [&](int kt, int kt_next) {
        const unsigned char* buf = lds + (kt % NBUF) * 24576 + lane * 16;
#pragma unroll
        for (int u = 0; u < 2; u++) {
            bf16x8 fa[4][3], fb[3];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int pl = 0; pl < 3; pl++) fa[i][pl] = *reinterpret_cast<const bf16x8*>(buf + ((i * 2 + u) * 3 + pl) * 1024);
#pragma unroll
            for (int pl = 0; pl < 3; pl++) fb[pl] = *reinterpret_cast<const bf16x8*>(&fbr[u][pl]);
            constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; t++)
#pragma unroll
                for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][pa[t]], fb[pb[t]], acc[i], 0, 0, 0);
            if (ISSUE == 0 && u == 0 && kt + NBUF - 1 < nk) dma_tile(kt + NBUF - 1);
            if (kt_next < nk) load_b(kt_next, u);
            if (u == 0) __builtin_amdgcn_sched_barrier(0);
        }
    };
#pragma unroll
    for (int s = 0; s < NBUF - 1; s++) dma_tile(s);
    load_b(0, 0);
    load_b(0, 1);
    for (int kt = 0; kt < nk; kt++) {
        // tile kt's pieces: everything issued before the (NBUF - 2) youngest DMA tiles and the B loads behind them must be in
        if (NBUF == 2) {
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                  // queue: DMA(kt) x6, B(kt,0) x3, B(kt,1) x3
        } else {
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                 // one more DMA tile (6) may stay in flight
        }
        __syncthreads();
        if (ISSUE == 1 && kt + NBUF - 1 < nk) dma_tile(kt + NBUF - 1);
        __builtin_amdgcn_sched_barrier(0);
        compute_tile(kt, kt + 1);
    }
    store_c(C, N, m0, n0 + wave * 32, lane, acc);
}

// ---------------------------------------------------------------------------------------------------------------- ... and B by asm loads
// Every memory operation of the loop is an asm statement, so every wait is counted by hand: the DMA of tile kt + NBUF - 1 is issued at the top
// of tile kt and has NBUF - 1 whole tiles to land.
__device__ __forceinline__ void bload(u32x4& dst, const i32x4 rsrc, unsigned voff, unsigned soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
template <int N> __device__ __forceinline__ void bwait(u32x4& a, u32x4& b, u32x4& c) {
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N) : "memory");
}

template <int NBUF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_dma2(const u32x4* __restrict__ Ap, const u32x4* __restrict__ Bp,
                                                                                          float* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tiles_n = N / BN;
    const int tile = tile_of_block();
    const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int KS = K / 16;
    const i32x4 rap = {(int)(unsigned)(size_t)Ap, (int)(unsigned)((size_t)Ap >> 32), (int)(unsigned)((size_t)M * K * 6), 0x00020000};
    const i32x4 rbp = {(int)(unsigned)(size_t)Bp, (int)(unsigned)((size_t)Bp >> 32), (int)(unsigned)((size_t)N * K * 6), 0x00020000};
    const unsigned bo = (unsigned)((((size_t)(n0 / 32 + wave) * KS) * 3 * 64 + lane) * 16);
    const unsigned a_voff = (unsigned)((((size_t)(m0 / 32 + wave) * KS) * 3 * 64 + lane) * 16);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    u32x4 fbr[2][3];
    auto load_b = [&](int kt, int u) {
#pragma unroll
        for (int p = 0; p < 3; p++) bload(fbr[u][p], rbp, bo, (unsigned)(((kt * 2 + u) * 3 + p) * 1024));
    };
    auto dma_tile = [&](int kt) {
        const unsigned dst = lds_base + (unsigned)((kt % NBUF) * 24576 + wave * 6144);
#pragma unroll
        for (int c = 0; c < 6; c++) dma16(dst + c * 1024, rap, a_voff, (unsigned)(kt * 6144 + c * 1024));
    };
    f32x16 acc[4];
    for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    const int nk = K / BKX;
    auto step = [&](int kt, int u) {
        const unsigned char* buf = lds + (kt % NBUF) * 24576 + lane * 16;
        bf16x8 fa[4][3], fb[3];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int pl = 0; pl < 3; pl++) fa[i][pl] = *reinterpret_cast<const bf16x8*>(buf + ((i * 2 + u) * 3 + pl) * 1024);
#pragma unroll
        for (int pl = 0; pl < 3; pl++) fb[pl] = *reinterpret_cast<const bf16x8*>(&fbr[u][pl]);
        constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; t++)
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][pa[t]], fb[pb[t]], acc[i], 0, 0, 0);
    };
    // one k-tile; W0 / W1 = operations that may stay in flight at the waits for step 0's / step 1's B fragments
    auto tile_body = [&]<int WTOP, int W0, int W1, bool DMA, bool NEXTB>(int kt) {
        if (WTOP >= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WTOP < 0 ? 0 : WTOP) : "memory");
        __syncthreads();
        if (DMA) dma_tile(kt + NBUF - 1);
        bwait<W0>(fbr[0][0], fbr[0][1], fbr[0][2]);
        step(kt, 0);
        if (NEXTB) load_b(kt + 1, 0);
        bwait<W1>(fbr[1][0], fbr[1][1], fbr[1][2]);
        step(kt, 1);
        if (NEXTB) load_b(kt + 1, 1);
    };
#pragma unroll
    for (int s = 0; s < NBUF - 1; s++) dma_tile(s);
    load_b(0, 0);
    load_b(0, 1);
    if constexpr (NBUF == 2) {
        // top: DMA(kt) x6, B(kt,0) x3, B(kt,1) x3 -> vmcnt(6); after issuing DMA(kt+1): wait B(kt,0) with 9 younger; wait B(kt,1): B(kt,1), DMA x6, B(kt+1,0) x3 -> 9
        int kt = 0;
        for (; kt + 1 < nk; kt++) tile_body.template operator()<6, 9, 9, true, true>(kt);
        tile_body.template operator()<6, 3, 0, false, false>(kt);
    } else {
        // queue at the top of tile kt: DMA(kt+1) x6, B(kt,0) x3, B(kt,1) x3 (DMA(kt) landed before B(kt-1,1) did); + DMA(kt+2) x6 issued here
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // tile 0 only: DMA(0) x6, DMA(1) x6, B x6
        int kt = 0;
        for (; kt + 2 < nk; kt++) tile_body.template operator()<-1, 9, 9, true, true>(kt);
        tile_body.template operator()<-1, 3, 3, false, true>(kt);
        tile_body.template operator()<-1, 3, 0, false, false>(kt + 1);
    }
    store_c(C, N, m0, n0 + wave * 32, lane, acc);
}

static float time_it(const char* name, void (*launch)(void*), void* ctx, double flops, float base) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 10; i++) launch(ctx);
    hipEventRecord(a);
    for (int i = 0; i < 20; i++) launch(ctx);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 20;
    printf("  %-52s %8.4f ms  %6.1f TF-eq   %+7.4f ms vs in-kernel split\n", name, ms, flops / ms * 1e-9, base > 0 ? ms - base : 0.f);
    return ms;
}

struct Ctx { const float* A; const u32x4 *Ap, *Bp; float* C; int M, N, K; };
static void l_split(void* c_) { Ctx* c = (Ctx*)c_; k_split<<<(c->M / BM) * (c->N / BN), 256, sizeof(__bf16) * 3 * BM * LDX>>>(c->A, c->Bp, c->C, c->M, c->N, c->K); }
template <int NBUF> static void l_dma2(void* c_) { Ctx* c = (Ctx*)c_; k_dma2<NBUF><<<(c->M / BM) * (c->N / BN), 256, NBUF * 24576>>>(c->Ap, c->Bp, c->C, c->M, c->N, c->K); }
template <int NBUF, int ISSUE> static void l_dma(void* c_) { Ctx* c = (Ctx*)c_; k_dma<NBUF, ISSUE><<<(c->M / BM) * (c->N / BN), 256, NBUF * 24576>>>(c->Ap, c->Bp, c->C, c->M, c->N, c->K); }

static void run_shape(int M, int N, int K) {
    float *A, *B, *C, *C2; u32x4 *Ap, *Bp;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * N * 4); hipMalloc(&C2, (size_t)M * N * 4);
    hipMalloc(&Ap, (size_t)M * K * 6); hipMalloc(&Bp, (size_t)N * K * 6);
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    unsigned st = 12345u;
    for (auto& v : hA) { st = st * 1664525u + 1013904223u; v = ((int)(st >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }
    for (auto& v : hB) { st = st * 1664525u + 1013904223u; v = ((int)(st >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }
    hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    pack_planes<<<(unsigned)(((size_t)(N / 32) * (K / 16) * 64 + 255) / 256), 256>>>(B, Bp, N, K);
    pack_planes<<<(unsigned)(((size_t)(M / 32) * (K / 16) * 64 + 255) / 256), 256>>>(A, Ap, M, K);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_dma<2, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 24576);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_dma<2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 24576);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_dma<3, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 24576);
    printf("GEMM %d x %d x %d (random operands), %d workgroups\n", M, N, K, (M / BM) * (N / BN));
    Ctx c{A, Ap, Bp, C, M, N, K};
    const double fl = 2.0 * M * N * K;
    // bit-exactness of every variant against the in-kernel split
    std::vector<float> ref((size_t)M * N), got((size_t)M * N);
    l_split(&c); hipDeviceSynchronize();
    hipMemcpy(ref.data(), C, ref.size() * 4, hipMemcpyDeviceToHost);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_dma2<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 24576);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_dma2<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 24576);
    constexpr int NV = 5;
    void (*variants[NV])(void*) = {l_dma<2, 0>, l_dma<2, 1>, l_dma<3, 1>, l_dma2<2>, l_dma2<3>};
    const char* names[NV] = {"A planes by LDS-DMA, 2 buffers, issue behind step 0", "A planes by LDS-DMA, 2 buffers, issue at the top", "A planes by LDS-DMA, 3 buffers, issue at the top",
                             "... + asm B loads, counted waits, 2 buffers", "... + asm B loads, counted waits, 3 buffers"};
    for (int v = 0; v < NV; v++) {
        hipMemset(C, 0, (size_t)M * N * 4);
        variants[v](&c); hipDeviceSynchronize();
        hipMemcpy(got.data(), C, got.size() * 4, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t i = 0; i < got.size(); i++) bad += memcmp(&got[i], &ref[i], 4) != 0;
        printf("  %-52s %s (%zu of %zu elements differ)\n", names[v], bad ? "MISMATCH" : "bit-identical to the in-kernel split", bad, got.size());
    }
    time_it("(warm-up)", l_split, &c, fl, 0.f);
    time_it("(warm-up)", l_split, &c, fl, 0.f);
    const float b = time_it("in-kernel split (library loop)", l_split, &c, fl, 0.f);
    for (int v = 0; v < NV; v++) time_it(names[v], variants[v], &c, fl, b);
    time_it("in-kernel split (again)", l_split, &c, fl, b);
    for (int v = 0; v < NV; v++) time_it(names[v], variants[v], &c, fl, b);
    hipFree(A); hipFree(B); hipFree(C); hipFree(C2); hipFree(Ap); hipFree(Bp);
}

int main(int argc, char** argv) {
    if (argc > 3) { run_shape(atoi(argv[1]), atoi(argv[2]), atoi(argv[3])); return 0; }
    run_shape(32768, 2048, 1024);
    run_shape(32768, 2048, 512);
    run_shape(32768, 512, 2048);
    run_shape(2048 * 36, 512, 512);   // the 36 Winograd GEMMs of a layer4 3x3 (2048 tiles each), as one tall GEMM
    return 0;
}
