// Shared helpers for libabr_iod_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <functional>
#include <vector>

#include "abr_iod_hip.h"

namespace abr {

// ReLU as torch computes it (at::relu = clamp_min(0): NaN in, NaN out).  fmaxf / v_max_f32 return the OTHER operand for a NaN, which would turn a
// diverged activation into a clean zero and hide it from the loss and from the range guard; the reference's run shows NaN (modeling/backbone/resnet.py:339-346).
__device__ __forceinline__ float relu_f(const float v) { return v < 0.f ? 0.f : v; }


void set_error(const char* fmt, ...);

#define ABR_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            ::abr::set_error(__VA_ARGS__);     \
            return ABR_E_INVALID;              \
        }                                      \
    } while (0)

#define ABR_CHECK_LAUNCH(name)                                                        \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) {                                                      \
            ::abr::set_error("%s: launch failed: %s", name, hipGetErrorString(e__));  \
            return ABR_E_LAUNCH;                                                      \
        }                                                                             \
    } while (0)

constexpr int kWave = 64;
constexpr int kNumXCD = 8;

// Bijective XCD-aware remap (guide T1): hardware places block b on XCD b % 8; give every XCD a contiguous
// chunk of the logical tile space so neighbouring tiles share one L2.  Speed only, never correctness.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
    const unsigned q = nblk / kNumXCD, r = nblk % kNumXCD;
    const unsigned xcd = bid % kNumXCD, pos = bid / kNumXCD;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + pos;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide sum for blockDim.x == 64*NW.  `sm` must hold NW floats.  Result valid in every thread.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* sm) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) sm[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NW; i++) t += sm[i];
    return t;
}

// Deterministic cross-workgroup sums (round 5: the step reproduces bit for bit).  Every workgroup parks its partial values in a scratch slot,
// takes a ticket, and the LAST one to arrive adds all partials up in slot order (strided over its threads, then a fixed tree): the result does
// not depend on which workgroup finishes last.  Replaces fp32 atomicAdd into the loss scalars / bias gradients, whose summation order -- and
// with it the last bits -- changed from run to run.  det_ws: per-stream ring of scratch (launches on a stream are ordered; a slot is reused
// thousands of launches later); .part == nullptr: no memory, the kernels fall back to atomics.
struct DetWs {
    float* part;
    unsigned* ticket;   // zero on entry, left zero
};
DetWs det_ws(hipStream_t st, size_t nfloats);
// vals: the workgroup's NV partial values, valid in thread 0.  Returns true in every thread of the last workgroup, with the totals in thread
// 0's `total`.  1-D or 2-D grids; blockDim.x a multiple of 64, <= 1024.
template <int NV>
__device__ __forceinline__ bool det_sum_last(const float (&vals)[NV], const DetWs ws, float (&total)[NV]) {
    __shared__ int s_last;
    __shared__ float s_w[16];
    const unsigned nblk = gridDim.x * gridDim.y, bid = blockIdx.y * gridDim.x + blockIdx.x;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int j = 0; j < NV; j++) __hip_atomic_store(ws.part + (size_t)bid * NV + j, vals[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the compiler may drop the wait behind the write-back: MI355X_MICROARCH.md)
        s_last = atomicAdd(ws.ticket, 1u) == nblk - 1u;
    }
    __syncthreads();
    if (!s_last) return false;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        *ws.ticket = 0u;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NV; j++) {
        float a = 0.f;
        for (unsigned i = threadIdx.x; i < nblk; i += blockDim.x) a += __hip_atomic_load(ws.part + (size_t)i * NV + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a = wave_sum(a);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = a;
        __syncthreads();
        float t = 0.f;
        for (unsigned w = 0; w < (blockDim.x >> 6); w++) t += s_w[w];
        total[j] = t;
    }
    return true;
}

// Optional per-launch timing with HIP events on the launch stream (bench.py's roofline leg).  Off by default.
enum ProfId { PROF_IGEMM_128x128 = 0, PROF_IGEMM_128x64, PROF_IGEMM_64x64, PROF_IGEMM_SMALLC, PROF_WGRAD, PROF_ROIALIGN_FWD,
              PROF_ROIALIGN_BWD, PROF_IGEMM_BF16, PROF_WGRAD_BF16,
              PROF_X6_128x128, PROF_X6_128x64, PROF_X6_64x64, PROF_X6W_128x128, PROF_X6W_128x64, PROF_X6W_64x64, PROF_X6W_TAIL64,
              PROF_H3W_128x128, PROF_H3W_128x64, PROF_H3W_64x64, PROF_WGRAD_H3, PROF_COUNT };   // (ids are positions in bench.py's PROF_NAMES)
// Winograd F(4x4,3x3) transform kernels (conv_winograd.hip); the batched GEMM between them is launched by conv_igemm.hip
struct AmaxRef;
// `amax` (optional, here and below): the kernel also writes max |value it stored| into that amax word (f16x3 consumers read it)
int wino_input_transform(const float* x, int B, int H, int W, int C, float* V, hipStream_t st, const AmaxRef* amax = nullptr);
int wino_weight_transform(const float* w, int N, int C, float* U, hipStream_t st);
int wino_output_transform(const float* M, int B, int H, int W, int N, const float* scale, const float* bias, int relu, const float* mask,
                          float* out, hipStream_t st, const AmaxRef* amax = nullptr);
int wino_outgrad_transform(const float* gy, int B, int H, int W, int N, float* Mg, hipStream_t st, const AmaxRef* amax = nullptr);
int wino_wgrad_inverse(const float* dU, int N, int C, const float* scale, float* dw, hipStream_t st);
float* wino_ws(hipStream_t st, size_t floats);
// cached Winograd-domain weights U [36][N][C] of the tensor at `w` (abr_conv_desc::w_version != 0), transformed on `st` when (w, version)
// has not been seen; nullptr = no memory (transform into scratch instead)
float* wino_u_cached(const float* w, int N, int C, int64_t version, hipStream_t st);
// the cache behind it, for every kind of data derived from a weight tensor (conv_winograd.hip): `fill(buf)` writes `bytes` on `st` (0 = ok)
enum DerivedKind { DERIVED_WINO_U = 0, DERIVED_X6_PLANES = 1, DERIVED_WINO_U_X6_PLANES = 2, DERIVED_H3_PLANES = 3, DERIVED_WINO_U_H3_PLANES = 4 };
void* derived_cached(const void* w, int kind, size_t bytes, int64_t version, hipStream_t st, const std::function<int(void*)>& fill);
// The same in two halves, for fills that are launched together (abr_conv_prepare_batch): derived_acquire returns the entry's buffer and, when the
// entry does not hold `version` yet, a token (the refill is already ordered behind the entry's readers); the caller fills every such buffer on `st`
// and then hands the tokens to derived_commit, which records the fill events.  Entries with a pending token are never evicted.
void* derived_acquire(const void* w, int kind, size_t bytes, int64_t version, hipStream_t st, void** token, std::vector<hipStream_t>* waited = nullptr);
void derived_commit(void* const* tokens, int n, hipStream_t st);
// the fill did NOT happen (an error between acquire and commit): the entries hold no version and are evictable / refillable again
void derived_abandon(void* const* tokens, int n);

// One job of a batched weight preparation (a table of these lives in device memory; workgroup b belongs to the job with first_block <= b).
struct PrepJob {
    const float* src;     // transpose: w [Cout][RS][Cin]; wino: w [N][3][3][C]; pack: matrix [rows][K]
    const float* scale;   // transpose: FrozenBN scale per Cout, or nullptr
    void* dst;            // transpose: wt [Cin][RS][Cout]; wino: U [36][N][C]; pack: planes
    int a, b, c;          // transpose: Cout, RS, Cin; wino: N, C, -; pack: rows, K, -
    int gx, gy;           // the job's grid (x, y) as the single-job kernel would have it (z = blocks / (gx * gy))
    int first_block;
};
// tile edge of a transpose job: 64 (16 B accesses) when both channel counts are multiples of 4 and both tensors are 16 B aligned, else 32
__host__ __device__ __forceinline__ int prep_transpose_tile(int Cout, int Cin, const void* w, const void* wt) {
    return (((Cout | Cin) & 3) == 0 && ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(wt)) & 15) == 0) ? 64 : 32;
}
int prep_transpose_multi(const PrepJob* jobs_dev, int njobs, int blocks, hipStream_t st);   // conv_igemm.hip
int prep_pack_multi(const PrepJob* jobs_dev, int njobs, int blocks, hipStream_t st);        // conv_igemm.hip
int prep_pack_h3_multi(const PrepJob* jobs_dev, int njobs, int blocks, int scale_blocks, hipStream_t st);     // conv_igemm.hip (f16x3 planes: one workgroup per 32-row block)
int prep_wino_u_multi(const PrepJob* jobs_dev, int njobs, int blocks, hipStream_t st);      // conv_winograd.hip (C % 4 == 0 jobs only)
// f16x3: w [N][3][3][C] -> packed planes + row scales of the Winograd-domain [36 N][C] matrix, U never written (N % 32 == 0, C % 64 == 0)
int prep_wino_h3_direct_multi(const PrepJob* jobs_dev, int njobs, int pack_blocks, int scale_blocks, unsigned* flags, hipStream_t st);   // conv_winograd.hip
__device__ __forceinline__ int prep_find_job(const PrepJob* jobs, int njobs, int block) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= block) lo = mid; else hi = mid - 1;
    }
    return lo;
}
void derived_cache_clear();
void derived_cache_drop_range(const void* base, size_t bytes);
size_t derived_cache_bytes();

// bf16x6 range guard (see abr_x6_range_flags in include/abr_iod_hip.h).  The flag word lives in device memory owned by common.hip.
unsigned* x6_flags_ptr();
bool x6_guard_enabled();   // ABR_X6_GUARD=0 switches the inspection off (A/B measurements of its cost only)
constexpr unsigned kX6TinyB = (17u << 24) - 1u;   // (bits << 1) - 1 of the smallest in-domain magnitude 2^-110 (biased exponent 17)
__device__ __forceinline__ void x6_report(unsigned bmin, float nonfin, unsigned* flags) {
    unsigned f = 0;
    if (bmin < kX6TinyB) f |= ABR_X6_FLAG_TINY;       // a non-zero operand below 2^-110: its low bf16 planes leave the normal range
    if (nonfin != nonfin) f |= ABR_X6_FLAG_NONFINITE;  // inf / nan operand: inf - inf poisons the low planes (NaN where fp32 gives inf)
    if (f) atomicOr(flags, f);
}

// ------------------------------------------------------------------------------------------------------------------------
// f16x3 arithmetic (ABR_MATH_F16X3, round 5): x = s (h0 + h1), h0 = fp16(x / s), h1 = fp16(x / s - h0), s = the power of two that puts the
// tensor's (weight row's) largest magnitude in [2^14, 2^15); x w = s_x s_w (h0 g0 + h0 g1 + h1 g0), three exact products per multiply-add on
// v_mfma_f32_32x32x16_f16, fp32 accumulation.  The operand's amax travels in an AMAX WORD: 64 bits = (epoch << 32) | bits of a non-negative
// float, written with ONE 64-bit atomic max per workgroup by whatever kernel produces the tensor; a later epoch always wins, so a word is
// re-used without ever being cleared, and a reader that finds another epoch than the one it was told knows the word is not its tensor's.
// Words come from a zero-initialised device ring owned by the library (h3_amax_alloc).
// ------------------------------------------------------------------------------------------------------------------------
struct AmaxRef {
    unsigned long long* word;   // nullptr = none
    unsigned epoch;
};
constexpr unsigned kH3RangeBinades = 18;   // elements below amax * 2^-18 keep an ABSOLUTE accuracy of 2^-40 amax instead of 2^-22 relative
AmaxRef h3_amax_alloc();                   // a fresh (word, epoch) of the ring; word == nullptr: no device memory
// max |x| of n floats into `ref` (one pass over x; the fallback for tensors whose producer did not emit its amax)
int h3_amax_reduce(const float* x, int64_t n, AmaxRef ref, hipStream_t st);
// amax refs of tensors the LIBRARY wrote and will read back through a caller-held buffer (a Winograd-domain input kept for the weight gradient)
void h3_amax_remember(const void* tensor, AmaxRef ref);
AmaxRef h3_amax_recall(const void* tensor);

// bits of the amax in `word` if it carries `epoch`, else 0xFFFFFFFF (stale: the caller raises ABR_H3_FLAG_STALE and treats it as 0)
__device__ __forceinline__ unsigned h3_amax_load(const unsigned long long* word, unsigned epoch) {
    const unsigned long long v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (unsigned)(v >> 32) == epoch ? (unsigned)v : 0xFFFFFFFFu;
}
// (s, 1 / s) for an amax given by its bits; a non-finite amax gives s = 1 (the caller flags it)
__device__ __forceinline__ void h3_scales(unsigned amax_bits, float& s, float& inv_s) {
    int e = (int)(amax_bits >> 23);
    if (amax_bits == 0u || e >= 255) { s = 1.f; inv_s = 1.f; return; }
    e = e < 15 ? 15 : e;                                  // amax < 2^-112: the scale stops following (operands then sit lower in fp16's range)
    s = __uint_as_float((unsigned)(e - 14) << 23);        // amax / s in [2^14, 2^15)
    inv_s = __uint_as_float((unsigned)(268 - e) << 23);
}
// block-wide max of a non-negative (or NaN) per-thread value given by its BITS, sent to the amax word by one atomic (all threads of the
// workgroup call it; blockDim.x a multiple of 64, <= 1024)
__device__ __forceinline__ void h3_amax_emit(unsigned long long* word, unsigned epoch, unsigned local_bits) {
    __shared__ unsigned sm[16];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local_bits = max(local_bits, (unsigned)__shfl_xor((int)local_bits, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = local_bits;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned m = sm[0];
        for (unsigned i = 1; i < (blockDim.x >> 6); i++) m = max(m, sm[i]);
        // Thousands of workgroups, one word: an atomic per workgroup would queue on one address (~12 ns each: a 9000-workgroup launch of 45 us
        // stretched to 150).  The word is read first and the atomic skipped when it already holds this epoch with at least this value -- a stale
        // read can only be SMALLER than the truth (the max grows monotonically inside an epoch), so skipping is always safe.
        const unsigned long long mine = ((unsigned long long)epoch << 32) | m;
        if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < mine) atomicMax(word, mine);
    }
}
// two-term fp16 split of a pair of fp32 values: four v_fma_mix (h0 = f16(x * inv_s), h1 = f16(fma(x, inv_s, -h0)): both exact before the one
// rounding to fp16); o0 / o1 = the pair's h0 / h1 as packed halves
__device__ __forceinline__ void h3_split2(const float a, const float b, const float inv_s, unsigned& o0, unsigned& o1) {
    unsigned h0, h1;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(h0) : "v"(a), "v"(inv_s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h0) : "v"(b), "v"(inv_s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(h1) : "v"(a), "v"(inv_s), "v"(h0));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(h1) : "v"(b), "v"(inv_s), "v"(h0));
    o0 = h0; o1 = h1;
}
// Range report of an f16x3 operand (one call per inspecting WAVE, all lanes): nsmall = this lane's count of non-zero elements more than 18
// binades below the amax (h3_small_threshold).  Flags as in include/abr_iod_hip.h (conditions somebody must act on); the count goes to the
// device statistics behind abr_h3_range_stats (data with such elements is ordinary: a Gaussian tensor of a million elements has a few).
constexpr int kH3StatSlots = 256;     // the small-element counts are spread over this many words (one address would serialise the atomics)
unsigned long long* h3_stats_ptr();   // device [kH3StatSlots] partial counts of small elements
void h3_stats_inspected(double n);    // host-side count of inspected operand elements (the launch functions know it)
__device__ __forceinline__ unsigned h3_small_threshold(unsigned amax_bits) {   // compare (bits << 1) - 1 of an element against this (zeros map to 0xFFFFFFFF)
    const unsigned e = amax_bits >> 23;
    return (e >= 255u || e <= kH3RangeBinades) ? 0u : ((e - kH3RangeBinades) << 24);
}
__device__ __forceinline__ void h3_report(unsigned amax_bits, unsigned nsmall, unsigned* flags, unsigned long long* stats) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nsmall += (unsigned)__shfl_xor((int)nsmall, o, 64);
    if ((threadIdx.x & 63) != 0) return;
    unsigned f = 0;
    if (amax_bits == 0xFFFFFFFFu) f |= ABR_H3_FLAG_STALE;
    else if ((amax_bits >> 23) >= 255u) f |= ABR_X6_FLAG_NONFINITE;
    if (f && (__hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & f) != f) atomicOr(flags, f);   // (the bits are sticky: set once)
    if (stats && nsmall) atomicAdd(stats + ((blockIdx.x * 4u + (threadIdx.x >> 6)) & (kH3StatSlots - 1)), (unsigned long long)nsmall);
}

// x - y as ONE v_sub_f32: the exact-split residuals must not be packed into v_pk_add_f32 (slow beside MFMAs, MI355X_MICROARCH.md)
__device__ __forceinline__ float x6_sub(float x, float y) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}

bool prof_enabled();
int prof_start(hipStream_t st, int id, double work);  // returns record index (or -1)
void prof_stop(hipStream_t st, int rec);
// Per-launch timing WITHOUT events for kernels that stamp themselves: returns a device {first start, last end} slot (wall_clock64 ticks)
// (+ two words for workgroup 0's clocks) for this launch, or nullptr when the launch is not sampled.  The kernel calls prof_stamp_begin / prof_stamp_end with it.  No event pair,
// so no dispatch bubble in front of the kernel, and the duration is the one a kernel trace reports (first wave in to last wave out).
unsigned long long* prof_stamp_slot(int id, double work);
// a slot for an EVENT-timed launch (record index of prof_start): only workgroup 0's clock words are read back
unsigned long long* prof_clock_slot(int rec);
// algorithmic HBM bytes of the launch just counted under `id` (what the kernel must move at least: every operand and the output once)
void prof_add_bytes(int id, double bytes);
// Only the first / last 512 workgroups of the (1-D) grid stamp: workgroups are dispatched in blockIdx order, so the first one in is among
// the former and the last one out among the latter -- and a 9000-workgroup grid does not send 18000 atomics to two addresses (which
// stretched short sampled launches by 7 %).
// Slot words 2 / 3: workgroup 0's own run in SHADER cycles (s_memtime) and in wall ticks (s_memrealtime, 100 MHz): their ratio is the clock the
// chip sustained under this kernel (DVFS: MI355X_MICROARCH.md "DVFS give-back").  Held in the slot between begin and end: no registers.
__device__ __forceinline__ void prof_stamp_begin(unsigned long long* ts) {
    if (ts && threadIdx.x == 0 && blockIdx.x < 512u) atomicMin(ts, (unsigned long long)wall_clock64());
    if (ts && threadIdx.x == 0 && blockIdx.x == 0u) {
        ts[2] = (unsigned long long)clock64();
        ts[3] = (unsigned long long)wall_clock64();
    }
}
__device__ __forceinline__ void prof_stamp_end(unsigned long long* ts) {
    if (ts && threadIdx.x == 0 && blockIdx.x == 0u) {
        ts[2] = (unsigned long long)clock64() - ts[2];
        ts[3] = (unsigned long long)wall_clock64() - ts[3];
    }
    if (ts && blockIdx.x + 512u >= gridDim.x) {
        __builtin_amdgcn_s_waitcnt(0);   // this wave's stores have landed: a kernel trace's duration includes them
        if (threadIdx.x == 0) atomicMax(ts + 1, (unsigned long long)wall_clock64());
    }
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline unsigned cdiv(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

}  // namespace abr
