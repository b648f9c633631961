#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <vector>

#include "common.h"

namespace abr {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

struct ProfRec { hipEvent_t a, b; double work; int id; bool overlapped; int slot; int clk = -1; };   // clk >= 0: event-timed launch with a clocks-only slot   // slot >= 0: self-stamped launch (no events)
constexpr int kMaxStampSlots = 1 << 16;
static unsigned long long* g_ts = nullptr;   // device [kMaxStampSlots][4] = {first start, last end} in wall_clock64 ticks, workgroup 0's {shader cycles, wall ticks}
static double g_clk[32][2] = {{0}};          // per kernel id: sums of the latter two over the sampled launches (filled by abr_prof_end)
static int g_ts_used = 0;
static bool g_prof = false;
static bool g_overlap = false;
static unsigned g_mask = 0xFFFFFFFFu;
static unsigned g_every = 1;
static unsigned g_seen[32] = {0};          // per kernel id: launches since the last abr_prof_step_begin
static unsigned g_step = 0;                // steps begun since abr_prof_begin
static double g_all_launches[32] = {0}, g_all_work[32] = {0};   // every launch between begin and end, bracketed or not
static double g_all_bytes[32] = {0};   // ALGORITHMIC HBM bytes of those launches (operands + output + fused residual / mask, each once)
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;
bool prof_enabled() { return g_prof; }
static hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
int prof_start(hipStream_t st, int id, double work) {
    if (!g_prof) return -1;
    if (id >= 0 && id < 32) { g_all_launches[id] += 1.0; g_all_work[id] += work; }
    if (!((g_mask >> id) & 1u)) return -1;
    if (g_every > 1) {
        // systematic sample: launch i of this kernel in step s is bracketed iff (i + s) % n == 0 -- over n consecutive steps every
        // launch position (= every shape of the step's fixed launch sequence) is sampled exactly once, so sampled averages equal
        // population averages when the number of steps is a multiple of n
        if ((g_seen[id & 31]++ + g_step) % g_every != 0) return -1;
    }
    ProfRec r{get_event(), get_event(), work, id, g_overlap, -1};
    (void)hipEventRecord(r.a, st);
    g_recs.push_back(r);
    return (int)g_recs.size() - 1;
}
unsigned long long* prof_stamp_slot(int id, double work) {
    if (!g_prof) return nullptr;
    if (id >= 0 && id < 32) { g_all_launches[id] += 1.0; g_all_work[id] += work; }
    if (!((g_mask >> id) & 1u) || !g_ts) return nullptr;
    if (g_every > 1 && (g_seen[id & 31]++ + g_step) % g_every != 0) return nullptr;   // the same systematic sample as prof_start
    if (g_ts_used >= kMaxStampSlots) return nullptr;
    ProfRec r{nullptr, nullptr, work, id, g_overlap, g_ts_used++};
    g_recs.push_back(r);
    return g_ts + 4 * (size_t)r.slot;
}
unsigned long long* prof_clock_slot(int rec) {
    if (rec < 0 || !g_ts || g_ts_used >= kMaxStampSlots) return nullptr;
    g_recs[rec].clk = g_ts_used++;
    return g_ts + 4 * (size_t)g_recs[rec].clk;
}
void prof_add_bytes(int id, double bytes) {
    if (g_prof && id >= 0 && id < 32) g_all_bytes[id] += bytes;
}
void prof_stop(hipStream_t st, int rec) {
    if (rec >= 0) (void)hipEventRecord(g_recs[rec].b, st);
}
}  // namespace abr

extern "C" int abr_prof_step_begin(void) {
    abr::g_step++;
    for (int i = 0; i < 32; i++) abr::g_seen[i] = 0;
    return ABR_OK;
}
extern "C" int abr_prof_begin(void) {
    if (!abr::g_ts && hipMalloc(&abr::g_ts, sizeof(unsigned long long) * 4 * abr::kMaxStampSlots) != hipSuccess) abr::g_ts = nullptr;
    if (abr::g_ts) {   // start = all ones (atomicMin), end = 0 (atomicMax)
        (void)hipDeviceSynchronize();
        (void)hipMemset2D(abr::g_ts, 32, 0xFF, 8, abr::kMaxStampSlots);
        (void)hipMemset2D(abr::g_ts + 1, 32, 0x00, 24, abr::kMaxStampSlots);
        (void)hipDeviceSynchronize();
    }
    abr::g_ts_used = 0;
    abr::g_prof = true;
    abr::g_step = 0;
    for (int i = 0; i < 32; i++) abr::g_all_launches[i] = abr::g_all_work[i] = abr::g_all_bytes[i] = abr::g_clk[i][0] = abr::g_clk[i][1] = 0.0;
    return ABR_OK;
}
// out[id*2 + {0,1}] = {launches, total work} of EVERY launch since abr_prof_begin (sampled or not): the executed flops of a step
extern "C" int abr_prof_totals(double* out, int n_ids) {
    for (int i = 0; i < n_ids && i < 32; i++) { out[2 * i] = abr::g_all_launches[i]; out[2 * i + 1] = abr::g_all_work[i]; }
    return ABR_OK;
}
// out[id] = algorithmic HBM bytes of EVERY launch of kernel id since abr_prof_begin (see abr_prof_totals for the launch counts)
extern "C" int abr_prof_bytes(double* out, int n_ids) {
    for (int i = 0; i < n_ids && i < 32; i++) out[i] = abr::g_all_bytes[i];
    return ABR_OK;
}
// out[id*2 + {0,1}] = {shader cycles, milliseconds} that workgroup 0 of the SAMPLED self-stamping launches of kernel id ran, summed by the
// last abr_prof_end: cycles / ms / 1e6 = the clock (GHz) the chip sustained under that kernel
extern "C" int abr_prof_clocks(double* out, int n_ids) {
    for (int i = 0; i < n_ids && i < 32; i++) { out[2 * i] = abr::g_clk[i][0]; out[2 * i + 1] = abr::g_clk[i][1]; }
    return ABR_OK;
}
extern "C" int abr_prof_set_mask(uint32_t mask, int every_nth) {
    abr::g_mask = mask;
    abr::g_every = every_nth > 1 ? (unsigned)every_nth : 1u;
    for (int i = 0; i < 32; i++) abr::g_seen[i] = 0;
    return ABR_OK;
}
extern "C" int abr_prof_mark_overlap(int on) {
    abr::g_overlap = on != 0;
    return ABR_OK;
}
// out[id*6 + {0,1,2}] = {launch count, total milliseconds, total work (flops or bytes)} of exclusive launches, +3: overlapped ones
extern "C" int abr_prof_end(double* out, int n_ids) {
    abr::g_prof = false;
    for (int i = 0; i < n_ids * 6; i++) out[i] = 0.0;
    std::vector<unsigned long long> ts;
    double ticks_per_ms = 1e5;   // wall_clock64: 100 MHz unless the device says otherwise
    if (abr::g_ts_used > 0) {
        (void)hipDeviceSynchronize();
        ts.resize(4 * (size_t)abr::g_ts_used);
        if (hipMemcpy(ts.data(), abr::g_ts, ts.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) ts.clear();
        int dev = 0, khz = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) == hipSuccess && khz > 0)
            ticks_per_ms = (double)khz;
    }
    for (auto& r : abr::g_recs) {
        if (r.slot >= 0) {   // self-stamped launch
            if ((size_t)(4 * r.slot + 3) < ts.size() && r.id < n_ids) {
                const unsigned long long t0 = ts[4 * r.slot], t1 = ts[4 * r.slot + 1], cyc = ts[4 * r.slot + 2], wall = ts[4 * r.slot + 3];
                if (r.id < 32 && wall > 100ull && wall < (1ull << 40) && cyc < (1ull << 48)) {   // workgroup 0 ran >= 1 us: a usable clock ratio
                    abr::g_clk[r.id][0] += (double)cyc;
                    abr::g_clk[r.id][1] += (double)wall / ticks_per_ms;
                }
                if (t1 > t0 && t0 != ~0ull) {
                    double* o = out + r.id * 6 + (r.overlapped ? 3 : 0);
                    o[0] += 1.0;
                    o[1] += (double)(t1 - t0) / ticks_per_ms;
                    o[2] += r.work;
                }
            }
            continue;
        }
        if (r.clk >= 0 && (size_t)(4 * r.clk + 3) < ts.size() && r.id < 32) {   // event-timed launch that carried a clocks-only slot
            const unsigned long long cyc = ts[4 * r.clk + 2], wall = ts[4 * r.clk + 3];
            if (wall > 100ull && wall < (1ull << 40) && cyc < (1ull << 48)) {
                abr::g_clk[r.id][0] += (double)cyc;
                abr::g_clk[r.id][1] += (double)wall / ticks_per_ms;
            }
        }
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess && r.id < n_ids) {
            double* o = out + r.id * 6 + (r.overlapped ? 3 : 0);
            o[0] += 1.0;
            o[1] += ms;
            o[2] += r.work;
        }
        abr::g_pool.push_back(r.a);
        abr::g_pool.push_back(r.b);
    }
    abr::g_recs.clear();
    return ABR_OK;
}

namespace {
__global__ void prof_empty_kernel() {}
}
// What an event pair costs around a kernel on a busy stream: n back-to-back [record, empty kernel, record] brackets; the median elapsed
// time is the part of an event-bracketed duration that is not the kernel (the dispatch gap the start event exposes), less the ~1 us an
// empty kernel runs.  bench.py subtracts it so that its live per-launch durations agree with rocprofv3's kernel durations.
extern "C" int abr_prof_event_overhead_ms(double* out_host, void* stream) {
    ABR_REQUIRE(out_host, "prof_event_overhead_ms: null pointer");
    hipStream_t st = abr::as_stream(stream);
    const int n = 33;
    std::vector<hipEvent_t> ev(2 * n);
    for (auto& e : ev) ABR_REQUIRE(hipEventCreate(&e) == hipSuccess, "prof_event_overhead_ms: no event");
    for (int i = 0; i < n; i++) {
        (void)hipEventRecord(ev[2 * i], st);
        prof_empty_kernel<<<1, 64, 0, st>>>();
        (void)hipEventRecord(ev[2 * i + 1], st);
    }
    std::vector<float> ms(n, 0.f);
    bool ok = hipStreamSynchronize(st) == hipSuccess;
    for (int i = 0; i < n && ok; i++) ok = hipEventElapsedTime(&ms[i], ev[2 * i], ev[2 * i + 1]) == hipSuccess;
    for (auto& e : ev) (void)hipEventDestroy(e);
    ABR_REQUIRE(ok, "prof_event_overhead_ms: event timing failed");
    std::sort(ms.begin(), ms.end());
    *out_host = ms[n / 2];
    return ABR_OK;
}

namespace abr {
bool x6_guard_enabled() {
    static const bool on = !(getenv("ABR_X6_GUARD") && atoi(getenv("ABR_X6_GUARD")) == 0);
    return on;
}
unsigned* x6_flags_ptr() {
    static unsigned* p = nullptr;
    if (!p) {
        if (hipMalloc(&p, sizeof(unsigned)) != hipSuccess) return nullptr;
        (void)hipMemset(p, 0, sizeof(unsigned));
    }
    return p;
}
}  // namespace abr
extern "C" int abr_x6_range_flags(uint32_t* out_host, int reset, void* stream) {
    ABR_REQUIRE(out_host, "x6_range_flags: null pointer");
    unsigned* p = abr::x6_flags_ptr();
    ABR_REQUIRE(p, "x6_range_flags: no device memory");
    hipStream_t st = abr::as_stream(stream);
    if (hipMemcpyAsync(out_host, p, sizeof(unsigned), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        abr::set_error("x6_range_flags: read-back failed");
        return ABR_E_LAUNCH;
    }
    if (reset && *out_host) (void)hipMemsetAsync(p, 0, sizeof(unsigned), st);
    return ABR_OK;
}

extern "C" int abr_x6_range_flags_async(uint32_t* out_pinned_host, void* stream) {
    ABR_REQUIRE(out_pinned_host, "x6_range_flags_async: null pointer");
    unsigned* p = abr::x6_flags_ptr();
    ABR_REQUIRE(p, "x6_range_flags_async: no device memory");
    if (hipMemcpyAsync(out_pinned_host, p, sizeof(unsigned), hipMemcpyDeviceToHost, abr::as_stream(stream)) != hipSuccess) {
        abr::set_error("x6_range_flags_async: copy failed");
        return ABR_E_LAUNCH;
    }
    return ABR_OK;
}

extern "C" int abr_x6_range_flags_to_device(uint32_t* out_device, void* stream) {
    ABR_REQUIRE(out_device, "x6_range_flags_to_device: null pointer");
    unsigned* p = abr::x6_flags_ptr();
    ABR_REQUIRE(p, "x6_range_flags_to_device: no device memory");
    if (hipMemcpyAsync(out_device, p, sizeof(unsigned), hipMemcpyDeviceToDevice, abr::as_stream(stream)) != hipSuccess) {
        abr::set_error("x6_range_flags_to_device: copy failed");
        return ABR_E_LAUNCH;
    }
    return ABR_OK;
}

// ---------------------------------------------------------------------------------------------------- deterministic sums (common.h)
namespace abr {
DetWs det_ws(hipStream_t st, size_t nfloats) {
    constexpr size_t kFloats = 1u << 16, kTickets = 4096;
    struct Ring { float* part = nullptr; unsigned* tick = nullptr; size_t head = 0, thead = 0; };
    static std::map<hipStream_t, Ring> rings;
    static std::mutex mu;
    std::lock_guard<std::mutex> g(mu);
    if (nfloats == 0 || nfloats > kFloats) return DetWs{nullptr, nullptr};
    Ring& r = rings[st];
    if (!r.part) {
        if (hipMalloc(&r.part, kFloats * sizeof(float)) != hipSuccess) { r.part = nullptr; return DetWs{nullptr, nullptr}; }
        if (hipMalloc(&r.tick, kTickets * sizeof(unsigned)) != hipSuccess) { (void)hipFree(r.part); r.part = nullptr; return DetWs{nullptr, nullptr}; }
        (void)hipMemset(r.tick, 0, kTickets * sizeof(unsigned));
    }
    if (r.head + nfloats > kFloats) r.head = 0;
    DetWs w{r.part + r.head, r.tick + (r.thead++ % kTickets)};
    r.head += nfloats;
    return w;
}
}  // namespace abr

// ---------------------------------------------------------------------------------------------------- f16x3: amax words (common.h)
namespace {
__global__ __launch_bounds__(256) void h3_amax_kernel(const float* __restrict__ x, int64_t n, unsigned long long* word, unsigned epoch) {
    unsigned m = 0;
    const int64_t n4 = n >> 2;
    const uint4* x4 = reinterpret_cast<const uint4*>(x);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const uint4 v = x4[i];
        m = max(max(m, max(v.x & 0x7FFFFFFFu, v.y & 0x7FFFFFFFu)), max(v.z & 0x7FFFFFFFu, v.w & 0x7FFFFFFFu));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = max(m, __float_as_uint(x[(n4 << 2) + threadIdx.x]) & 0x7FFFFFFFu);
    abr::h3_amax_emit(word, epoch, m);
}
std::mutex g_amax_mu;
unsigned long long* g_amax_ring = nullptr;
uint64_t g_amax_count = 0;
std::map<const void*, abr::AmaxRef> g_amax_map;
}  // namespace
namespace abr {
static std::mutex g_h3_stats_mu;
static double g_h3_inspected = 0.0;
unsigned long long* h3_stats_ptr() {
    static unsigned long long* p = nullptr;
    std::lock_guard<std::mutex> g(g_h3_stats_mu);
    if (!p) {
        if (hipMalloc(&p, kH3StatSlots * sizeof(unsigned long long)) != hipSuccess) return nullptr;
        (void)hipMemset(p, 0, kH3StatSlots * sizeof(unsigned long long));
    }
    return p;
}
void h3_stats_inspected(double n) {
    std::lock_guard<std::mutex> g(g_h3_stats_mu);
    g_h3_inspected += n;
}
AmaxRef h3_amax_alloc() {
    std::lock_guard<std::mutex> g(g_amax_mu);
    if (!g_amax_ring) {
        if (hipMalloc(&g_amax_ring, (size_t)ABR_H3_AMAX_RING * 8) != hipSuccess) { g_amax_ring = nullptr; return AmaxRef{nullptr, 0}; }
        (void)hipMemset(g_amax_ring, 0, (size_t)ABR_H3_AMAX_RING * 8);
    }
    const uint64_t c = g_amax_count++;
    // epochs grow with every allocation, so a word's later owner always outranks its earlier ones in the 64-bit max (0 is never used: a cleared
    // word carries no epoch); 2^32 allocations ~ 10^7 training steps
    return AmaxRef{g_amax_ring + c % ABR_H3_AMAX_RING, (unsigned)(c + 1)};
}
// n consecutive allocations: the block's first is an ordinary allocation (which also makes the ring); the rest follow it unless another thread
// allocated in between (then the block restarts at the current count and that first allocation is simply not used)
bool h3_amax_alloc_block(int n, unsigned long long** base, uint64_t* first_count) {
    const AmaxRef r = h3_amax_alloc();
    if (!r.word) return false;
    std::lock_guard<std::mutex> g(g_amax_mu);
    uint64_t first = (uint64_t)r.epoch - 1;
    if (g_amax_count != first + 1) first = g_amax_count;
    g_amax_count = first + (uint64_t)n;
    *base = g_amax_ring;
    *first_count = first;
    return true;
}
int h3_amax_reduce(const float* x, int64_t n, AmaxRef ref, hipStream_t st) {
    if (!ref.word) return 1;
    const int64_t blocks = std::min<int64_t>(std::max<int64_t>((n / 4 + 255) / 256, 1), 2048);
    h3_amax_kernel<<<(unsigned)blocks, 256, 0, st>>>(x, n, ref.word, ref.epoch);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
void h3_amax_remember(const void* tensor, AmaxRef ref) {
    std::lock_guard<std::mutex> g(g_amax_mu);
    if (g_amax_map.size() > 4096) g_amax_map.clear();   // (refs older than the ring are dead anyway)
    g_amax_map[tensor] = ref;
}
AmaxRef h3_amax_recall(const void* tensor) {
    std::lock_guard<std::mutex> g(g_amax_mu);
    auto it = g_amax_map.find(tensor);
    if (it == g_amax_map.end() || g_amax_count - (uint64_t)it->second.epoch >= (uint64_t)ABR_H3_AMAX_RING) return AmaxRef{nullptr, 0};
    return it->second;
}
}  // namespace abr
extern "C" int abr_h3_amax_alloc(uint64_t** word_out, uint32_t* epoch_out) {
    ABR_REQUIRE(word_out && epoch_out, "h3_amax_alloc: null pointer");
    const abr::AmaxRef r = abr::h3_amax_alloc();
    ABR_REQUIRE(r.word, "h3_amax_alloc: no device memory");
    *word_out = reinterpret_cast<uint64_t*>(r.word);
    *epoch_out = r.epoch;
    return ABR_OK;
}
extern "C" int abr_h3_amax_alloc_block(int n, uint64_t** ring_base_out, uint64_t* first_count_out) {
    ABR_REQUIRE(ring_base_out && first_count_out && n > 0 && n <= ABR_H3_AMAX_RING / 16, "h3_amax_alloc_block: bad args");
    unsigned long long* base = nullptr;
    ABR_REQUIRE(abr::h3_amax_alloc_block(n, &base, first_count_out), "h3_amax_alloc_block: no device memory");
    *ring_base_out = reinterpret_cast<uint64_t*>(base);
    return ABR_OK;
}
// one event per signalling stream, re-recorded by every call: hipStreamWaitEvent waits for the record that precedes it, later records do not move it
extern "C" int abr_stream_wait_stream(void* waiter, void* signaller) {
    static std::mutex mu;
    static std::map<hipStream_t, hipEvent_t> events;
    hipStream_t w = abr::as_stream(waiter), s = abr::as_stream(signaller);
    if (w == s) return ABR_OK;
    std::lock_guard<std::mutex> g(mu);
    hipEvent_t& ev = events[s];
    if (!ev) ABR_REQUIRE(hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess, "stream_wait_stream: no event");
    ABR_REQUIRE(hipEventRecord(ev, s) == hipSuccess && hipStreamWaitEvent(w, ev, 0) == hipSuccess, "stream_wait_stream: record / wait failed");
    return ABR_OK;
}
extern "C" int abr_h3_range_stats(uint64_t* out_host, int reset, void* stream) {
    ABR_REQUIRE(out_host, "h3_range_stats: null pointer");
    unsigned long long* p = abr::h3_stats_ptr();
    ABR_REQUIRE(p, "h3_range_stats: no device memory");
    hipStream_t st = abr::as_stream(stream);
    unsigned long long slots[abr::kH3StatSlots];
    if (hipMemcpyAsync(slots, p, sizeof slots, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        abr::set_error("h3_range_stats: read-back failed");
        return ABR_E_LAUNCH;
    }
    unsigned long long small = 0;
    for (int i = 0; i < abr::kH3StatSlots; i++) small += slots[i];
    out_host[0] = small;
    {
        std::lock_guard<std::mutex> g(abr::g_h3_stats_mu);
        out_host[1] = (uint64_t)abr::g_h3_inspected;
        if (reset) abr::g_h3_inspected = 0.0;
    }
    if (reset) (void)hipMemsetAsync(p, 0, sizeof slots, st);
    return ABR_OK;
}
namespace {
__global__ void h3_stats_sum_kernel(unsigned long long* slots, unsigned long long inspected, int reset, unsigned long long* out) {
    unsigned long long v = slots[threadIdx.x];
    if (reset) slots[threadIdx.x] = 0;
    __shared__ unsigned long long sm[abr::kH3StatSlots];
    sm[threadIdx.x] = v;
    __syncthreads();
    for (int o = abr::kH3StatSlots / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = sm[0]; out[1] = inspected; }
}
}  // namespace
extern "C" int abr_h3_range_stats_to_device(uint64_t* out_device, int reset, void* stream) {
    ABR_REQUIRE(out_device, "h3_range_stats_to_device: null pointer");
    unsigned long long* p = abr::h3_stats_ptr();
    ABR_REQUIRE(p, "h3_range_stats_to_device: no device memory");
    unsigned long long inspected;
    {
        std::lock_guard<std::mutex> g(abr::g_h3_stats_mu);
        inspected = (unsigned long long)abr::g_h3_inspected;
        if (reset) abr::g_h3_inspected = 0.0;
    }
    h3_stats_sum_kernel<<<1, abr::kH3StatSlots, 0, abr::as_stream(stream)>>>(p, inspected, reset, reinterpret_cast<unsigned long long*>(out_device));
    ABR_CHECK_LAUNCH("h3_range_stats_to_device");
    return ABR_OK;
}
extern "C" int abr_h3_amax(const float* x, int64_t n, uint64_t* word, uint32_t epoch, void* stream) {
    ABR_REQUIRE(word && n >= 0 && (n == 0 || x), "h3_amax: bad args");
    ABR_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, "h3_amax: x must be 16-byte aligned");
    ABR_REQUIRE(abr::h3_amax_reduce(x, n, abr::AmaxRef{reinterpret_cast<unsigned long long*>(word), epoch}, abr::as_stream(stream)) == 0, "h3_amax: launch failed");
    return ABR_OK;
}

extern "C" const char* abr_last_error(void) { return abr::g_err; }
extern "C" int abr_version(void) { return 100; }
extern "C" int abr_device_info(int32_t* out) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) {
        abr::set_error("abr_device_info: no HIP device");
        return ABR_E_LAUNCH;
    }
    out[0] = p.multiProcessorCount;
    out[1] = (int32_t)p.maxSharedMemoryPerMultiProcessor;
    out[2] = p.warpSize;
    return ABR_OK;
}
