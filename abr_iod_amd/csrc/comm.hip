// Flat-buffer gradient all-reduce over RCCL behind the C ABI ("multi-tensor SGD, flat-buffer all-reduce").
//
// Replaces, for the reference's hot path, what DistributedDataParallel's reducer does between backward and optimizer.step()
// (tools/train_incremental.py:231-235, maskrcnn_benchmark/engine/trainer.py:15-37): there 25 MB buckets of per-parameter gradients are flattened,
// all-reduced over NCCL and copied back; here the gradients of all 52 trainable tensors ALREADY live in one flat fp32 buffer (modeling/_flat.py), so
// the exchange is an in-place sum all-reduce of a few element ranges of that buffer -- one RCCL group per bucket, on the stream the caller names, no
// staging copy.  The 1 / world factor is folded into the SGD kernel (elementwise.hip::sgd_kernel).
//
// librccl is loaded at first use with dlopen, not linked: a process that already holds a copy (PyTorch-ROCm ships one with the same soname,
// librccl.so.1) keeps using that one, and a single-GPU run that never creates a communicator never loads it.  Rendezvous is the caller's: rank 0
// draws the 128-byte unique id (abr_comm_unique_id) and hands it to the other ranks by whatever channel it has (the Python host: its
// torch.distributed store); abr_comm_init is collective over the ranks.  xGMI is point-to-point, so RCCL's rings are per-link bound: the host side
// sends few, large ranges (three buckets of 60 / 38 / 33 MB) rather than DDP's 25 MB buckets.
#include <dlfcn.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "common.h"

namespace {

constexpr int kUniqueIdBytes = 128;   // NCCL_UNIQUE_ID_BYTES (rccl.h)
struct UniqueId { char internal[kUniqueIdBytes]; };
typedef void* Comm;                   // ncclComm_t
constexpr int kFloat32 = 7, kSum = 0; // ncclFloat32, ncclSum (rccl.h)

struct Rccl {
    void* handle = nullptr;
    int (*GetVersion)(int*) = nullptr;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*CommCount)(Comm, int*) = nullptr;
    int (*CommUserRank)(Comm, int*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    const char* (*GetLastError)(Comm) = nullptr;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    static bool ok = false;
    std::call_once(once, [] {
        // a copy that is already mapped wins (same soname): one RCCL per process
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (r.handle) break;
        }
        if (!r.handle)
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (r.handle) break;
            }
        if (!r.handle) return;
        auto sym = [&](const char* n) { return dlsym(r.handle, n); };
        r.GetVersion = reinterpret_cast<int (*)(int*)>(sym("ncclGetVersion"));
        r.GetUniqueId = reinterpret_cast<int (*)(UniqueId*)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<int (*)(Comm*, int, UniqueId, int)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<int (*)(Comm)>(sym("ncclCommDestroy"));
        r.CommCount = reinterpret_cast<int (*)(Comm, int*)>(sym("ncclCommCount"));
        r.CommUserRank = reinterpret_cast<int (*)(Comm, int*)>(sym("ncclCommUserRank"));
        r.AllReduce = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, Comm, hipStream_t)>(sym("ncclAllReduce"));
        r.GroupStart = reinterpret_cast<int (*)()>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<int (*)()>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<const char* (*)(int)>(sym("ncclGetErrorString"));
        r.GetLastError = reinterpret_cast<const char* (*)(Comm)>(sym("ncclGetLastError"));
        ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.GroupStart && r.GroupEnd;
    });
    return ok ? &r : nullptr;
}

const char* err_str(Rccl* r, int rc) { return r->GetErrorString ? r->GetErrorString(rc) : "rccl error"; }

struct AbrComm {
    Comm comm = nullptr;
    int world = 0, rank = 0, device = 0;
};

}  // namespace

#define ABR_RCCL(call, what)                                                              \
    do {                                                                                  \
        const int rc__ = (call);                                                          \
        if (rc__ != 0) {                                                                  \
            ::abr::set_error("%s: RCCL error %d (%s)", what, rc__, err_str(r, rc__));     \
            return ABR_E_LAUNCH;                                                          \
        }                                                                                 \
    } while (0)

extern "C" int abr_comm_rccl_version(void) {
    Rccl* r = rccl();
    int v = 0;
    if (!r || !r->GetVersion || r->GetVersion(&v) != 0) return 0;
    return v;
}

extern "C" int abr_comm_unique_id(void* id128_host) {
    ABR_REQUIRE(id128_host, "comm_unique_id: null pointer");
    Rccl* r = rccl();
    ABR_REQUIRE(r, "comm_unique_id: librccl.so.1 could not be loaded (%s)", dlerror() ? dlerror() : "no loader message");
    UniqueId id;
    ABR_RCCL(r->GetUniqueId(&id), "comm_unique_id");
    memcpy(id128_host, id.internal, kUniqueIdBytes);
    return ABR_OK;
}

extern "C" int abr_comm_init(int world, int rank, const void* id128_host, void** comm_out) {
    ABR_REQUIRE(comm_out && id128_host && world >= 1 && rank >= 0 && rank < world, "comm_init: bad arguments (world %d, rank %d)", world, rank);
    Rccl* r = rccl();
    ABR_REQUIRE(r, "comm_init: librccl.so.1 could not be loaded");
    UniqueId id;
    memcpy(id.internal, id128_host, kUniqueIdBytes);
    AbrComm* c = new AbrComm();
    c->world = world; c->rank = rank;
    (void)hipGetDevice(&c->device);
    const int rc = r->CommInitRank(&c->comm, world, id, rank);   // collective: returns once every rank has joined
    if (rc != 0) {
        ::abr::set_error("comm_init: ncclCommInitRank failed on rank %d of %d: %d (%s)", rank, world, rc, err_str(r, rc));
        delete c;
        return ABR_E_LAUNCH;
    }
    *comm_out = c;
    return ABR_OK;
}

extern "C" int abr_comm_info(void* comm, int32_t* out_host) {
    ABR_REQUIRE(comm && out_host, "comm_info: null pointer");
    AbrComm* c = static_cast<AbrComm*>(comm);
    Rccl* r = rccl();
    ABR_REQUIRE(r, "comm_info: RCCL is not loaded");
    int n = c->world, me = c->rank;
    if (r->CommCount) ABR_RCCL(r->CommCount(c->comm, &n), "comm_info");
    if (r->CommUserRank) ABR_RCCL(r->CommUserRank(c->comm, &me), "comm_info");
    out_host[0] = n; out_host[1] = me; out_host[2] = c->device;
    return ABR_OK;
}

extern "C" int abr_comm_destroy(void* comm) {
    if (!comm) return ABR_OK;
    AbrComm* c = static_cast<AbrComm*>(comm);
    Rccl* r = rccl();
    int rc = 0;
    if (r && c->comm) rc = r->CommDestroy(c->comm);
    delete c;
    ABR_REQUIRE(rc == 0, "comm_destroy: ncclCommDestroy failed: %d", rc);
    return ABR_OK;
}

extern "C" int abr_allreduce_flat(void* comm, float* flat, const int64_t* ranges_host, int n_ranges, void* stream) {
    ABR_REQUIRE(comm && n_ranges >= 0, "allreduce_flat: bad arguments");
    if (n_ranges == 0) return ABR_OK;
    ABR_REQUIRE(flat && ranges_host, "allreduce_flat: null pointer");
    AbrComm* c = static_cast<AbrComm*>(comm);
    Rccl* r = rccl();
    ABR_REQUIRE(r, "allreduce_flat: RCCL is not loaded");
    for (int i = 0; i < n_ranges; i++)
        ABR_REQUIRE(ranges_host[2 * i] >= 0 && ranges_host[2 * i + 1] >= ranges_host[2 * i], "allreduce_flat: range %d is [%lld, %lld)", i,
                    (long long)ranges_host[2 * i], (long long)ranges_host[2 * i + 1]);
    hipStream_t st = abr::as_stream(stream);
    // one group: RCCL schedules the ranges back to back on `stream` (in place: send == recv)
    ABR_RCCL(r->GroupStart(), "allreduce_flat (group start)");
    for (int i = 0; i < n_ranges; i++) {
        const int64_t a = ranges_host[2 * i], b = ranges_host[2 * i + 1];
        if (b == a) continue;
        const int rc = r->AllReduce(flat + a, flat + a, (size_t)(b - a), kFloat32, kSum, c->comm, st);
        if (rc != 0) {
            (void)r->GroupEnd();
            ::abr::set_error("allreduce_flat: ncclAllReduce of range %d failed: %d (%s)", i, rc, err_str(r, rc));
            return ABR_E_LAUNCH;
        }
    }
    ABR_RCCL(r->GroupEnd(), "allreduce_flat (group end)");
    return ABR_OK;
}
