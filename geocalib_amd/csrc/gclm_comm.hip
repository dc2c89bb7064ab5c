// gclm_comm.hip -- thin C ABI over RCCL for the two collectives of the multi-GPU path (include/gclm.h):
// ONE all-gather of the packed result rows (independent intrinsics, BASELINE configs[2]) and ONE
// all-reduce(sum) of the per-group Schur partials per LM step (shared intrinsics split, configs[4]).
// One process per GPU; the 128-byte unique id is produced on rank 0 and handed to the other ranks by the
// caller (file, environment, torch.distributed store ...).  Payloads are KBs: latency-bound, so both
// collectives are issued once, in place, on the solve's stream.  The reference has no counterpart (its LM
// is single-process; its only collectives are DDP training calls, siclib/train.py:275-337,491,537,678).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <utility>

#include "../../include/gclm.h"

struct gclm_comm {
    ncclComm_t comm = nullptr;
    int nranks = 0, rank = 0, device = 0;
    std::string err;
};

namespace {
// WHICH librccl this library talks to is decided here, not by link order: the RCCL the process has already loaded (a
// torch process: torch/lib/librccl.so -- two RCCL instances in one process would each bring their own topology state and
// kernels, and torch.distributed's communicators live in that one), otherwise ROCm's (/opt/rocm/lib/librccl.so.1).  The
// library is therefore NOT linked against librccl; the seven entry points are resolved once, on first use.
struct Rccl {
    void* lib = nullptr;
    std::string origin;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    std::string why;      // !ok: the loader's message, or the entry point that is missing
};
Rccl g_rccl;
std::once_flag g_rccl_once;
std::atomic<bool> g_rccl_bound{false};
const Rccl& rccl() {
    std::call_once(g_rccl_once, [] {
        Rccl& r = g_rccl;
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names)                                    // already in the process?
            if (!r.lib && (r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) r.origin = std::string(n) + " (already loaded by the process)";
        if (!r.lib && (r.lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL))) r.origin = "/opt/rocm/lib/librccl.so.1";
        if (!r.lib) {
            dlerror();
            if ((r.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL))) r.origin = "librccl.so.1 (loader path)";
            else if (const char* e = dlerror()) r.why = e;
        }
        g_rccl_bound.store(true);
        if (!r.lib) {
            r.why = "no librccl in the process, under /opt/rocm/lib or on the loader path (" + r.why + ")";
            return;
        }
        r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(dlsym(r.lib, "ncclGetVersion"));
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.lib, "ncclAllReduce"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
        const std::pair<const char*, bool> need[] = {{"ncclGetVersion", r.GetVersion}, {"ncclGetUniqueId", r.GetUniqueId},
            {"ncclCommInitRank", r.CommInitRank}, {"ncclCommDestroy", r.CommDestroy}, {"ncclAllGather", r.AllGather},
            {"ncclAllReduce", r.AllReduce}, {"ncclGetErrorString", r.GetErrorString}};
        r.ok = true;
        for (const auto& n : need)
            if (!n.second) {
                r.ok = false;
                r.why += (r.why.empty() ? r.origin + " lacks " : std::string(", ")) + n.first;
            }
    });
    return g_rccl;
}
thread_local std::string g_comm_error;
int cfail(gclm_comm* c, int code, const char* what, const char* detail) {
    std::string m = std::string(what) + ": " + detail;
    if (c) c->err = m; else g_comm_error = m;
    return code;
}
}  // namespace

extern "C" {

int gclm_comm_unique_id(void* id_out) {
    if (!id_out) return cfail(nullptr, -1, "gclm_comm_unique_id", "null argument");
    static_assert(sizeof(ncclUniqueId) == GCLM_COMM_ID_BYTES, "unique id size");
    const Rccl& L = rccl();
    if (!L.ok) return cfail(nullptr, -22, "gclm_comm_unique_id", L.why.c_str());
    ncclResult_t r = L.GetUniqueId(static_cast<ncclUniqueId*>(id_out));
    return r == ncclSuccess ? 0 : cfail(nullptr, -20, "ncclGetUniqueId", L.GetErrorString(r));
}

int gclm_comm_versions(int* compiled, int* runtime) {
    // The rccl.h this file was compiled against, and the librccl the gclm_comm_* calls talk to.  This query must not be
    // what DECIDES the latter (a caller asking for versions before torch has loaded its own librccl would pin ROCm's for
    // the life of the process): once a gclm_comm_* call has bound one, that one is reported; before that, the librccl
    // already loaded in the process (what a binding now would pick) is asked without being bound; 0 = none loaded yet.
    int rt = 0;
    if (g_rccl_bound.load()) {
        const Rccl& L = rccl();
        if (!L.ok || L.GetVersion(&rt) != ncclSuccess) rt = 0;
    } else {
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names) {
            void* lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
            if (!lib) continue;
            auto get = reinterpret_cast<ncclResult_t (*)(int*)>(dlsym(lib, "ncclGetVersion"));
            if (!get || get(&rt) != ncclSuccess) rt = 0;
            dlclose(lib);
            break;
        }
    }
    if (compiled) *compiled = NCCL_VERSION_CODE;
    if (runtime) *runtime = rt;
    return 0;
}

int gclm_comm_create(gclm_comm** out, const void* unique_id, int nranks, int rank, int device) {
    if (!out || !unique_id || nranks <= 0 || rank < 0 || rank >= nranks)
        return cfail(nullptr, -1, "gclm_comm_create", "bad arguments");
    *out = nullptr;
    // the three entry points used here (ncclCommInitRank / ncclAllGather / ncclAllReduce) are stable within a major
    // version; a librccl of another major version must not be driven through this header's declarations
    const Rccl& L = rccl();
    if (!L.ok) return cfail(nullptr, -22, "gclm_comm_create", L.why.c_str());
    int rt = 0;
    if (L.GetVersion(&rt) != ncclSuccess || rt / 10000 != NCCL_VERSION_CODE / 10000) {
        std::string m = "librccl at run time reports version " + std::to_string(rt) + ", this library was compiled against " +
                        std::to_string(NCCL_VERSION_CODE) + " (major versions differ)";
        return cfail(nullptr, -21, "gclm_comm_create", m.c_str());
    }
    // ncclCommInitRank binds the communicator to the CURRENT device: switch to `device` for the call and hand the caller's
    // current device back afterwards, like every handle-taking entry point does (include/gclm.h)
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (hipSetDevice(device) != hipSuccess) return cfail(nullptr, -10, "gclm_comm_create", "hipSetDevice failed");
    gclm_comm* c = new (std::nothrow) gclm_comm();
    if (!c) {
        if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
        return cfail(nullptr, -12, "gclm_comm_create", "out of host memory");
    }
    c->nranks = nranks; c->rank = rank; c->device = device;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclResult_t r = L.CommInitRank(&c->comm, nranks, id, rank);
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (r != ncclSuccess) {
        cfail(nullptr, -20, "ncclCommInitRank", L.GetErrorString(r));
        delete c;
        return -20;
    }
    *out = c;
    return 0;
}

int gclm_comm_destroy(gclm_comm* c) {
    if (!c) return 0;
    if (c->comm) rccl().CommDestroy(c->comm);
    delete c;
    return 0;
}

const char* gclm_comm_last_error(const gclm_comm* c) { return c ? c->err.c_str() : g_comm_error.c_str(); }

int gclm_comm_all_gather(gclm_comm* c, const float* d_send, float* d_recv, size_t count_per_rank, void* stream) {
    if (c && count_per_rank == 0) return 0;        // every rank passes the same count: nothing to exchange anywhere
    if (!c || !d_send || !d_recv) return cfail(c, -1, "gclm_comm_all_gather", "null argument");
    ncclResult_t r = rccl().AllGather(d_send, d_recv, count_per_rank, ncclFloat, c->comm, static_cast<hipStream_t>(stream));
    return r == ncclSuccess ? 0 : cfail(c, -20, "ncclAllGather", rccl().GetErrorString(r));
}

int gclm_comm_all_reduce_sum(gclm_comm* c, float* d_buf, size_t count, void* stream) {
    if (!c || !d_buf) return cfail(c, -1, "gclm_comm_all_reduce_sum", "null argument");
    ncclResult_t r = rccl().AllReduce(d_buf, d_buf, count, ncclFloat, ncclSum, c->comm, static_cast<hipStream_t>(stream));
    return r == ncclSuccess ? 0 : cfail(c, -20, "ncclAllReduce", rccl().GetErrorString(r));
}

int gclm_comm_all_reduce_sum_i32(gclm_comm* c, int32_t* d_buf, size_t count, void* stream) {
    if (!c || !d_buf) return cfail(c, -1, "gclm_comm_all_reduce_sum_i32", "null argument");
    ncclResult_t r = rccl().AllReduce(d_buf, d_buf, count, ncclInt32, ncclSum, c->comm, static_cast<hipStream_t>(stream));
    return r == ncclSuccess ? 0 : cfail(c, -20, "ncclAllReduce", rccl().GetErrorString(r));
}

}  // extern "C"
