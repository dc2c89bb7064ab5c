// gclm_comm.hip -- thin C ABI over RCCL for the two collectives of the multi-GPU path (include/gclm.h):
// ONE all-gather of the packed result rows (independent intrinsics, BASELINE configs[2]) and ONE
// all-reduce(sum) of the per-group Schur partials per LM step (shared intrinsics split, configs[4]).
// One process per GPU; the 128-byte unique id is produced on rank 0 and handed to the other ranks by the
// caller (file, environment, torch.distributed store ...).  Payloads are KBs: latency-bound, so both
// collectives are issued once, in place, on the solve's stream.  The reference has no counterpart (its LM
// is single-process; its only collectives are DDP training calls, siclib/train.py:275-337,491,537,678).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <new>
#include <string>

#include "../../include/gclm.h"

struct gclm_comm {
    ncclComm_t comm = nullptr;
    int nranks = 0, rank = 0, device = 0;
    std::string err;
};

namespace {
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
    ncclResult_t r = ncclGetUniqueId(static_cast<ncclUniqueId*>(id_out));
    return r == ncclSuccess ? 0 : cfail(nullptr, -20, "ncclGetUniqueId", ncclGetErrorString(r));
}

int gclm_comm_versions(int* compiled, int* runtime) {
    // Which librccl answers is decided by the dynamic loader: this library names /opt/rocm/lib/librccl.so.1 (RUNPATH), but a
    // process that has already loaded another librccl with the same soname (a torch process: torch/lib/librccl.so) keeps
    // that one.  Both are reported so that the caller can see which pair is running.
    int rt = 0;
    if (ncclGetVersion(&rt) != ncclSuccess) rt = 0;
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
    int rt = 0;
    if (ncclGetVersion(&rt) != ncclSuccess || rt / 10000 != NCCL_VERSION_CODE / 10000) {
        std::string m = "librccl at run time reports version " + std::to_string(rt) + ", this library was compiled against " +
                        std::to_string(NCCL_VERSION_CODE) + " (major versions differ)";
        return cfail(nullptr, -21, "gclm_comm_create", m.c_str());
    }
    if (hipSetDevice(device) != hipSuccess) return cfail(nullptr, -10, "gclm_comm_create", "hipSetDevice failed");
    gclm_comm* c = new (std::nothrow) gclm_comm();
    if (!c) return cfail(nullptr, -12, "gclm_comm_create", "out of host memory");
    c->nranks = nranks; c->rank = rank; c->device = device;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclResult_t r = ncclCommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        cfail(nullptr, -20, "ncclCommInitRank", ncclGetErrorString(r));
        delete c;
        return -20;
    }
    *out = c;
    return 0;
}

int gclm_comm_destroy(gclm_comm* c) {
    if (!c) return 0;
    if (c->comm) ncclCommDestroy(c->comm);
    delete c;
    return 0;
}

const char* gclm_comm_last_error(const gclm_comm* c) { return c ? c->err.c_str() : g_comm_error.c_str(); }

int gclm_comm_all_gather(gclm_comm* c, const float* d_send, float* d_recv, size_t count_per_rank, void* stream) {
    if (c && count_per_rank == 0) return 0;        // every rank passes the same count: nothing to exchange anywhere
    if (!c || !d_send || !d_recv) return cfail(c, -1, "gclm_comm_all_gather", "null argument");
    ncclResult_t r = ncclAllGather(d_send, d_recv, count_per_rank, ncclFloat, c->comm, static_cast<hipStream_t>(stream));
    return r == ncclSuccess ? 0 : cfail(c, -20, "ncclAllGather", ncclGetErrorString(r));
}

int gclm_comm_all_reduce_sum(gclm_comm* c, float* d_buf, size_t count, void* stream) {
    if (!c || !d_buf) return cfail(c, -1, "gclm_comm_all_reduce_sum", "null argument");
    ncclResult_t r = ncclAllReduce(d_buf, d_buf, count, ncclFloat, ncclSum, c->comm, static_cast<hipStream_t>(stream));
    return r == ncclSuccess ? 0 : cfail(c, -20, "ncclAllReduce", ncclGetErrorString(r));
}

int gclm_comm_all_reduce_sum_i32(gclm_comm* c, int32_t* d_buf, size_t count, void* stream) {
    if (!c || !d_buf) return cfail(c, -1, "gclm_comm_all_reduce_sum_i32", "null argument");
    ncclResult_t r = ncclAllReduce(d_buf, d_buf, count, ncclInt32, ncclSum, c->comm, static_cast<hipStream_t>(stream));
    return r == ncclSuccess ? 0 : cfail(c, -20, "ncclAllReduce", ncclGetErrorString(r));
}

}  // extern "C"
