// vt_comm.hip -- the data-parallel collectives of the path as C entry points: sum all-reduce of a gradient bucket
// (reference: DistributedDataParallel's bucketed all-reduce behind `strategy: ddp`, configs/base.yaml:17-19) and of one
// BatchNorm layer's folded statistics (SyncBatchNorm, configs/base.yaml:22), issued on the caller's HIP stream so that a
// launch list can carry them as ordinary ops (VT_OP_ALLREDUCE / VT_OP_STAT_SYNC): no host round trip, no second library
// stream, no event hop per collective.
//
// RCCL is bound at run time (dlopen + dlsym), not at link time: a process that already holds an RCCL (PyTorch's
// libtorch_hip.so brings its own librccl.so.1) must talk to THAT instance -- two RCCL images in one process would each
// run their own bootstrap and proxy threads -- and a consumer that never goes multi-GPU needs no RCCL at all.  One
// communicator per process (one process per GPU).
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and enumerators only; no symbol of it is linked

#include <mutex>

#include "vt_common.h"

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

std::mutex g_mu;
Rccl g_rccl;
ncclComm_t g_comm = nullptr;
ncclComm_t g_comm_stat = nullptr;  // optional second communicator: the SyncBatchNorm exchanges (vt_comm_init_stat)
int g_world = 0, g_rank = -1;

int bind_rccl() {
    if (g_rccl.handle) return VT_OK;
    // the soname first: if the process already mapped an RCCL under it (PyTorch), the loader hands back that image
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (h) break;
    }
    for (int k = 0; !h && k < 3; ++k) h = dlopen(names[k], RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        vt_set_error("vt_comm: RCCL not found (librccl.so.1): %s", dlerror());
        return VT_ERR_UNSUPPORTED;
    }
    Rccl r;
    r.handle = h;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.GetErrorString) {
        vt_set_error("vt_comm: the RCCL image lacks an entry point this library binds");
        return VT_ERR_UNSUPPORTED;
    }
    g_rccl = r;
    return VT_OK;
}

int rccl_error(const char* what, ncclResult_t rc) {
    vt_set_error("%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error");
    return VT_ERR_HIP;
}

}  // namespace

extern "C" {

int vt_comm_unique_id(void* id128) {
    VT_REQUIRE(id128, VT_ERR_INVALID, "vt_comm_unique_id: null id");
    std::lock_guard<std::mutex> lk(g_mu);
    const int rc = bind_rccl();
    if (rc != VT_OK) return rc;
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return rccl_error("vt_comm_unique_id", r);
    static_assert(sizeof(id) == VT_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, sizeof(id));
    return VT_OK;
}

int vt_comm_init(const void* id128, int32_t rank, int32_t world) {
    VT_REQUIRE(id128 && world >= 1 && rank >= 0 && rank < world, VT_ERR_INVALID, "vt_comm_init: bad argument");
    std::lock_guard<std::mutex> lk(g_mu);
    VT_REQUIRE(g_comm == nullptr, VT_ERR_INVALID, "vt_comm_init: this process already holds a communicator");
    const int rc = bind_rccl();
    if (rc != VT_OK) return rc;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    const ncclResult_t r = g_rccl.CommInitRank(&c, world, id, rank);  // (collective over the `world` processes; current device)
    if (r != ncclSuccess) return rccl_error("vt_comm_init", r);
    g_comm = c, g_world = world, g_rank = rank;
    return VT_OK;
}

// A second communicator over the same ranks for vt_stat_sync.  RCCL orders the operations of ONE communicator in issue
// order whatever streams they sit on: with the statistics exchanges (main stream, on the critical path of every layer)
// and the multi-megabyte bucket all-reduces (filter-gradient stream) on one communicator, each exchange would wait for
// the bucket issued before it -- and for the filter gradients queued ahead of that bucket.  Invisible with one rank.
// CO-RESIDENCY ASSUMPTION: the two communicators' kernels are issued from two streams with no cross-rank order
// between them, which RCCL / NCCL document as safe only while both collectives can be resident at once -- rank A may start
// the bucket all-reduce first and rank B the statistics exchange; each then needs its peer's kernel of the SAME communicator
// to be running.  They can be here: a collective's channel kernels take a few workgroups (<= 64 KiB of LDS), the CU-owning
// kernels of the step (span6 / wgrad6 / pspan: one 12-wave workgroup per CU) leave at least a wave slot and 56 KiB of LDS
// on every CU and, being persistent over a bounded tile list, always drain.  No multi-rank soak run exists (one-GPU
// boxes); the environment variable VT_STAT_COMM=0 (read by ensure_library_comm) keeps the single-communicator form, where
// issue order alone is the cross-rank order, at the price described above.
int vt_comm_init_stat(const void* id128) {
    VT_REQUIRE(id128, VT_ERR_INVALID, "vt_comm_init_stat: null id");
    std::lock_guard<std::mutex> lk(g_mu);
    VT_REQUIRE(g_comm != nullptr, VT_ERR_INVALID, "vt_comm_init_stat: vt_comm_init first");
    VT_REQUIRE(g_comm_stat == nullptr, VT_ERR_INVALID, "vt_comm_init_stat: already initialised");
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    const ncclResult_t r = g_rccl.CommInitRank(&c, g_world, id, g_rank);
    if (r != ncclSuccess) return rccl_error("vt_comm_init_stat", r);
    g_comm_stat = c;
    return VT_OK;
}

int vt_comm_has_stat(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    return g_comm_stat != nullptr;
}

int vt_comm_world(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    return g_comm ? g_world : 0;
}

int vt_comm_destroy(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_comm) return VT_OK;
    ncclResult_t r2 = ncclSuccess;
    if (g_comm_stat) r2 = g_rccl.CommDestroy(g_comm_stat);
    g_comm_stat = nullptr;
    const ncclResult_t r = g_rccl.CommDestroy(g_comm);
    g_comm = nullptr, g_world = 0, g_rank = -1;
    if (r != ncclSuccess) return rccl_error("vt_comm_destroy", r);
    return r2 == ncclSuccess ? VT_OK : rccl_error("vt_comm_destroy(stat)", r2);
}

static int allreduce_on(bool stat, void* buf, int64_t count, int32_t dtype, void* stream);

int vt_allreduce_bucket(void* buf, int64_t count, int32_t dtype, void* stream) {
    return allreduce_on(false, buf, count, dtype, stream);
}

static int allreduce_on(bool stat, void* buf, int64_t count, int32_t dtype, void* stream) {
    VT_REQUIRE(buf && count > 0, VT_ERR_INVALID, "vt_allreduce_bucket: bad argument");
    ncclDataType_t t;
    switch (dtype) {
        case VT_F32: t = ncclFloat32; break;
        case VT_BF16: t = ncclBfloat16; break;
        case VT_I64: t = ncclInt64; break;
        default: vt_set_error("vt_allreduce_bucket: dtype %d", dtype); return VT_ERR_UNSUPPORTED;
    }
    ncclComm_t c;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        c = (stat && g_comm_stat) ? g_comm_stat : g_comm;
    }
    VT_REQUIRE(c != nullptr, VT_ERR_INVALID, "vt_allreduce_bucket: no communicator (vt_comm_init first)");
    const ncclResult_t r = g_rccl.AllReduce(buf, buf, (size_t)count, t, ncclSum, c, (hipStream_t)stream);
    return r == ncclSuccess ? VT_OK : rccl_error("vt_allreduce_bucket", r);
}

int vt_stat_sync(float* stats, int32_t C, void* stream) {
    // one layer's BatchNorm sums over all ranks: fold the replicas into replica 0 (exact integer adds), then an int64 sum
    // all-reduce of those 4 C words (two 128-bit fixed-point sums per channel as independent 64-bit halves) -- the
    // algorithmic payload, exact and order-free, so every rank finalises identical statistics
    const int rc = vt_stat_fold(stats, C, stream);
    if (rc != VT_OK) return rc;
    // (on the statistics communicator when there is one: vt_comm_init_stat)
    return allreduce_on(true, stats, 4 * (int64_t)C, VT_I64, stream);
}

}  // extern "C"
