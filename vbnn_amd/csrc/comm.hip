// comm.hip -- the data-parallel exchange of the hot path behind the C ABI: a SUM all-reduce of the (d/dmeans, d/dlvars,
// d/dbias) gradient buckets over RCCL / xGMI, one communicator per process (= per GPU), issued on a stream of its own
// so that it overlaps the rest of backward (include/vbnn_hip.h, "data-parallel exchange").
//
// The reference has no multi-GPU path at all (its only parallelism is BLAS threads, main.lua:142); what this replaces
// is nothing in the reference -- it is the exchange step BASELINE.json's north_star adds after accGradParameters
// (VBLinear.lua:112-118), reachable from a LuaJIT host exactly as from Python.
//
// librccl is bound at run time (dlopen), not at link time: single-GPU users never load its half gigabyte, and inside
// a PyTorch process the loader hands back the copy PyTorch already mapped (same SONAME librccl.so.1), so the process
// holds ONE RCCL. Only types come from <rccl/rccl.h>.
#include "common.h"
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

// 0 = bound. Tries, in order: VBNN_RCCL_PATH, whatever the process already holds under the SONAME, the ROCm install.
int bind_rccl() {
    if (g_rccl.handle) return VBNN_OK;
    const char* names[] = {getenv("VBNN_RCCL_PATH"), "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    void* h = nullptr;
    std::string tried;
    for (const char* n : names) {
        if (!n || !*n) continue;
        h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
        tried += std::string(n) + ": " + dlerror() + "; ";
    }
    if (!h) { vbnn_set_error("librccl could not be loaded (%s)", tried.c_str()); return VBNN_ERR_UNSUPPORTED; }
    Rccl r;
    r.handle = h;
#define VBNN_SYM(field, name)                                                                       \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name));                                  \
    if (!r.field) { vbnn_set_error("librccl has no symbol %s", name); dlclose(h); return VBNN_ERR_UNSUPPORTED; }
    VBNN_SYM(GetUniqueId, "ncclGetUniqueId")
    VBNN_SYM(CommInitRank, "ncclCommInitRank")
    VBNN_SYM(CommDestroy, "ncclCommDestroy")
    VBNN_SYM(CommCount, "ncclCommCount")
    VBNN_SYM(AllReduce, "ncclAllReduce")
    VBNN_SYM(AllGather, "ncclAllGather")
    VBNN_SYM(ReduceScatter, "ncclReduceScatter")
    VBNN_SYM(GetErrorString, "ncclGetErrorString")
#undef VBNN_SYM
    g_rccl = r;
    return VBNN_OK;
}
}  // namespace

#define VBNN_CHECK_NCCL(expr)                                                                        \
    do {                                                                                             \
        ncclResult_t _r = (expr);                                                                    \
        if (_r != ncclSuccess) {                                                                     \
            vbnn_set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__); \
            return VBNN_ERR_HIP;                                                                     \
        }                                                                                            \
    } while (0)

struct vbnn_comm {
    vbnn_ctx* ctx;
    ncclComm_t comm;
    int rank, world;
    hipStream_t stream;        // the exchange stream: all-reduces run here, beside the compute stream
    hipEvent_t ready;          // compute stream -> exchange stream: the bucket is complete (the event form of the hand-off)
    hipEvent_t done;           // exchange stream -> compute stream: every all-reduce issued so far has finished
    int64_t pending;           // all-reduces issued since the last vbnn_comm_finish
    unsigned* trig;            // uncached device word: the TRIGGER form of the hand-off (comm_handoff), or NULL: events
    unsigned trig_count, done_count;
};

// ---- compute stream -> exchange stream: "the bucket is complete". As in p2p.hip (r05): an event record is a marker packet on the
// compute stream -- ~8 us of bubble between the two launches it separates, per bucket -- and the exchange stream needs ~12 us to wake
// up behind it (kernel trace of the one-GPU stand-in). The trigger form: a one-thread kernel on the compute stream bumps an uncached
// word; a one-wave kernel on the exchange stream, in front of the collective, polls it. VBNN_COMM_FLAG_TRIGGER=0 at create: events.
__global__ __launch_bounds__(64) void k_comm_signal(unsigned* word, unsigned value) {
    if (threadIdx.x == 0) __hip_atomic_store(word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ __launch_bounds__(64) void k_comm_wait(const unsigned* word, unsigned value) {
    // (unbounded on purpose: the trigger sits on the compute stream behind launches already enqueued -- it comes unless the device
    // is lost; the sleep keeps the wave off the SIMD's issue slots)
    while ((int)(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - value) < 0) __builtin_amdgcn_s_sleep(8);
}
// signal on one stream, poll on the other -- the poll goes out only behind a signal that was accepted: a wait without its signal
// would hold its stream for ever
static int comm_signal_wait(hipStream_t from, hipStream_t to, unsigned* word, unsigned value, const char* what) {
    hipLaunchKernelGGL(k_comm_signal, dim3(1), dim3(64), 0, from, word, value);
    const int st = vbnn_check_launch(what);
    if (st != VBNN_OK) return st;
    hipLaunchKernelGGL(k_comm_wait, dim3(1), dim3(64), 0, to, word, value);
    return vbnn_check_launch(what);
}
static int comm_handoff(vbnn_comm* c) {
    if (!c->trig) {
        VBNN_CHECK_HIP(hipEventRecord(c->ready, c->ctx->stream));
        VBNN_CHECK_HIP(hipStreamWaitEvent(c->stream, c->ready, 0));
        return VBNN_OK;
    }
    c->trig_count += 1;
    return comm_signal_wait(c->ctx->stream, c->stream, c->trig, c->trig_count, "comm hand-off");
}

extern "C" int vbnn_comm_unique_id(void* id_out) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(id_out, "null id");
    static_assert(VBNN_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id the host broadcasts is RCCL's unique id");
    int st = bind_rccl();
    if (st != VBNN_OK) return st;
    ncclUniqueId id;
    VBNN_CHECK_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
    return VBNN_OK;
    VBNN_API_END
}

extern "C" int vbnn_comm_create(vbnn_ctx* ctx, int rank, int world, const void* id, vbnn_comm** out) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(ctx && id && out, "null argument");
    VBNN_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank / world");
    int st = bind_rccl();
    if (st != VBNN_OK) return st;
    VBNN_CHECK_HIP(hipSetDevice(ctx->device));
    ncclUniqueId uid;
    memcpy(uid.internal, id, NCCL_UNIQUE_ID_BYTES);
    vbnn_comm* c = new vbnn_comm();
    c->ctx = ctx; c->rank = rank; c->world = world; c->pending = 0; c->trig = nullptr; c->trig_count = 0; c->done_count = 0;
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, uid, rank);
    if (r != ncclSuccess) {
        vbnn_set_error("ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, world, ctx->device, g_rccl.GetErrorString(r));
        delete c;
        return VBNN_ERR_HIP;
    }
    int least = 0, greatest = 0;                               // the exchange goes first whenever a CU frees up
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    hipError_t e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, greatest);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->done, hipEventDisableTiming);
    {
        // (the polled hand-offs need the two streams on different hardware queues, i.e. of different priority: p2p.hip)
        const char* ev = getenv("VBNN_COMM_FLAG_TRIGGER");
        int prio = least;
        const bool same_prio = (ctx->stream && hipStreamGetPriority(ctx->stream, &prio) == hipSuccess && prio == greatest) || greatest == least;
        if (e == hipSuccess && !(ev && ev[0] == '0') && !same_prio) {
            e = hipExtMallocWithFlags((void**)&c->trig, 4096, hipDeviceMallocUncached);
            if (e == hipSuccess) e = hipMemset(c->trig, 0, 4096);
        }
    }
    if (e != hipSuccess) {
        vbnn_set_error("exchange stream / events: %s", hipGetErrorString(e));
        (void)g_rccl.CommDestroy(c->comm);
        delete c;
        return VBNN_ERR_HIP;
    }
    *out = c;
    return VBNN_OK;
    VBNN_API_END
}

extern "C" int vbnn_comm_destroy(vbnn_comm* c) {
    VBNN_API_BEGIN
    if (!c) return VBNN_OK;
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->stream);
    (void)g_rccl.CommDestroy(c->comm);
    (void)hipEventDestroy(c->ready);
    (void)hipEventDestroy(c->done);
    (void)hipStreamDestroy(c->stream);
    if (c->trig) (void)hipFree(c->trig);
    delete c;
    return VBNN_OK;
    VBNN_API_END
}

extern "C" int vbnn_comm_info(vbnn_comm* c, int* rank, int* world, int* ranks_in_comm) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(c, "null comm");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (ranks_in_comm) VBNN_CHECK_NCCL(g_rccl.CommCount(c->comm, ranks_in_comm));
    return VBNN_OK;
    VBNN_API_END
}

extern "C" int vbnn_allreduce_grads(vbnn_comm* c, float* buf, int64_t n) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(c && buf && n > 0, "argument");
    // everything enqueued on the compute stream so far (the accGradParameters launch that fills `buf`) comes first
    { const int hst = comm_handoff(c); if (hst != VBNN_OK) return hst; }
    VBNN_CHECK_NCCL(g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, c->comm, c->stream));
    c->pending += 1;
    return VBNN_OK;
    VBNN_API_END
}

extern "C" int vbnn_allreduce_grads_bf16(vbnn_comm* c, void* buf, int64_t n) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(c && buf && n > 0, "argument");
    { const int hst = comm_handoff(c); if (hst != VBNN_OK) return hst; }
    VBNN_CHECK_NCCL(g_rccl.AllReduce(buf, buf, (size_t)n, ncclBfloat16, ncclSum, c->comm, c->stream));
    c->pending += 1;
    return VBNN_OK;
    VBNN_API_END
}

extern "C" int vbnn_comm_finish(vbnn_comm* c) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(c, "null comm");
    if (c->pending == 0) return VBNN_OK;
    // no host wait: the compute stream's NEXT launch (the update, the next minibatch) is ordered behind the exchange -- through the
    // trigger page's second word (bumped on the exchange stream, polled by a one-wave kernel on the compute stream), or an event
    if (c->trig) {
        c->done_count += 1;
        c->pending = 0;
        return comm_signal_wait(c->stream, c->ctx->stream, c->trig + 1, c->done_count, "vbnn_comm_finish");
    }
    VBNN_CHECK_HIP(hipEventRecord(c->done, c->stream));
    VBNN_CHECK_HIP(hipStreamWaitEvent(c->ctx->stream, c->done, 0));
    c->pending = 0;
    return VBNN_OK;
    VBNN_API_END
}

// one 8-byte word per rank, gathered on the exchange path itself: the host proves with it that `world` distinct
// processes / devices really took part (bench.py's `ranks_seen`)
extern "C" int vbnn_comm_allgather_u64(vbnn_comm* c, const uint64_t* mine_dev, uint64_t* all_dev) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(c && mine_dev && all_dev, "argument");
    { const int hst = comm_handoff(c); if (hst != VBNN_OK) return hst; }
    VBNN_CHECK_NCCL(g_rccl.AllGather(mine_dev, all_dev, 1, ncclUint64, c->comm, c->stream));
    c->pending += 1;
    return VBNN_OK;
    VBNN_API_END
}

// ---- the two halves of the all-reduce as calls of their own (the sharded-update exchange, include/vbnn_hip.h): in place, on the
// exchange stream, ordered behind the compute stream like vbnn_allreduce_grads
extern "C" int vbnn_comm_reduce_scatter(vbnn_comm* c, float* buf, int64_t n_per_rank) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(c && buf && n_per_rank > 0, "argument");
    { const int hst = comm_handoff(c); if (hst != VBNN_OK) return hst; }
    VBNN_CHECK_NCCL(g_rccl.ReduceScatter(buf, buf + (size_t)c->rank * (size_t)n_per_rank, (size_t)n_per_rank, ncclFloat32, ncclSum, c->comm, c->stream));
    c->pending += 1;
    return VBNN_OK;
    VBNN_API_END
}

extern "C" int vbnn_comm_all_gather(vbnn_comm* c, void* buf, int64_t bytes_per_rank) {
    VBNN_API_BEGIN
    VBNN_REQUIRE(c && buf && bytes_per_rank > 0, "argument");
    { const int hst = comm_handoff(c); if (hst != VBNN_OK) return hst; }
    char* b = static_cast<char*>(buf);
    VBNN_CHECK_NCCL(g_rccl.AllGather(b + (size_t)c->rank * (size_t)bytes_per_rank, b, (size_t)bytes_per_rank, ncclInt8, c->comm, c->stream));
    c->pending += 1;
    return VBNN_OK;
    VBNN_API_END
}
