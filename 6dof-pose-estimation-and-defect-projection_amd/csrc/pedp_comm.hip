// Collectives of the sharded hot path, issued by the library itself on the context's stream:
// RCCL over xGMI, one communicator per context (one process per GPU).
//
//   rays   contiguous ray blocks per rank -> one all-gather of the 8-byte hit records
//   ICP    scene shards per rank          -> one all-reduce (sum) of the 29-double packet per pass
//
// RCCL is bound at run time (dlopen): the process normally already holds a copy -- PyTorch-ROCm
// ships its own librccl with the same SONAME as /opt/rocm's -- and the library must load and
// export every symbol on a box without RCCL too.  The unique id travels between the ranks by
// whatever the host side uses for rendezvous (pedp_hip.dist: torch.distributed's store).
#include "pedp_internal.h"
#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>
#include <new>

namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

RcclApi g_rccl;
std::mutex g_rccl_mutex;  // contexts of different threads may create their communicators at once

int rccl_load() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.ok) return PEDP_OK;
    // a copy the process already holds first (torch's), then the system one
    const char *names[] = {"librccl.so.1", "librccl.so"};
    void *h = nullptr;
    for (const char *n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;
    if (!h)
        for (const char *n : names)
            if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        pedp_set_error("pedp_comm: librccl not found (%s)", dlerror());
        return PEDP_ERR_COLLECTIVE;
    }
    g_rccl.handle = h;
#define PEDP_SYM(field, name)                                                    \
    *(void **)(&g_rccl.field) = dlsym(h, name);                                  \
    if (!g_rccl.field) {                                                         \
        pedp_set_error("pedp_comm: librccl lacks %s", name);                     \
        return PEDP_ERR_COLLECTIVE;                                              \
    }
    PEDP_SYM(GetUniqueId, "ncclGetUniqueId")
    PEDP_SYM(CommInitRank, "ncclCommInitRank")
    PEDP_SYM(CommDestroy, "ncclCommDestroy")
    PEDP_SYM(AllReduce, "ncclAllReduce")
    PEDP_SYM(AllGather, "ncclAllGather")
    PEDP_SYM(GetErrorString, "ncclGetErrorString")
#undef PEDP_SYM
    g_rccl.ok = true;
    return PEDP_OK;
}

#define PEDP_NCCL_CHECK(expr)                                                                          \
    do {                                                                                               \
        ncclResult_t r_ = (expr);                                                                      \
        if (r_ != ncclSuccess) {                                                                       \
            pedp_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, g_rccl.GetErrorString(r_));   \
            return PEDP_ERR_COLLECTIVE;                                                                \
        }                                                                                              \
    } while (0)

}  // namespace

struct pedp_comm_s {
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0;
};

int pedp_comm_allreduce_sum_f64(pedp_ctx_t c, double *buf, int64_t n) {
    PEDP_REQUIRE(c && c->comm && c->comm->comm, "pedp_comm: the context has no communicator");
    PEDP_NCCL_CHECK(g_rccl.AllReduce(buf, buf, (size_t)n, ncclDouble, ncclSum, c->comm->comm, c->stream));
    return PEDP_OK;
}

extern "C" {

int pedp_comm_unique_id(uint8_t id[PEDP_COMM_ID_BYTES]) {
    PEDP_REQUIRE(id, "pedp_comm_unique_id: null output");
    static_assert(PEDP_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
    int rc = rccl_load();
    if (rc) return rc;
    ncclUniqueId u;
    PEDP_NCCL_CHECK(g_rccl.GetUniqueId(&u));
    memcpy(id, u.internal, PEDP_COMM_ID_BYTES);
    return PEDP_OK;
}

int pedp_comm_create(pedp_ctx_t c, const uint8_t id[PEDP_COMM_ID_BYTES], int nranks, int rank) {
    PEDP_REQUIRE(c && id, "pedp_comm_create: null argument");
    PEDP_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "pedp_comm_create: rank %d of %d", rank, nranks);
    PEDP_REQUIRE(!c->comm, "pedp_comm_create: the context already has a communicator");
    int rc = rccl_load();
    if (rc) return rc;
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    pedp_comm_s *m = new (std::nothrow) pedp_comm_s();
    if (!m) { pedp_set_error("pedp_comm_create: out of host memory"); return PEDP_ERR_ALLOC; }
    ncclUniqueId u;
    memcpy(u.internal, id, PEDP_COMM_ID_BYTES);
    ncclResult_t r = g_rccl.CommInitRank(&m->comm, nranks, u, rank);
    if (r != ncclSuccess) {
        pedp_set_error("pedp_comm_create: ncclCommInitRank -> %s", g_rccl.GetErrorString(r));
        delete m;
        return PEDP_ERR_COLLECTIVE;
    }
    m->nranks = nranks;
    m->rank = rank;
    c->comm = m;
    return PEDP_OK;
}

int pedp_comm_destroy(pedp_ctx_t c) {
    PEDP_REQUIRE(c, "pedp_comm_destroy: null context");
    if (!c->comm) return PEDP_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm->comm && g_rccl.ok) (void)g_rccl.CommDestroy(c->comm->comm);
    delete c->comm;
    c->comm = nullptr;
    return PEDP_OK;
}

int pedp_comm_size(pedp_ctx_t c, int *nranks, int *rank) {
    PEDP_REQUIRE(c, "pedp_comm_size: null context");
    if (nranks) *nranks = c->comm ? c->comm->nranks : 1;
    if (rank) *rank = c->comm ? c->comm->rank : 0;
    return PEDP_OK;
}

int pedp_comm_allgather(pedp_ctx_t c, const void *send, void *recv, int64_t bytes_per_rank) {
    PEDP_REQUIRE(c && send && recv && bytes_per_rank >= 0, "pedp_comm_allgather: bad argument");
    PEDP_REQUIRE(c->comm && c->comm->comm, "pedp_comm_allgather: the context has no communicator");
    if (bytes_per_rank == 0) return PEDP_OK;
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    PEDP_NCCL_CHECK(g_rccl.AllGather(send, recv, (size_t)bytes_per_rank, ncclUint8, c->comm->comm, c->stream));
    return PEDP_OK;
}

int pedp_comm_allreduce_f64(pedp_ctx_t c, double *buf, int64_t n) {
    PEDP_REQUIRE(c && buf && n >= 0, "pedp_comm_allreduce_f64: bad argument");
    if (n == 0) return PEDP_OK;
    PEDP_HIP_CHECK(hipSetDevice(c->device));
    return pedp_comm_allreduce_sum_f64(c, buf, n);
}

}  // extern "C"
