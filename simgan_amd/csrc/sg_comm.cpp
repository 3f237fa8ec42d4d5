// sg_comm.cpp -- data-parallel collectives over RCCL (xGMI), one process per GPU.
//
// The reference is single-process (a2c/main_gail_dyn_ppo.py:64); data parallelism over env
// columns is added per BASELINE.json north_star.  RCCL is loaded with
// dlopen so a single-GPU run has no dependency on it.  Collectives are enqueued on the
// library's own stream, between the gradient kernels and the optimizer kernels, with no host
// synchronisation.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include "sg_common.h"

// Minimal RCCL surface (ABI-stable NCCL 2 API).
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5,
               ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;

struct SgRccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
};

struct SgComm {
    ncclComm_t comm = nullptr;
};

static SgRccl g_rccl;

static int rccl_load() {
    if (g_rccl.handle) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.handle) break;
    }
    SG_REQUIRE(g_rccl.handle, "RCCL not found (dlopen librccl.so.1 failed: %s)", dlerror());
#define SG_SYM(field, name)                                                         \
    *(void**)(&g_rccl.field) = dlsym(g_rccl.handle, name);                          \
    SG_REQUIRE(g_rccl.field, "RCCL symbol %s missing", name)
    SG_SYM(GetUniqueId, "ncclGetUniqueId");
    SG_SYM(CommInitRank, "ncclCommInitRank");
    SG_SYM(CommDestroy, "ncclCommDestroy");
    SG_SYM(AllReduce, "ncclAllReduce");
    SG_SYM(AllGather, "ncclAllGather");
    SG_SYM(GetErrorString, "ncclGetErrorString");
    SG_SYM(CommCount, "ncclCommCount");
    SG_SYM(CommUserRank, "ncclCommUserRank");
#undef SG_SYM
    return 0;
}

#define SG_NCCL(expr)                                                                              \
    do {                                                                                           \
        ncclResult_t _r = (expr);                                                                  \
        if (_r != ncclSuccess) {                                                                   \
            sg_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, g_rccl.GetErrorString(_r)); \
            return -3;                                                                             \
        }                                                                                          \
    } while (0)

extern "C" int sg_comm_unique_id(uint8_t id[128]) {
    SG_REQUIRE(id, "sg_comm_unique_id: id is NULL");
    SG_TRY(rccl_load());
    ncclUniqueId u;
    SG_NCCL(g_rccl.GetUniqueId(&u));
    memcpy(id, u.internal, 128);
    return 0;
}

extern "C" int sg_ctx_comm_init(sg_ctx* ctx, const uint8_t id[128], int rank, int world) {
    SG_REQUIRE(ctx && id, "sg_ctx_comm_init: NULL argument");
    SG_REQUIRE(world >= 1 && rank >= 0 && rank < world, "sg_ctx_comm_init: bad rank %d / world %d", rank, world);
    SG_REQUIRE(!ctx->comm, "sg_ctx_comm_init: communicator already initialised");
    SG_TRY(rccl_load());
    SG_CHECK(hipSetDevice(ctx->device));
    ncclUniqueId u;
    memcpy(u.internal, id, 128);
    SgComm* c = new SgComm();
    SG_NCCL(g_rccl.CommInitRank(&c->comm, world, u, rank));
    ctx->comm = c;
    ctx->rank = rank;
    ctx->world = world;
    // SG_COMM_ALWAYS=1 keeps the collectives in the launch sequence even for a single rank (where
    // they are the identity): the 1-GPU self-test of the RCCL path (tests/test_gpu_comm.py)
    const char* always = getenv("SG_COMM_ALWAYS");
    ctx->use_comm = world > 1 || (always && always[0] == '1');
    const char* dp = getenv("SG_DISC_DP");
    ctx->disc_sharded = dp && strcmp(dp, "sharded") == 0;
    // One tiny all-reduce now: RCCL sets up its channels and proxy connections lazily on the first collective, which must
    // not happen inside a stream capture (the updates capture their collectives into hipGraphs).
    float* warm = nullptr;
    SG_TRY(sg_ctx_scratch(ctx, 64, &warm));
    SG_CHECK(hipMemsetAsync(warm, 0, 64, ctx->stream));
    SG_NCCL(g_rccl.AllReduce(warm, warm, 4, ncclFloat32, ncclSum, c->comm, ctx->stream));
    SG_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int sg_ctx_comm_info(sg_ctx* ctx, int* rank, int* world) {
    SG_REQUIRE(ctx, "sg_ctx_comm_info: ctx is NULL");
    if (rank) *rank = ctx->rank;
    if (world) *world = ctx->world;
    if (ctx->comm) {   // what the communicator itself reports (ncclCommUserRank / ncclCommCount), not what it was asked for
        int r = -1, n = -1;
        SG_NCCL(g_rccl.CommUserRank(ctx->comm->comm, &r));
        SG_NCCL(g_rccl.CommCount(ctx->comm->comm, &n));
        SG_REQUIRE(r == ctx->rank && n == ctx->world, "RCCL reports rank %d of %d, the context was initialised as %d of %d", r, n,
                   ctx->rank, ctx->world);
        if (rank) *rank = r;
        if (world) *world = n;
    }
    return 0;
}

extern "C" int sg_ctx_set_disc_dp(sg_ctx* ctx, int sharded) {
    SG_REQUIRE(ctx, "sg_ctx_set_disc_dp: ctx is NULL");
    ctx->disc_sharded = sharded != 0;
    return 0;
}

int sg_comm_allreduce_f32(sg_ctx* ctx, float* dev, int64_t n) {
    SG_REQUIRE(ctx->comm, "all-reduce requested but no communicator (call sg_ctx_comm_init)");
    SG_NCCL(g_rccl.AllReduce(dev, dev, (size_t)n, ncclFloat32, ncclSum, ctx->comm->comm, ctx->stream));
    return 0;
}

int sg_comm_allreduce_f64(sg_ctx* ctx, double* dev, int64_t n) {
    SG_REQUIRE(ctx->comm, "all-reduce requested but no communicator (call sg_ctx_comm_init)");
    SG_NCCL(g_rccl.AllReduce(dev, dev, (size_t)n, ncclFloat64, ncclSum, ctx->comm->comm, ctx->stream));
    return 0;
}

int sg_comm_allgather_f32(sg_ctx* ctx, const float* dev_in, float* dev_out, int64_t n_per_rank) {
    SG_REQUIRE(ctx->comm, "all-gather requested but no communicator (call sg_ctx_comm_init)");
    SG_NCCL(g_rccl.AllGather(dev_in, dev_out, (size_t)n_per_rank, ncclFloat32, ctx->comm->comm, ctx->stream));
    return 0;
}
