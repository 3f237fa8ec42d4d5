// sg_comm.cpp -- data-parallel collectives over RCCL (xGMI), one process per GPU.
//
// The reference is single-process (a2c/main_gail_dyn_ppo.py:64); data parallelism over env
// columns is added per BASELINE.json north_star.  RCCL is loaded with
// dlopen so a single-GPU run has no dependency on it.  Collectives are enqueued on the
// library's own stream, between the gradient kernels and the optimizer kernels, with no host
// synchronisation.
//
// A second transport, the LOOPBACK communicator, carries the same three collectives between `world` contexts that
// live on ONE host -- threads of one process or separate processes, on different devices or all on the same one --
// through a POSIX shared-memory segment (device -> segment, host barrier, every rank sums / concatenates the ranks'
// slots in rank order, segment -> device).  It exists so that every world > 1 branch of the library (row sharding,
// global statistics, rank slicing of the discriminator batch, the replicated mode's all-gather) can be executed and
// checked against the oracle on a one-GPU box, and so that bench.py --gpus N can be self-tested there; it synchronises
// the stream, so it is never captured into a hipGraph (sg_comm_graph_ok) and it is not a performance path.
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <vector>

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

// ------------------------------------------------------------------------------------------- loopback transport
static const char LB_MAGIC[8] = {'S', 'G', 'L', 'O', 'O', 'P', 'B', 'K'};

struct LbHeader {                        // at offset 0 of the shared segment
    std::atomic<uint32_t> ready;         // 0x5347 once rank 0 has initialised the header
    std::atomic<uint32_t> arrived;       // barrier: ranks that have arrived in the current generation
    std::atomic<uint32_t> generation;    // barrier: bumped by the last rank to arrive
    std::atomic<uint32_t> failed;        // a rank timed out or hit an error: every waiter gives up
    uint32_t world;
    uint32_t pad;
    uint64_t slot_bytes;                 // bytes per (parity, rank) slot
    std::atomic<uint32_t> attached[64];  // rank -> 1 once it has mapped the segment
};

struct SgLoopback {
    LbHeader* hdr = nullptr;
    uint8_t* data = nullptr;             // [2 parities][world][slot_bytes]
    size_t map_bytes = 0;
    uint64_t seq = 0;                    // collectives issued so far (slot parity = seq & 1)
    void* h_tmp = nullptr;               // private pinned staging for the reduced / gathered result
    size_t tmp_bytes = 0;
    double timeout_s = 120.0;
};

// ------------------------------------------------------------------------------------------- peer mesh (SG_COMM_PEER=1)
// The small float32 all-reduce of an optimizer step (one per PPO step, one per sharded discriminator step: 60-300 KB, 160 to
// thousands of times per update) as ONE kernel over peer-mapped device memory instead of a library collective (SURVEY.md
// section 5 "one-shot peer-write all-reduce"): every rank owns a slot buffer that all ranks have mapped (hipIpc between
// processes, the plain pointer between contexts of one process); a collective is
//     each workgroup: write its chunk of the rank's vector into EVERY rank's slot [parity][rank] (system-scope stores),
//                     drain them, raise its flag word [parity][rank][workgroup] on every rank,
//                     wait for the `world` flag words of its chunk in its OWN buffer, add the `world` slots in rank order
// -- one launch, no host involvement, capturable into a hipGraph, bit-identical on every rank (fixed order, the loopback
// transport's order).  The wait is one-way per rank pair and bounded by the wall clock; slots and flags are double-buffered by
// the parity of the collective's number, which is enough because no rank can start collective s + 2 before every rank has
// finished reading collective s (it has seen their flags of s + 1).  The base communicator (RCCL or loopback) stays for set-up
// (the exchange of the memory handles is one all-gather over it), the float64 statistics and the all-gather.
// OPT-IN and, across GPUs, UNTESTED: a gpurun box has one GPU, so what runs there is ranks sharing a device (threads: plain
// pointers; processes: hipIpc handles) -- the same kernel and protocol, local HBM instead of xGMI.
#define SG_PEER_MAX_WORLD 16
#define SG_PEER_MAX_BLOCKS 128                  // = SG_PEER_MAX_FLOATS / 1024: one workgroup per 1024 floats
#define SG_PEER_MAX_FLOATS (128 * 1024)      // per rank and collective: 512 KB
#define SG_PEER_FLAG_STRIDE 32               // words: one 128-byte line per flag
#define SG_PEER_TIMEOUT_TICKS 2000000000ll   // 20 s of the 100 MHz wall clock: a peer may be a whole kernel queue behind

struct SgPeerDev {                           // kernel argument, by value
    float* slots[SG_PEER_MAX_WORLD];         // rank r's slot buffer as mapped HERE: [2][world][SG_PEER_MAX_FLOATS]
    unsigned* flags[SG_PEER_MAX_WORLD];      // rank r's flag words as mapped here: [2][world][SG_PEER_MAX_BLOCKS] lines
    unsigned* ctrl;                          // own: {collectives completed, -, error}
    int rank, world;
};

struct SgPeer {
    SgPeerDev dev;
    void* own = nullptr;                     // this rank's allocation (slots | flags | ctrl)
    void* mapped[SG_PEER_MAX_WORLD] = {nullptr};   // hipIpcOpenMemHandle results (other processes' buffers)
    uint64_t launches = 0;
    uint32_t generation = 0;                 // process-wide, never 0, never reused: keys the hipGraphs that captured this mesh's pointers
};

struct SgComm {
    ncclComm_t comm = nullptr;
    SgLoopback* lb = nullptr;
    SgPeer* peer = nullptr;
};

static double lb_now() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

// Sense-reversing barrier over the segment's atomics (lock-free 32-bit atomics are address-free: valid across processes).
static int lb_barrier(SgLoopback* lb, const char* what) {
    LbHeader* h = lb->hdr;
    const uint32_t gen = h->generation.load(std::memory_order_acquire);
    if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == h->world) {
        h->arrived.store(0, std::memory_order_relaxed);
        h->generation.fetch_add(1, std::memory_order_acq_rel);
        return 0;
    }
    const double t0 = lb_now();
    for (unsigned spin = 0; h->generation.load(std::memory_order_acquire) == gen; ++spin) {
        if (h->failed.load(std::memory_order_relaxed)) { sg_set_error("loopback communicator: a peer rank failed (%s)", what); return -3; }
        if ((spin & 63) == 63) {
            sched_yield();
            if (lb_now() - t0 > lb->timeout_s) {
                h->failed.store(1, std::memory_order_relaxed);
                sg_set_error("loopback communicator: %s timed out after %.0f s waiting for the other ranks", what, lb->timeout_s);
                return -3;
            }
        }
    }
    return 0;
}

static void lb_name(const uint8_t id[128], char out[64]) {
    static const char* hex = "0123456789abcdef";
    int n = snprintf(out, 64, "/sglb_");
    for (int i = 8; i < 24; ++i) { out[n++] = hex[id[i] >> 4]; out[n++] = hex[id[i] & 15]; }
    out[n] = 0;
}

static int lb_open_inner(SgLoopback* lb, const uint8_t id[128], int rank, int world, bool* created) {
    SG_REQUIRE(world <= 64, "loopback communicator: at most 64 ranks");
    char name[64];
    lb_name(id, name);
    if (const char* e = getenv("SG_LOOPBACK_TIMEOUT_S")) { const double v = atof(e); if (v > 0) lb->timeout_s = v; }
    size_t slot = (size_t)32 << 20;
    if (const char* e = getenv("SG_LOOPBACK_SLOT_MB")) { const long v = atol(e); if (v > 0) slot = (size_t)v << 20; }
    const size_t hdr_bytes = 4096;
    int fd = -1;
    const double t0 = lb_now();
    if (rank == 0) {
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        SG_REQUIRE(fd >= 0, "loopback communicator: shm_open(%s) failed: %s", name, strerror(errno));
        *created = true;
        lb->map_bytes = hdr_bytes + 2 * (size_t)world * slot;     // sparse: pages are committed when first touched
        if (ftruncate(fd, (off_t)lb->map_bytes) != 0) { close(fd); SG_REQUIRE(false, "loopback communicator: ftruncate failed: %s", strerror(errno)); }
    } else {
        for (;;) {   // rank 0 creates the segment; wait until it exists and has its final size
            fd = shm_open(name, O_RDWR, 0600);
            struct stat sb;
            if (fd >= 0 && fstat(fd, &sb) == 0 && (size_t)sb.st_size > hdr_bytes) { lb->map_bytes = (size_t)sb.st_size; break; }
            if (fd >= 0) { close(fd); fd = -1; }
            SG_REQUIRE(lb_now() - t0 < lb->timeout_s, "loopback communicator: rank %d never saw rank 0's segment %s", rank, name);
            usleep(1000);
        }
    }
    void* m = mmap(nullptr, lb->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    SG_REQUIRE(m != MAP_FAILED, "loopback communicator: mmap of %zu bytes failed: %s", lb->map_bytes, strerror(errno));
    lb->hdr = reinterpret_cast<LbHeader*>(m);
    lb->data = reinterpret_cast<uint8_t*>(m) + hdr_bytes;
    if (rank == 0) {
        lb->hdr->world = (uint32_t)world;
        lb->hdr->slot_bytes = slot;
        lb->hdr->ready.store(0x5347, std::memory_order_release);
    } else {
        while (lb->hdr->ready.load(std::memory_order_acquire) != 0x5347) {
            SG_REQUIRE(lb_now() - t0 < lb->timeout_s, "loopback communicator: rank 0 never initialised the segment");
            usleep(200);
        }
        SG_REQUIRE(lb->hdr->world == (uint32_t)world, "loopback communicator: rank 0 created a world of %u, rank %d was given %d",
                   lb->hdr->world, rank, world);
    }
    SG_REQUIRE(lb->hdr->attached[rank].exchange(1) == 0, "loopback communicator: rank %d attached twice", rank);
    SG_TRY(lb_barrier(lb, "communicator set-up"));
    return 0;
}

// The segment's NAME is needed only until every rank has mapped it; rank 0 removes it on success (the memory lives until the
// last munmap) and on EVERY failure after it created it -- a failed mmap, a peer that died or timed out before the set-up
// barrier -- so a broken start-up leaves nothing behind in /dev/shm.
static int lb_open(SgLoopback* lb, const uint8_t id[128], int rank, int world) {
    bool created = false;
    const int rc = lb_open_inner(lb, id, rank, world, &created);
    if (created) {
        char name[64];
        lb_name(id, name);
        shm_unlink(name);
    }
    return rc;
}

static int lb_tmp(SgLoopback* lb, size_t bytes) {
    if (bytes <= lb->tmp_bytes) return 0;
    if (lb->h_tmp) SG_CHECK(sg_host_release(lb->h_tmp));
    lb->tmp_bytes = bytes + bytes / 2 + 4096;
    SG_CHECK(sg_host_malloc(&lb->h_tmp, lb->tmp_bytes));
    return 0;
}

// One collective over the segment.  op 0: element-wise sum (rank order 0..world-1, the same on every rank, so all ranks
// hold bit-identical results, like a ring all-reduce); op 1: concatenation in rank order.  esize 4 (float) or 8 (double).
static int lb_collective(sg_ctx* ctx, int op, const void* dev_in, void* dev_out, size_t n, int esize) {
    SgLoopback* lb = ctx->comm->lb;
    const int world = ctx->world, rank = ctx->rank;
    const size_t slot = lb->hdr->slot_bytes, chunk_elems = slot / (size_t)esize;
    SG_CHECK(hipSetDevice(ctx->device));
    SG_REQUIRE(n > 0, "loopback communicator: empty collective");
    for (size_t off = 0; off < n; off += chunk_elems) {
        const size_t cnt = n - off < chunk_elems ? n - off : chunk_elems, bytes = cnt * (size_t)esize;
        uint8_t* base = lb->data + (lb->seq & 1) * (size_t)world * slot;
        lb->seq += 1;
        SG_CHECK(hipMemcpyAsync(base + (size_t)rank * slot, (const uint8_t*)dev_in + off * esize, bytes, hipMemcpyDeviceToHost, ctx->stream));
        SG_CHECK(hipStreamSynchronize(ctx->stream));
        // Double-buffered slots: a rank overwrites parity p again only after the barrier of the collective in between,
        // i.e. after every rank has finished reading the previous contents -- one barrier per collective is enough.
        SG_TRY(lb_barrier(lb, op == 0 ? "all-reduce" : "all-gather"));
        if (op == 0) {
            SG_TRY(lb_tmp(lb, bytes));
            if (esize == 4) {
                float* acc = (float*)lb->h_tmp;
                memcpy(acc, base, bytes);
                for (int r = 1; r < world; ++r) { const float* s = (const float*)(base + (size_t)r * slot); for (size_t i = 0; i < cnt; ++i) acc[i] += s[i]; }
            } else {
                double* acc = (double*)lb->h_tmp;
                memcpy(acc, base, bytes);
                for (int r = 1; r < world; ++r) { const double* s = (const double*)(base + (size_t)r * slot); for (size_t i = 0; i < cnt; ++i) acc[i] += s[i]; }
            }
            SG_CHECK(hipMemcpyAsync((uint8_t*)dev_out + off * esize, lb->h_tmp, bytes, hipMemcpyHostToDevice, ctx->stream));
            SG_CHECK(hipStreamSynchronize(ctx->stream));
        } else {
            // rank r's chunk [off, off+cnt) of its n elements lands at out[r*n + off]
            for (int r = 0; r < world; ++r)
                SG_CHECK(hipMemcpyAsync((uint8_t*)dev_out + ((size_t)r * n + off) * esize, base + (size_t)r * slot, bytes, hipMemcpyHostToDevice, ctx->stream));
            SG_CHECK(hipStreamSynchronize(ctx->stream));
        }
    }
    return 0;
}

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

// An id for the loopback transport: the magic tag + 16 random bytes that name the shared segment.  Drawn on rank 0 and
// handed to every rank exactly like an RCCL id; sg_ctx_comm_init recognises the tag.
extern "C" int sg_comm_loopback_id(uint8_t id[128]) {
    SG_REQUIRE(id, "sg_comm_loopback_id: id is NULL");
    memset(id, 0, 128);
    memcpy(id, LB_MAGIC, 8);
    FILE* f = fopen("/dev/urandom", "rb");
    const bool ok = f && fread(id + 8, 1, 16, f) == 16;
    if (f) fclose(f);
    if (!ok) {   // no urandom: pid + clock
        uint64_t v[2] = {(uint64_t)getpid() * 0x9E3779B97F4A7C15ull, (uint64_t)(lb_now() * 1e9)};
        memcpy(id + 8, v, 16);
    }
    return 0;
}

// system-scope (sc0 sc1) accesses: past every cache of this GPU and of the peer
typedef unsigned int sg_pu4 __attribute__((ext_vector_type(4)));
#define SG_PEER_AUX 17   // sc0 | sc1
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sg_peer_rsrc(const float* p, int floats) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, floats * 4, 0x00020000);
}

// grid = ceil(n / 1024) workgroups of 256 lanes: every lane owns ONE 16-byte piece of the vector (n4 = n / 4 whole pieces;
// the 0..3 floats of a ragged tail go through lane 0 of the last workgroup one by one)
__global__ __launch_bounds__(256) void k_peer_allreduce(SgPeerDev P, float* buf, int n) {
    __shared__ int sh_ok;
    const int b = blockIdx.x, nb = gridDim.x, tid = threadIdx.x, W = P.world, me = P.rank;
    // collectives completed so far: committed by workgroup 0 of the previous collective once all of that launch's workgroups
    // had read it (see 6.), and behind a kernel boundary
    const unsigned s = __hip_atomic_load(P.ctrl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    const int par = (int)(s & 1u);
    const int n4 = n >> 2, i4 = b * 256 + tid;
    const bool mine = i4 < n4, tail = (n & 3) != 0 && b == nb - 1 && tid == 0;
    // 1. this rank's piece into every rank's slot [par][me] (own slot included: the sum below reads slots only)
    sg_pu4 v = sg_pu4{0u, 0u, 0u, 0u};
    if (mine) v = __builtin_amdgcn_raw_buffer_load_b128(sg_peer_rsrc(buf, n4 * 4), i4 * 16, 0, 0);
    for (int q = 0; q < W; ++q) {
        const int r = (me + 1 + q) % W;      // start with the neighbour: the ranks do not all hit rank 0's memory first
        float* dst = P.slots[r] + ((size_t)(par * W + me)) * SG_PEER_MAX_FLOATS;
        if (mine) __builtin_amdgcn_raw_buffer_store_b128(v, sg_peer_rsrc(dst, n4 * 4), i4 * 16, 0, SG_PEER_AUX);
        if (tail) for (int i = n4 * 4; i < n; ++i) __hip_atomic_store(dst + i, buf[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // 2. the stores are acknowledged before the flag is raised (spelt out: a barrier alone implies no vmcnt wait on gfx950)
    //    (no __threadfence_system(): the stores above ARE system-scope write-through stores, and a fence would write back and
    //    invalidate this XCD's whole L2 once per workgroup -- measured: 10 us per collective at 55 workgroups instead of 4)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) sh_ok = 1;
    __syncthreads();
    // 3. one flag word per (parity, writer, workgroup) on every rank
    if (tid < W) __hip_atomic_store(P.flags[tid] + ((size_t)(par * W + me) * SG_PEER_MAX_BLOCKS + b) * SG_PEER_FLAG_STRIDE, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // 4. wait for every rank's piece b in this rank's own buffer
    if (tid < W) {
        const unsigned* f = P.flags[me] + ((size_t)(par * W + tid) * SG_PEER_MAX_BLOCKS + b) * SG_PEER_FLAG_STRIDE;
        const long long deadline = wall_clock64() + SG_PEER_TIMEOUT_TICKS;
        for (int it = 0; __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != s; ++it) {
            if (__hip_atomic_load(P.ctrl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || ((it & 63) == 63 && wall_clock64() > deadline)) { sh_ok = 0; break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    // 5. the sum in rank order, all `world` loads in flight (a time-out leaves NaN and raises the error word: sticky, later
    //    collectives give up at once)
    const float* src = P.slots[me] + (size_t)par * W * SG_PEER_MAX_FLOATS;
    if (sh_ok) {
        if (mine) {
            sg_pu4 x[SG_PEER_MAX_WORLD];
#pragma unroll
            for (int r = 0; r < SG_PEER_MAX_WORLD; ++r)
                if (r < W) x[r] = __builtin_amdgcn_raw_buffer_load_b128(sg_peer_rsrc(src + (size_t)r * SG_PEER_MAX_FLOATS, n4 * 4), i4 * 16, 0, SG_PEER_AUX);
            float a0 = __uint_as_float(x[0].x), a1 = __uint_as_float(x[0].y), a2 = __uint_as_float(x[0].z), a3 = __uint_as_float(x[0].w);
#pragma unroll
            for (int r = 1; r < SG_PEER_MAX_WORLD; ++r)
                if (r < W) { a0 += __uint_as_float(x[r].x); a1 += __uint_as_float(x[r].y); a2 += __uint_as_float(x[r].z); a3 += __uint_as_float(x[r].w); }
            *reinterpret_cast<float4*>(buf + (size_t)i4 * 4) = float4{a0, a1, a2, a3};
        }
        if (tail)
            for (int i = n4 * 4; i < n; ++i) {
                float acc = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                for (int r = 1; r < W; ++r) acc += __hip_atomic_load(src + (size_t)r * SG_PEER_MAX_FLOATS + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                buf[i] = acc;
            }
    } else {
        const float qnan = __builtin_nanf("");
        if (mine) *reinterpret_cast<float4*>(buf + (size_t)i4 * 4) = float4{qnan, qnan, qnan, qnan};
        if (tail) for (int i = n4 * 4; i < n; ++i) buf[i] = qnan;
        if (tid == 0) atomicOr(P.ctrl + 2, 1u);
    }
    // 6. workgroup 0 commits the collective's number -- once every workgroup of this launch has raised its own flag on this
    //    rank, i.e. has read the old number (no read-modify-write on a shared word: 60 of those cost microseconds).  A timed-out
    //    launch commits as well, so that the numbering stays aligned with the ranks that did not time out.
    if (b == 0 && tid < 64) {
        const long long deadline = wall_clock64() + SG_PEER_TIMEOUT_TICKS;
        for (int it = 0;; ++it) {
            bool ok = true;
            for (int j = tid; j < nb; j += 64)
                ok = ok & (__hip_atomic_load(P.flags[me] + ((size_t)(par * W + me) * SG_PEER_MAX_BLOCKS + j) * SG_PEER_FLAG_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == s);
            if (__all(ok) || ((it & 63) == 63 && wall_clock64() > deadline)) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (tid == 0) __hip_atomic_store(P.ctrl, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

struct SgPeerRecord {            // what the ranks tell each other about their buffers: 24 floats' worth of bytes
    hipIpcMemHandle_t handle;    // 64 bytes
    uint64_t pid, ptr;
    int32_t device, pad[3];
};
static_assert(sizeof(SgPeerRecord) == 96, "SgPeerRecord travels as 24 floats");

static void peer_release(sg_ctx* ctx, SgPeer* p) {
    (void)hipSetDevice(ctx->device);
    if (p->own && p->launches) {     // a collective that gave up left its word up (sticky): say so once, on the way out
        unsigned err = 0;
        const size_t off = sizeof(float) * 2 * (size_t)ctx->world * SG_PEER_MAX_FLOATS + sizeof(unsigned) * 2 * (size_t)ctx->world * SG_PEER_MAX_BLOCKS * SG_PEER_FLAG_STRIDE;
        if (hipMemcpy(&err, reinterpret_cast<uint8_t*>(p->own) + off + 8, sizeof err, hipMemcpyDeviceToHost) == hipSuccess && err)
            fprintf(stderr, "[simgan_hip] rank %d: a peer-mesh all-reduce gave up waiting for a peer's flags (20 s) during this run; the vectors of that step were NaN\n", ctx->rank);
        (void)hipGetLastError();
    }
    for (int r = 0; r < SG_PEER_MAX_WORLD; ++r) if (p->mapped[r]) (void)hipIpcCloseMemHandle(p->mapped[r]);
    if (p->own) (void)sg_dev_free(p->own);
    delete p;
}

// Builds the mesh over the base communicator that is already up.  Every rank must call it: it contains two collectives -- the
// all-gather of the buffer records and ONE status all-reduce at the end, which makes the outcome collective: either every
// rank leaves with the mesh or none does (a rank whose hipIpcOpenMemHandle failed used to be left on the base communicator
// while its peers ran the mesh kernel and waited 20 s for it).  A rank that fails locally still takes part in both.
// SG_COMM_PEER_FAIL_RANK=r (test hook): rank r behaves as if opening a peer's handle had failed.
static int peer_setup(sg_ctx* ctx) {
    const int W = ctx->world, me = ctx->rank;
    SG_REQUIRE(W <= SG_PEER_MAX_WORLD, "SG_COMM_PEER: at most %d ranks (world is %d)", SG_PEER_MAX_WORLD, W);
    SG_REQUIRE(!ctx->comm->peer, "SG_COMM_PEER: the mesh is already up");
    const size_t slot_b = sizeof(float) * 2 * (size_t)W * SG_PEER_MAX_FLOATS;
    const size_t flag_b = sizeof(unsigned) * 2 * (size_t)W * SG_PEER_MAX_BLOCKS * SG_PEER_FLAG_STRIDE;
    const size_t total = slot_b + flag_b + 256;
    SgPeer* p = new SgPeer();
    static std::atomic<uint32_t> next_generation{1};
    p->generation = next_generation.fetch_add(1);
    int rc = 0;            // this rank's own outcome; the error text of the first local failure is kept
    char why[512] = "";
    auto fail = [&](int code, const char* fmt, auto... args) {
        if (rc != 0) return;
        rc = code;
        if constexpr (sizeof...(args) == 0) snprintf(why, sizeof why, "%s", fmt);
        else snprintf(why, sizeof why, fmt, args...);
    };
    float *d_rec = nullptr, *d_all = nullptr;
    std::vector<SgPeerRecord> all((size_t)W);
    SgPeerRecord rec;
    memset(&rec, 0, sizeof rec);
    rec.pid = (uint64_t)getpid(); rec.device = ctx->device;
    // the two staging buffers of the collectives themselves: without them this rank cannot take part at all (the others then
    // meet the base communicator's own time-out); nothing else below returns before the status all-reduce
    if (sg_dev_malloc((void**)&d_rec, sizeof rec) != hipSuccess || sg_dev_malloc((void**)&d_all, sizeof rec * (size_t)W) != hipSuccess) {
        if (d_rec) (void)sg_dev_free(d_rec);
        delete p;
        SG_REQUIRE(false, "SG_COMM_PEER: hipMalloc of the set-up staging buffers failed");
    }
    if (sg_dev_malloc(&p->own, total) != hipSuccess) { p->own = nullptr; (void)hipGetLastError(); fail(-1, "SG_COMM_PEER: hipMalloc of %zu bytes failed", total); }
    else if (hipMemsetAsync(p->own, 0, total, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) fail(-1, "SG_COMM_PEER: clearing the slot buffer failed");
    rec.ptr = (uint64_t)(uintptr_t)p->own;
    rec.pad[0] = rc != 0;      // "this rank has no buffer to offer"
    // (a handle is only needed by ranks in OTHER processes; a failure here surfaces there, when they try to open it)
    if (p->own && hipIpcGetMemHandle(&rec.handle, p->own) != hipSuccess) { (void)hipGetLastError(); memset(&rec.handle, 0, sizeof rec.handle); }
    bool gathered = false;
    if (hipMemcpyAsync(d_rec, &rec, sizeof rec, hipMemcpyHostToDevice, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) fail(-1, "SG_COMM_PEER: upload failed");
    {
        const int g = sg_comm_allgather_f32(ctx, d_rec, d_all, (int64_t)(sizeof rec / 4));
        if (g != 0) fail(g, "SG_COMM_PEER: the all-gather of the buffer records failed: %s", sg_last_error());
        else if (hipMemcpyAsync(all.data(), d_all, sizeof rec * (size_t)W, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) fail(-1, "SG_COMM_PEER: download failed");
        else gathered = true;
    }
    if (gathered) {
        for (int r = 0; r < W; ++r) if (all[(size_t)r].pad[0]) fail(-1, "SG_COMM_PEER: rank %d could not allocate its slot buffer", r);
        // Ranks that are contexts of ONE process share that process's hardware queues (HIP maps streams onto GPU_MAX_HW_QUEUES
        // of them, 4 by default): a collective kernel that waits for the kernel of a rank whose stream sits behind it on the
        // same queue would wait for ever.  (One process per rank -- the deployment -- has a queue set per rank.)  The count says
        // nothing about WHICH queue a stream lands on: in-process ranks are a test arrangement, run with as many queues as ranks.
        int same_pid = 0;
        for (int r = 0; r < W; ++r) same_pid += all[(size_t)r].pid == (uint64_t)getpid();
        const char* hq = getenv("GPU_MAX_HW_QUEUES");
        const int n_hq = hq && atoi(hq) > 0 ? atoi(hq) : 4;
        if (same_pid > n_hq)
            fail(-2, "SG_COMM_PEER: %d ranks are contexts of one process, which has %d hardware queues (GPU_MAX_HW_QUEUES): "
                     "streams sharing a queue cannot wait for each other's kernels; use one process per rank", same_pid, n_hq);
        const char* inj = getenv("SG_COMM_PEER_FAIL_RANK");
        for (int r = 0; r < W && rc == 0; ++r) {
            void* base = nullptr;
            if (r == me) base = p->own;
            else if (inj && atoi(inj) == me) fail(-1, "SG_COMM_PEER: hipIpcOpenMemHandle of rank %d's buffer failed: injected by SG_COMM_PEER_FAIL_RANK", r);
            else if (all[(size_t)r].pid == (uint64_t)getpid()) {      // a context of this process: its pointer is valid here
                base = (void*)(uintptr_t)all[(size_t)r].ptr;
                if (all[(size_t)r].device != ctx->device) {           // ... on another device: peer access (idempotent)
                    const hipError_t e = hipDeviceEnablePeerAccess(all[(size_t)r].device, 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) fail(-1, "SG_COMM_PEER: no peer access from device %d to device %d: %s", ctx->device, all[(size_t)r].device, hipGetErrorString(e));
                    (void)hipGetLastError();
                }
            } else {
                const hipError_t e = hipIpcOpenMemHandle(&base, all[(size_t)r].handle, hipIpcMemLazyEnablePeerAccess);
                if (e != hipSuccess) { fail(-1, "SG_COMM_PEER: hipIpcOpenMemHandle of rank %d's buffer failed: %s", r, hipGetErrorString(e)); (void)hipGetLastError(); }
                else p->mapped[r] = base;
            }
            p->dev.slots[r] = reinterpret_cast<float*>(base);
            p->dev.flags[r] = reinterpret_cast<unsigned*>(reinterpret_cast<uint8_t*>(base) + slot_b);
        }
    }
    // the outcome, made collective: how many ranks failed
    float failed = rc != 0 ? 1.f : 0.f;
    int n_failed = -1;
    if (hipMemcpyAsync(d_rec, &failed, sizeof failed, hipMemcpyHostToDevice, ctx->stream) == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess &&
        sg_comm_allreduce_f32(ctx, d_rec, 1) == 0 &&
        hipMemcpyAsync(&failed, d_rec, sizeof failed, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess)
        n_failed = (int)(failed + 0.5f);
    (void)sg_dev_free(d_rec);
    (void)sg_dev_free(d_all);
    if (rc != 0 || n_failed != 0) {
        peer_release(ctx, p);
        (void)hipGetLastError();
        if (rc != 0) { sg_set_error("%s (%d of %d ranks failed; no rank uses the mesh)", why, n_failed, W); return rc; }
        if (n_failed < 0) { sg_set_error("SG_COMM_PEER: the status all-reduce of the set-up failed; this rank does not use the mesh"); return -1; }
        sg_set_error("SG_COMM_PEER: %d of %d ranks could not set the mesh up (their own messages say why); no rank uses it", n_failed, W);
        return -3;
    }
    p->dev.ctrl = reinterpret_cast<unsigned*>(reinterpret_cast<uint8_t*>(p->own) + slot_b + flag_b);
    p->dev.rank = me; p->dev.world = W;
    ctx->comm->peer = p;
    return 0;
}

// The mesh's sticky time-out word (device address), or NULL without a mesh: published with an update's scalars
// (sg_results_publish) so that a collective that gave up is reported at the next read of the losses.
unsigned* sg_comm_peer_err_word(sg_ctx* ctx) { return (ctx->comm && ctx->comm->peer) ? ctx->comm->peer->dev.ctrl + 2 : nullptr; }
bool sg_comm_peer_on(const sg_ctx* ctx) { return ctx->comm && ctx->comm->peer; }
// The captured update graphs hold k_peer_allreduce nodes with THIS mesh's slot and flag pointers by value: a mesh that was torn
// down and set up again (sg_ctx_comm_set_peer(0) then (1)) must not match the old graph's key.
uint32_t sg_comm_peer_generation(const sg_ctx* ctx) { return (ctx->comm && ctx->comm->peer) ? ctx->comm->peer->generation : 0u; }

// Collective: every rank of the communicator calls it with the same `enable`.  1: build the mesh over the base communicator
// (what SG_COMM_PEER=1 does at sg_ctx_comm_init); 0: drop it, the small all-reduces go back to the base communicator.
extern "C" int sg_ctx_comm_set_peer(sg_ctx* ctx, int enable) {
    SG_REQUIRE(ctx && ctx->comm, "sg_ctx_comm_set_peer: no communicator (sg_ctx_comm_init first)");
    SG_CHECK(hipSetDevice(ctx->device));
    SG_CHECK(hipStreamSynchronize(ctx->stream));
    if (enable) return ctx->comm->peer ? 0 : peer_setup(ctx);
    if (ctx->comm->peer) {
        // every rank's last mesh collective is complete before any rank unmaps: one collective on the base communicator
        float* w = nullptr;
        SG_TRY(sg_ctx_scratch(ctx, 64, &w));
        SG_CHECK(hipMemsetAsync(w, 0, 64, ctx->stream));
        SgPeer* p = ctx->comm->peer;
        ctx->comm->peer = nullptr;
        const int rc = sg_comm_allreduce_f32(ctx, w, 4);
        (void)hipStreamSynchronize(ctx->stream);
        peer_release(ctx, p);
        return rc;
    }
    return 0;
}

static int peer_allreduce(sg_ctx* ctx, float* dev, int64_t n) {
    SgPeer* p = ctx->comm->peer;
    SG_REQUIRE(((uintptr_t)dev & 15) == 0, "peer all-reduce: the vector must be 16-byte aligned");
    const int nb = (int)((n + 1023) / 1024);   // <= SG_PEER_MAX_BLOCKS by the caller's n <= SG_PEER_MAX_FLOATS
    hipLaunchKernelGGL(k_peer_allreduce, dim3((unsigned)nb), dim3(256), 0, ctx->stream, p->dev, dev, (int)n);
    SG_CHECK(hipGetLastError());
    p->launches += 1;
    return 0;
}

static bool peer_wanted() { const char* e = getenv("SG_COMM_PEER"); return e && e[0] == '1'; }

static void comm_common_flags(sg_ctx* ctx, int world) {
    // SG_COMM_ALWAYS=1 keeps the collectives in the launch sequence even for a single rank (where
    // they are the identity): the 1-GPU self-test of the RCCL path (tests/test_gpu_comm.py)
    const char* always = getenv("SG_COMM_ALWAYS");
    ctx->use_comm = world > 1 || (always && always[0] == '1');
    const char* dp = getenv("SG_DISC_DP");
    ctx->disc_sharded = dp && strcmp(dp, "sharded") == 0;
}

extern "C" int sg_ctx_comm_init(sg_ctx* ctx, const uint8_t id[128], int rank, int world) {
    SG_REQUIRE(ctx && id, "sg_ctx_comm_init: NULL argument");
    SG_REQUIRE(world >= 1 && rank >= 0 && rank < world, "sg_ctx_comm_init: bad rank %d / world %d", rank, world);
    SG_REQUIRE(!ctx->comm, "sg_ctx_comm_init: communicator already initialised");
    SG_CHECK(hipSetDevice(ctx->device));
    if (memcmp(id, LB_MAGIC, 8) == 0) {   // loopback transport (sg_comm_loopback_id)
        SgLoopback* lb = new SgLoopback();
        const int rc = lb_open(lb, id, rank, world);
        if (rc != 0) { if (lb->hdr) munmap(lb->hdr, lb->map_bytes); delete lb; return rc; }
        SgComm* c = new SgComm();
        c->lb = lb;
        ctx->comm = c;
        ctx->rank = rank;
        ctx->world = world;
        comm_common_flags(ctx, world);
        return peer_wanted() ? peer_setup(ctx) : 0;
    }
    SG_TRY(rccl_load());
    ncclUniqueId u;
    memcpy(u.internal, id, 128);
    SgComm* c = new SgComm();
    SG_NCCL(g_rccl.CommInitRank(&c->comm, world, u, rank));
    ctx->comm = c;
    ctx->rank = rank;
    ctx->world = world;
    comm_common_flags(ctx, world);
    // One tiny all-reduce now: RCCL sets up its channels and proxy connections lazily on the first collective, which must
    // not happen inside a stream capture (the updates capture their collectives into hipGraphs).
    float* warm = nullptr;
    SG_TRY(sg_ctx_scratch(ctx, 64, &warm));
    SG_CHECK(hipMemsetAsync(warm, 0, 64, ctx->stream));
    SG_NCCL(g_rccl.AllReduce(warm, warm, 4, ncclFloat32, ncclSum, c->comm, ctx->stream));
    SG_CHECK(hipStreamSynchronize(ctx->stream));
    return peer_wanted() ? peer_setup(ctx) : 0;
}

extern "C" int sg_ctx_comm_info(sg_ctx* ctx, int* rank, int* world) {
    SG_REQUIRE(ctx, "sg_ctx_comm_info: ctx is NULL");
    if (rank) *rank = ctx->rank;
    if (world) *world = ctx->world;
    if (ctx->comm && ctx->comm->lb) {   // loopback: the world rank 0 wrote into the segment, and this rank's attach mark
        const LbHeader* h = ctx->comm->lb->hdr;
        SG_REQUIRE((int)h->world == ctx->world && h->attached[ctx->rank].load() == 1,
                   "loopback segment reports a world of %u, the context was initialised as %d of %d", h->world, ctx->rank, ctx->world);
        return 0;
    }
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

// 1: RCCL (collectives are stream operations and can be captured into a hipGraph), 0: loopback (synchronises the stream)
int sg_comm_graph_ok(const sg_ctx* ctx) { return !(ctx->comm && ctx->comm->lb); }

// 0: none, 1: RCCL, 2: loopback; + 4: the small float32 all-reduces run over the peer mesh (SG_COMM_PEER=1)
extern "C" int sg_ctx_comm_kind(sg_ctx* ctx, int* kind) {
    SG_REQUIRE(ctx && kind, "sg_ctx_comm_kind: NULL argument");
    *kind = !ctx->comm ? 0 : (ctx->comm->lb ? 2 : 1) | (ctx->comm->peer ? 4 : 0);
    return 0;
}

void sg_comm_destroy(sg_ctx* ctx) {
    if (!ctx->comm) return;
    // (the stream has been synchronised: this rank's last collective is complete, so no rank writes into its buffer any more
    // -- a collective completes only when every rank's chunk has arrived, and a rank cannot start another without this one)
    if (ctx->comm->peer) { peer_release(ctx, ctx->comm->peer); ctx->comm->peer = nullptr; }
    if (ctx->comm->lb) {
        SgLoopback* lb = ctx->comm->lb;
        if (lb->h_tmp) (void)sg_host_release(lb->h_tmp);
        if (lb->hdr) munmap(lb->hdr, lb->map_bytes);
        delete lb;
    }
    // (an RCCL communicator is left to process exit: ncclCommDestroy blocks when a peer rank has already gone)
    delete ctx->comm;
    ctx->comm = nullptr;
}

static int allreduce_f32(sg_ctx* ctx, float* dev, int64_t n) {
    if (ctx->comm->peer && n > 0 && n <= SG_PEER_MAX_FLOATS) return peer_allreduce(ctx, dev, n);
    if (ctx->comm->lb) return lb_collective(ctx, 0, dev, dev, (size_t)n, 4);
    SG_NCCL(g_rccl.AllReduce(dev, dev, (size_t)n, ncclFloat32, ncclSum, ctx->comm->comm, ctx->stream));
    return 0;
}

// The per-step gradient all-reduce.  With profiling on (sg_ctx_profile: bench.py's separate HIP-event pass) the collective is
// bracketed by two events on the stream -- what the device spends between reaching it and leaving it, the wait for the
// slowest peer included -- and accumulated in slot SG_PROF_COMM_F32.
int sg_comm_allreduce_f32(sg_ctx* ctx, float* dev, int64_t n) {
    SG_REQUIRE(ctx->comm, "all-reduce requested but no communicator (call sg_ctx_comm_init)");
    if (!ctx->profile) return allreduce_f32(ctx, dev, n);
    const SgEv ev = sg_prof_events(ctx, SG_PROF_COMM_F32);
    SG_CHECK(hipEventRecord(ev.a, ctx->stream));
    const int rc = allreduce_f32(ctx, dev, n);
    SG_CHECK(hipEventRecord(ev.b, ctx->stream));
    return rc;
}

int sg_comm_allreduce_f64(sg_ctx* ctx, double* dev, int64_t n) {
    SG_REQUIRE(ctx->comm, "all-reduce requested but no communicator (call sg_ctx_comm_init)");
    if (ctx->comm->lb) return lb_collective(ctx, 0, dev, dev, (size_t)n, 8);
    SG_NCCL(g_rccl.AllReduce(dev, dev, (size_t)n, ncclFloat64, ncclSum, ctx->comm->comm, ctx->stream));
    return 0;
}

int sg_comm_allgather_f32(sg_ctx* ctx, const float* dev_in, float* dev_out, int64_t n_per_rank) {
    SG_REQUIRE(ctx->comm, "all-gather requested but no communicator (call sg_ctx_comm_init)");
    if (ctx->comm->lb) return lb_collective(ctx, 1, dev_in, dev_out, (size_t)n_per_rank, 4);
    SG_NCCL(g_rccl.AllGather(dev_in, dev_out, (size_t)n_per_rank, ncclFloat32, ctx->comm->comm, ctx->stream));
    return 0;
}
