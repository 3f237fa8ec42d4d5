// sg_ctx.cpp -- context, error reporting, scratch arenas, HIP-event profiling slots.
#include <errno.h>
#include <atomic>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include "sg_common.h"

static thread_local char g_err[1024] = "";

void sg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" const char* sg_last_error(void) { return g_err; }
extern "C" const char* sg_version(void) { return "simgan_hip 0.1 (gfx950)"; }

// contexts of this process, per device, that own at least one learner object (sg_ppo / sg_disc): the objects whose updates may
// pick a launch that waits inside itself
static std::atomic<int> g_learner_ctx[64];

// ... and OTHER processes: while a process owns learner objects on a device it holds a shared (read) record lock on
// /dev/shm/sg_gpu_<pci bus id>.lock (SG_LOCK_DIR overrides the directory); sg_ctx_exclusive asks the kernel whether anybody else
// holds one (F_GETLK for a write lock: a process's own locks never conflict with it, the query changes nothing, and a process
// that dies releases its lock with its descriptors).  POSIX record locks rather than flock(): converting a flock from shared
// to exclusive and back drops it for a moment, and two processes probing at once can each find the other gone.  What it cannot
// see: processes in another container (their own /dev/shm) -- SG_DISC_FUSED=0 SG_PPO_PAIR=0 remain for that.
static std::mutex g_lock_m;
static int g_lock_fd[64];
static bool g_lock_failed[64];   // the lock could not be taken: exclusivity cannot be established (sg_ctx_exclusive: false)
static bool g_lock_init = false;

static void device_lock_take(int device) {
    std::lock_guard<std::mutex> l(g_lock_m);
    if (!g_lock_init) { for (int& f : g_lock_fd) f = -1; g_lock_init = true; }
    if (g_lock_fd[device & 63] >= 0) return;
    char bus[64] = "unknown";
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) { (void)hipGetLastError(); snprintf(bus, sizeof bus, "dev%d", device); }
    for (char* c = bus; *c; ++c) if (*c == ':' || *c == '/' || *c == '.') *c = '_';
    const char* dir = getenv("SG_LOCK_DIR");
    char path[256];
    snprintf(path, sizeof path, "%s/sg_gpu_%s.lock", dir && *dir ? dir : "/dev/shm", bus);
    // (no umask() games: the mask is process-wide and another thread may be creating files; the mode is widened on the descriptor,
    // which only the file's creator can and need do)
    const int fd = open(path, O_CREAT | O_RDWR | O_CLOEXEC, 0666);
    if (fd >= 0) (void)fchmod(fd, 0666);
    struct flock fl;
    memset(&fl, 0, sizeof fl);
    fl.l_type = F_RDLCK; fl.l_whence = SEEK_SET;
    if (fd < 0 || fcntl(fd, F_SETLK, &fl) != 0) {
        // Without the lock other processes' learners cannot be seen: this process then must not claim the device for itself --
        // sg_ctx_exclusive answers false and the updates take their multi-launch forms (bit-identical, a few percent slower).
        static std::atomic<bool> told{false};
        if (!told.exchange(true))
            fprintf(stderr, "[simgan_hip] cannot take the per-device lock %s (%s): other processes on this GPU cannot be detected, so the "
                            "one-launch update forms are off (SG_LOCK_DIR names another directory; SG_LOCK_OPTIONAL=1 keeps them on)\n",
                    path, strerror(errno));
        if (fd >= 0) close(fd);
        const char* opt = getenv("SG_LOCK_OPTIONAL");
        g_lock_failed[device & 63] = !(opt && opt[0] == '1');
        return;
    }
    g_lock_failed[device & 63] = false;
    g_lock_fd[device & 63] = fd;
}

static void device_lock_drop(int device) {
    std::lock_guard<std::mutex> l(g_lock_m);
    if (!g_lock_init || g_lock_fd[device & 63] < 0) return;
    close(g_lock_fd[device & 63]);     // releases the record lock
    g_lock_fd[device & 63] = -1;
}

static bool device_other_process(int device) {
    std::lock_guard<std::mutex> l(g_lock_m);
    if (g_lock_init && g_lock_failed[device & 63]) return true;    // unknown: assume somebody is there
    if (!g_lock_init || g_lock_fd[device & 63] < 0) return false;
    struct flock fl;
    memset(&fl, 0, sizeof fl);
    fl.l_type = F_WRLCK; fl.l_whence = SEEK_SET;
    if (fcntl(g_lock_fd[device & 63], F_GETLK, &fl) != 0) return false;
    return fl.l_type != F_UNLCK;
}

void sg_ctx_learner_born(sg_ctx* ctx) {
    if (ctx->n_learners.fetch_add(1, std::memory_order_relaxed) == 0 && g_learner_ctx[ctx->device & 63].fetch_add(1, std::memory_order_relaxed) == 0)
        device_lock_take(ctx->device);
}
void sg_ctx_learner_gone(sg_ctx* ctx) {
    if (ctx->n_learners.fetch_sub(1, std::memory_order_relaxed) == 1 && g_learner_ctx[ctx->device & 63].fetch_sub(1, std::memory_order_relaxed) == 1)
        device_lock_drop(ctx->device);
}
bool sg_ctx_exclusive(const sg_ctx* ctx) {
    return g_learner_ctx[ctx->device & 63].load(std::memory_order_relaxed) <= 1 && (ctx->world <= 1 || sg_comm_graph_ok(ctx)) &&
           !device_other_process(ctx->device);
}

uint64_t sg_next_feat_version() {
    static std::atomic<uint64_t> g_next{1};
    return g_next.fetch_add(1, std::memory_order_relaxed);
}

extern "C" int sg_ctx_create(int device, sg_ctx** out) {
    SG_DEVICE_WIDE();
    SG_REQUIRE(out != nullptr, "sg_ctx_create: out is NULL");
    // Kernel arguments in device memory instead of host-coherent memory: every kernel of the
    // 5,000-launch update chain starts by reading its kernarg segment, and that first scalar load
    // is a PCIe round trip otherwise.  Must be in the environment before the HIP runtime
    // initialises; an explicit user setting wins.
    setenv("HIP_FORCE_DEV_KERNARG", "1", 0);
    // Small copies on shader blits rather than the SDMA engines: an update ends each phase with a few
    // tiny read-backs, and on this platform SDMA scheduling intermittently adds milliseconds to them.
    setenv("HSA_ENABLE_SDMA", "0", 0);
    int count = 0;
    SG_CHECK(hipGetDeviceCount(&count));
    SG_REQUIRE(device >= 0 && device < count, "sg_ctx_create: device %d out of range (%d visible)", device, count);
    SG_CHECK(hipSetDevice(device));
    sg_ctx* ctx = new sg_ctx();
    ctx->device = device;
    hipDeviceProp_t prop;
    SG_CHECK(hipGetDeviceProperties(&prop, device));
    ctx->num_cu = prop.multiProcessorCount;
    ctx->lds_bytes = (int)prop.sharedMemPerBlock;
    SG_REQUIRE(strstr(prop.gcnArchName, "gfx950") != nullptr,
               "sg_ctx_create: device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    SG_CHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    SG_CHECK(sg_host_malloc((void**)&ctx->mailbox, sizeof(double) * 128));
    SG_CHECK(sg_host_malloc((void**)&ctx->results, sizeof(double) * 16 * SG_RESULT_SLOTS));
    for (int i = 0; i < SG_RESULT_SLOTS; ++i) SG_CHECK(hipEventCreateWithFlags(&ctx->res_ev[i], hipEventDisableTiming));
    *out = ctx;
    return 0;
}

extern "C" int sg_ctx_destroy(sg_ctx* ctx) {
    SG_DEVICE_WIDE();
    if (!ctx) return 0;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    sg_comm_destroy(ctx);
    for (auto& s : ctx->prof)
        for (auto& p : s.pending) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    for (auto e : ctx->marks) (void)hipEventDestroy(e);
    if (ctx->d_scratch) (void)sg_dev_free(ctx->d_scratch);
    if (ctx->h_pinned) (void)sg_host_release(ctx->h_pinned);
    if (ctx->mailbox) (void)sg_host_release(ctx->mailbox);
    if (ctx->results) (void)sg_host_release(ctx->results);
    for (auto e : ctx->res_ev) if (e) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return 0;
}

__global__ void k_copy_f64(double* dst, const double* src, int n) {
    if ((int)threadIdx.x < n) dst[threadIdx.x] = src[threadIdx.x];
}

int sg_ctx_fetch_f64(sg_ctx* ctx, const double* dev, double* host, int n) {
    SG_REQUIRE(n >= 0 && n <= 64, "sg_ctx_fetch_f64: at most 64 values");
    hipLaunchKernelGGL(k_copy_f64, dim3(1), dim3(64), 0, ctx->stream, ctx->mailbox, dev, n);
    SG_CHECK(hipGetLastError());
    SG_CHECK(hipStreamSynchronize(ctx->stream));
    memcpy(host, ctx->mailbox, sizeof(double) * n);
    return 0;
}

int sg_ctx_put_f64(sg_ctx* ctx, double* dev, const double* host, int n) {
    SG_REQUIRE(n >= 0 && n <= 64, "sg_ctx_put_f64: at most 64 values");
    SG_CHECK(hipStreamSynchronize(ctx->stream));   // a previous put may still be reading the slot
    memcpy(ctx->mailbox + 64, host, sizeof(double) * n);
    hipLaunchKernelGGL(k_copy_f64, dim3(1), dim3(64), 0, ctx->stream, dev, ctx->mailbox + 64, n);
    SG_CHECK(hipGetLastError());
    return 0;
}

extern "C" int sg_ctx_synchronize(sg_ctx* ctx) {
    SG_REQUIRE(ctx, "sg_ctx_synchronize: ctx is NULL");
    SG_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int sg_ctx_device_info(sg_ctx* ctx, char* name, int name_len, int* num_cu, int64_t* hbm_bytes) {
    SG_REQUIRE(ctx, "sg_ctx_device_info: ctx is NULL");
    hipDeviceProp_t prop;
    SG_CHECK(hipGetDeviceProperties(&prop, ctx->device));
    if (name && name_len > 0) snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
    if (num_cu) *num_cu = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return 0;
}

int sg_ctx_scratch(sg_ctx* ctx, size_t bytes, float** out) {
    if (bytes > ctx->scratch_bytes) {
        SG_CHECK(hipStreamSynchronize(ctx->stream));
        if (ctx->d_scratch) SG_CHECK(sg_dev_free(ctx->d_scratch));
        size_t cap = bytes + bytes / 2 + 4096;
        SG_CHECK(sg_dev_malloc((void**)&ctx->d_scratch, cap));
        ctx->scratch_bytes = cap;
    }
    *out = ctx->d_scratch;
    return 0;
}

int sg_ctx_pinned(sg_ctx* ctx, size_t bytes, void** out) {
    if (bytes > ctx->pinned_bytes) {
        SG_CHECK(hipStreamSynchronize(ctx->stream));
        if (ctx->h_pinned) SG_CHECK(sg_host_release(ctx->h_pinned));
        size_t cap = bytes + bytes / 2 + 4096;
        SG_CHECK(sg_host_malloc(&ctx->h_pinned, cap));
        ctx->pinned_bytes = cap;
    }
    *out = ctx->h_pinned;
    return 0;
}

// ------------------------------------------------------------------------------- profiling
static hipEvent_t take_event(sg_ctx* ctx) {
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}

SgEv sg_prof_events(sg_ctx* ctx, int which) {
    SgEv ev;
    if (!ctx->profile || which < 0) return ev;
    ev.a = take_event(ctx);
    ev.b = take_event(ctx);
    ctx->prof[which].pending.emplace_back(ev.a, ev.b);
    return ev;
}

static void prof_drain(sg_ctx* ctx) {
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& s : ctx->prof) {
        for (auto& p : s.pending) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
                s.total_ms += ms;
                s.launches += 1;
            }
            ctx->event_pool.push_back(p.first);
            ctx->event_pool.push_back(p.second);
        }
        s.pending.clear();
    }
}

extern "C" int sg_ctx_profile(sg_ctx* ctx, int enable) {
    SG_REQUIRE(ctx, "sg_ctx_profile: ctx is NULL");
    prof_drain(ctx);
    ctx->profile = enable != 0;
    return 0;
}

extern "C" int sg_ctx_profile_read(sg_ctx* ctx, int which, double* total_ms, int64_t* launches) {
    SG_REQUIRE(ctx, "sg_ctx_profile_read: ctx is NULL");
    SG_REQUIRE(which >= 0 && which < SG_PROF_COUNT, "sg_ctx_profile_read: bad slot %d", which);
    prof_drain(ctx);
    if (total_ms) *total_ms = ctx->prof[which].total_ms;
    if (launches) *launches = ctx->prof[which].launches;
    return 0;
}

extern "C" int sg_ctx_profile_reset(sg_ctx* ctx) {
    SG_REQUIRE(ctx, "sg_ctx_profile_reset: ctx is NULL");
    prof_drain(ctx);
    for (auto& s : ctx->prof) { s.total_ms = 0.0; s.launches = 0; }
    return 0;
}

// Marks: one HIP event recorded on the library's stream per call (between two updates: a timestamp the device takes when it
// gets there, the host does not wait); the time between two marks is read after the loop.  id < 0 forgets every mark.
extern "C" int sg_ctx_mark(sg_ctx* ctx, int* id) {
    SG_REQUIRE(ctx, "sg_ctx_mark: ctx is NULL");
    SG_CHECK(hipSetDevice(ctx->device));
    if (!id) {
        SG_CHECK(hipStreamSynchronize(ctx->stream));
        for (auto e : ctx->marks) (void)hipEventDestroy(e);
        ctx->marks.clear();
        return 0;
    }
    SG_REQUIRE(ctx->marks.size() < 65536, "sg_ctx_mark: 65536 marks outstanding (sg_ctx_mark(ctx, NULL) forgets them)");
    hipEvent_t e;
    SG_CHECK(hipEventCreate(&e));
    SG_CHECK(hipEventRecord(e, ctx->stream));
    ctx->marks.push_back(e);
    *id = (int)ctx->marks.size() - 1;
    return 0;
}

extern "C" int sg_ctx_mark_elapsed(sg_ctx* ctx, int from, int to, double* ms) {
    SG_REQUIRE(ctx && ms, "sg_ctx_mark_elapsed: NULL argument");
    const int n = (int)ctx->marks.size();
    SG_REQUIRE(from >= 0 && from < n && to >= 0 && to < n, "sg_ctx_mark_elapsed: marks %d, %d of %d", from, to, n);
    SG_CHECK(hipEventSynchronize(ctx->marks[to]));
    float f = 0.f;
    SG_CHECK(hipEventElapsedTime(&f, ctx->marks[from], ctx->marks[to]));
    *ms = (double)f;
    return 0;
}

// Page-locked host memory for the caller's side of the boundary: a rollout whose host tensors live in it crosses PCIe at the
// link's speed (one DMA per field) instead of through the runtime's pageable staging path (measured 1.8 GB/s on the
// north-star rollout's 40 MB).  Plain hipHostMalloc / hipHostFree; needs no context.
extern "C" int sg_host_alloc(int64_t bytes, void** out) {
    SG_REQUIRE(out && bytes > 0, "sg_host_alloc: bad argument");
    SG_CHECK(sg_host_malloc(out, (size_t)bytes));
    return 0;
}

extern "C" int sg_host_free(void* p) {
    if (p) SG_CHECK(sg_host_release(p));
    return 0;
}
