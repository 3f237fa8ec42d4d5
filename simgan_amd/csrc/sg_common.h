// sg_common.h -- host-side structures shared by the translation units of libsimgan_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <atomic>
#include <vector>
#include <mutex>
// hipMalloc / hipFree (and the pinned-host pair) are device-wide operations: issued by one host thread while ANOTHER thread's
// stream capture is in flight they invalidate that capture on this runtime, even in the relaxed capture mode ("operation failed
// due to a previous error during capture"; round 4's tools/pair_stress.py: 2 of 160 rounds).  Every allocation and release
// of the library therefore goes through these wrappers, which take the mutex the captures are serialised on (recursive: a
// capture's own thread may grow a scratch arena before it begins to record).
inline std::recursive_mutex& sg_capture_mutex() { static std::recursive_mutex m; return m; }
// Object creation and destruction as a whole (allocations, the stream synchronisation and graph-exec release of a destroy
// call, whose errors are deliberately ignored): never beside a capture in flight.  The guard also drops whatever error a
// best-effort call of the scope left in the thread's HIP error slot, so that it cannot surface in a later, unrelated call.
struct SgDeviceWideGuard {
    std::lock_guard<std::recursive_mutex> l{sg_capture_mutex()};
    ~SgDeviceWideGuard() { (void)hipGetLastError(); }
};
#define SG_DEVICE_WIDE() SgDeviceWideGuard _sg_device_wide_guard
static inline hipError_t sg_dev_malloc(void** p, size_t n) { std::lock_guard<std::recursive_mutex> l(sg_capture_mutex()); return hipMalloc(p, n); }
static inline hipError_t sg_dev_free(void* p) { std::lock_guard<std::recursive_mutex> l(sg_capture_mutex()); return hipFree(p); }
static inline hipError_t sg_host_malloc(void** p, size_t n) { std::lock_guard<std::recursive_mutex> l(sg_capture_mutex()); return hipHostMalloc(p, n, hipHostMallocDefault); }
static inline hipError_t sg_host_release(void* p) { std::lock_guard<std::recursive_mutex> l(sg_capture_mutex()); return hipHostFree(p); }

#include "../../include/simgan_hip.h"
#include "sg_gemm.hpp"

void sg_set_error(const char* fmt, ...);

#define SG_CHECK(expr)                                                                      \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            sg_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr,                \
                         hipGetErrorString(_e));                                            \
            return -1;                                                                      \
        }                                                                                   \
    } while (0)

#define SG_REQUIRE(cond, ...)        \
    do {                             \
        if (!(cond)) {               \
            sg_set_error(__VA_ARGS__); \
            return -2;               \
        }                            \
    } while (0)

// Synchronous host<->device copy on the context's OWN stream.  The runtime's synchronous hipMemcpy goes through the legacy
// stream, which the runtime refuses (and which invalidates the capture) while any stream of the process is capturing --
// contexts driven from several host threads would break each other's graph captures.
#define SG_COPY_SYNC(ctx, dst, src, bytes, kind)                                         \
    do {                                                                                 \
        SG_CHECK(hipMemcpyAsync(dst, src, bytes, kind, (ctx)->stream));                  \
        SG_CHECK(hipStreamSynchronize((ctx)->stream));                                   \
    } while (0)

#define SG_TRY(expr)            \
    do {                        \
        int _r = (expr);        \
        if (_r != 0) return _r; \
    } while (0)

// ----------------------------------------------------------------------------- padded layouts
// Device parameter vectors are stored "tile padded": a weight [n_out, n_in] occupies
// [pad16(n_out)][pad16(n_in)+4] floats (zero in the padding) so that a trunk's parameter block is
// byte-for-byte the LDS image the kernels compute on.  Adam moments and gradient slabs use the
// same layout; padding entries have zero gradient forever and therefore stay zero.

struct SgTrunk {       // one 2-hidden-layer tanh trunk + its stacked linear heads
    int off;           // start of the block in the padded vector (floats)
    int w1, b1, w2, b2, wh, bh, ex;  // offsets relative to `off`
    int size;          // block length (floats, multiple of 4)
    int P, Pp, ldP;    // head outputs (real / padded / LDS stride of the [R][Pp] output tile)
    int EX;            // extra per-trunk vector (MLP actor: logstd[A]); 0 if none
    int H, Hp, ldH;    // this trunk's hidden width (real / padded / LDS stride): the critic's may differ from the actors'
};

struct SgPolicyDesc {
    int kind, O, A, H, num_feet;
    int Hc;               // hidden width of the CRITIC trunk (== H unless Policy.reset_critic rebuilt it: a2c/model.py:80-87)
    int Op, ldO, Hp, ldH; // Hp / ldH: the WIDEST trunk's (sizes LDS tiles and the row stacks; a trunk computes with its own)
    int n_trunks;         // 2 (MLP: actor, critic) or 3 (split: contact, actuator, critic)
    SgTrunk trunk[3];
    int total;            // padded parameter count
    int nc, na;           // split: 4*feet, 3*feet
};

struct SgDiscDesc {
    int F, Hd, Fp, ldF, Hp, ldH;
    int w1, b1, w2, b2, w3, b3;  // offsets in the padded vector
    int total;
};

SgPolicyDesc sg_make_policy_desc(int kind, int O, int A, int H, int num_feet, int Hc = 0);   // Hc 0: critic as wide as the actors
int64_t sg_policy_flat_count(const SgPolicyDesc& d);
void sg_policy_pad(const SgPolicyDesc& d, const float* flat, float* padded);    // padded must be zeroed
void sg_policy_unpad(const SgPolicyDesc& d, const float* padded, float* flat);
struct sg_ctx;
// true: the policy's trunks do not fit a CU's LDS beside a row tile (or SG_POLICY_GW=1): the forward / PPO kernels run
// their global-weight instances (sg_policy.hip)
bool sg_policy_needs_gw(const sg_ctx* ctx, const SgPolicyDesc& d);
SgDiscDesc sg_make_disc_desc(int F, int Hd);
int64_t sg_disc_flat_count(const SgDiscDesc& d);
void sg_disc_pad(const SgDiscDesc& d, const float* flat, float* padded);
void sg_disc_unpad(const SgDiscDesc& d, const float* padded, float* flat);

// ------------------------------------------------------------------------------------ handles
enum { SG_PROF_DISC_CHAIN = 0, SG_PROF_DISC_WGRAD, SG_PROF_PPO_FWD, SG_PROF_PPO_BWD,
       SG_PROF_PPO_REDUCE, SG_PROF_RELABEL, SG_PROF_PPO_ADAM, SG_PROF_DISC_STEP, SG_PROF_COMM_F32, SG_PROF_COUNT };

struct SgProfSlot {
    double total_ms = 0.0;
    int64_t launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

struct SgComm;  // RCCL communicator (sg_comm.cpp)

struct sg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int num_cu = 0;
    int lds_bytes = 0;
    bool profile = false;
    SgProfSlot prof[SG_PROF_COUNT];
    std::vector<hipEvent_t> event_pool;
    std::vector<hipEvent_t> marks;   // sg_ctx_mark: timestamps on the stream between updates (bench.py's per-update spread)
    SgComm* comm = nullptr;
    int rank = 0, world = 1;
    bool disc_sharded = false;    // SG_DISC_DP=sharded: all-reduce D gradients per step instead of replicating D
    bool use_comm = false;        // issue the data-parallel collectives (world > 1, or forced for a 1-rank self-test)
    std::atomic<int> n_learners{0};   // live sg_ppo / sg_disc objects of this context (sg_ctx_exclusive)
    // scratch
    float* d_scratch = nullptr;   // generic device scratch (host<->device staging for API calls)
    size_t scratch_bytes = 0;
    void* h_pinned = nullptr;     // pinned host staging
    size_t pinned_bytes = 0;
    double* mailbox = nullptr;    // 2 x 64 doubles of device-visible pinned host memory: small results / small inputs
    // results ring: SG_RESULT_SLOTS x 16 doubles of device-visible pinned host memory.  An update that runs without a host
    // synchronisation publishes its scalars (losses, r_sa, ret_rms) into a slot with a one-wave kernel and records the slot's
    // event; the host reads the slot when it wants the numbers (sg_results_publish / sg_results_fetch).
    double* results = nullptr;
    hipEvent_t res_ev[8] = {nullptr};
    // the objects whose scalars a slot holds (cleared by their destroy calls): sg_results_fetch reports a hand-off time-out of
    // their self-waiting launches (slot words [13], [14]) against them
    struct sg_disc* res_d[8] = {nullptr};
    struct sg_ppo* res_a[8] = {nullptr};
};
#define SG_RESULT_SLOTS 8

// The launches that wait inside themselves (k_disc_step4, k_ppo_pair) assume that ALL their workgroups become resident: the
// chip dispatches a grid's workgroups XCD by XCD, each XCD in index order on its own, so when another grid competes for the
// CUs a waiting workgroup can hold the slot that the workgroup it waits for needs on another XCD (two such grids can block
// each other until their spins time out; seen with 8 processes sharing one GPU).  With the device to itself -- one process per
// GPU, the deployment this library is built for -- every workgroup of those grids is resident at once and the hazard does not
// exist.  false: more than one context of this process owns learner objects (sg_ppo / sg_disc) on the device, or the
// context's communicator is the loopback transport (ranks of one host, typically sharing a device): the callers then use the
// multi-launch forms; so do they while ANOTHER process of this host holds learner objects on the device (a per-device advisory
// lock under /dev/shm, sg_ctx.cpp; processes in other containers cannot be seen: SG_DISC_FUSED=0 / SG_PPO_PAIR=0).
bool sg_ctx_exclusive(const sg_ctx* ctx);
void sg_ctx_learner_born(sg_ctx* ctx);   // sg_ppo_create / sg_disc_create
void sg_ctx_learner_gone(sg_ctx* ctx);   // ... and their destroy calls
int sg_ctx_scratch(sg_ctx* ctx, size_t bytes, float** out);
int sg_ctx_pinned(sg_ctx* ctx, size_t bytes, void** out);
// n <= 64 doubles device -> host / host -> device through the mailbox: a kernel reads or writes pinned host
// memory directly, so the 24-byte loss read-backs at the end of every update phase do not go through the
// runtime's DMA copy path (which costs tens of microseconds and occasionally stalls for milliseconds).
// fetch synchronises the stream; put is asynchronous but ordered (it synchronises before reusing the slot).
int sg_ctx_fetch_f64(sg_ctx* ctx, const double* dev, double* host, int n);
int sg_ctx_put_f64(sg_ctx* ctx, double* dev, const double* host, int n);
// Profiling: when ctx->profile is on, a launch is given a start/stop event pair through
// hipExtLaunchKernelGGL, which timestamps the kernel's own begin and end on the device (the same
// quantity rocprofv3 --kernel-trace reports), not the gaps around it.  Off: null events.
struct SgEv { hipEvent_t a = nullptr, b = nullptr; };
SgEv sg_prof_events(sg_ctx* ctx, int which);
#ifdef __HIPCC__
#include <hip/hip_ext.h>
#define SG_LAUNCH(ctx, which, kernel, grid, block, lds, ...)                                                  \
    do {                                                                                                      \
        if ((ctx)->profile) {                                                                                 \
            SgEv _ev = sg_prof_events(ctx, which);                                                            \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, (ctx)->stream, _ev.a, _ev.b, 0, __VA_ARGS__);     \
        } else {                                                                                              \
            hipLaunchKernelGGL(kernel, grid, block, lds, (ctx)->stream, __VA_ARGS__);                         \
        }                                                                                                     \
    } while (0)
#endif

struct sg_policy {
    sg_ctx* ctx;
    SgPolicyDesc desc;
    float* d_params = nullptr;   // padded
    float* d_io = nullptr;       // staging for the host-pointer entry points
    size_t io_bytes = 0;
};

// Layout of sg_disc::d_state behind the SgOptState at its head, in unsigned words: k_disc_step4's hand-off flags (one
// 128-byte line per chain workgroup) and its sticky time-out word; sg_ppo::d_pair: k_ppo_pair's time-out word.
#define SG_STEP4_FLAG_WORD0 64
#define SG_STEP4_MAX_FLAGS 1024
#ifndef SG_STEP4_FLAG_STRIDE
#define SG_STEP4_FLAG_STRIDE 32           // words between two workgroups' flags: one 128-byte line each
#endif
#define SG_STEP4_ERR_WORD (SG_STEP4_FLAG_WORD0 + SG_STEP4_MAX_FLAGS * SG_STEP4_FLAG_STRIDE)
#define SG_STEP4_STATE_BYTES (4 * (SG_STEP4_ERR_WORD + 16))
#define SG_PAIR_ERR_WORD 0
unsigned* sg_comm_peer_err_word(struct sg_ctx* ctx);   // sg_comm.cpp: the peer mesh's sticky time-out word, or NULL
bool sg_comm_peer_on(const struct sg_ctx* ctx);
uint32_t sg_comm_peer_generation(const struct sg_ctx* ctx);   // 0: no mesh; else unique to the mesh instance (graph keys)
unsigned* sg_disc_err_word(struct sg_disc* d);   // sg_disc.hip: k_disc_step4's sticky time-out word (device address)
uint64_t sg_next_feat_version();   // sg_ctx.cpp: process-wide, monotonic, never 0
struct sg_rollout {
    sg_ctx* ctx;
    int T, N, O, A, F;
    float* d_field[SG_F_COUNT] = {nullptr};
    int64_t field_count[SG_F_COUNT] = {0};
    int field_slots[SG_F_COUNT] = {0};   // T or T+1
    int field_width[SG_F_COUNT] = {0};
    int64_t* d_perm = nullptr;           // [T*N] scratch permutation
    // bumped by every entry point that writes obs_feat on the device (upload, upload_step, after_update, fill_synthetic): lets
    // the discriminator's replicated data-parallel mode see that the rows it all-gathered for the previous epoch are still current
    // content stamp of obs_feat: drawn from ONE process-wide counter (sg_next_feat_version) at creation and at every change, so a
    // stamp is never shared by two rollouts -- a destroyed rollout's successor at the same address with the same upload history
    // cannot be mistaken for it by a cache keyed on (pointer, rows, stamp) (sg_disc.hip: the replicated-mode all-gather cache)
    uint64_t feat_version = sg_next_feat_version();
};

struct sg_ppo {
    int64_t opt_t = 0;           // completed Adam steps (mirrors SgOptState::t0 on the device)
    uint64_t scratch_key = 0;    // layout the scratch buffers were last cleared for
    hipGraphExec_t steps_graph = nullptr;   // the update's optimizer steps, captured once and replayed
    bool graph_refused = false;             // a capture with collectives failed once: stay on direct launches
    uint64_t steps_graph_key[16] = {0};
    sg_ctx* ctx;
    sg_policy* policy;
    sg_ppo_config cfg;
    float *d_m = nullptr, *d_v = nullptr, *d_grad = nullptr;
    float* d_slabs = nullptr;      // [row groups][total+8] partial bias-type gradients and loss sums
    size_t slabs_cap = 0;
    float* d_stacks = nullptr;     // row stacks of one minibatch (activations / their gradients)
    size_t stacks_cap = 0;
    float* d_state = nullptr;      // device scalars: see SgOptState
    int64_t* d_perms = nullptr;    // [ppo_epoch][T*N]
    int64_t perms_cap = 0;
    int64_t last_perm_count = 0;   // entries of d_perms the last update consumed (sg_ppo_last_perms)
    double* d_loss_acc = nullptr;  // [3] running loss sums over the update
    float* d_part = nullptr;       // per-block partial sums (sumsq, losses)
    long long* d_dbg = nullptr;    // phase-timestamp buffer (test hook)
    unsigned* d_pair = nullptr;    // k_ppo_pair: error word
    bool self_wait_failed = false; // a k_ppo_pair hand-off timed out on this object: its later updates run the two-launch step
    bool pair_primed = false;      // the row stacks were cleared for k_ppo_pair's tagged words and no other mode has run since
};

struct sg_disc {
    sg_ctx* ctx;
    SgDiscDesc desc;
    bool self_wait_failed = false; // a k_disc_step4 hand-off timed out on this object: its later steps run as two launches
    float *d_params = nullptr, *d_m = nullptr, *d_v = nullptr;
    float* d_slabs = nullptr;
    int n_slabs = 0;
    float* d_state = nullptr;
    float* d_expert = nullptr;
    int64_t n_expert = 0;
    int64_t *d_eperm = nullptr, *d_pperm = nullptr;
    int64_t eperm_cap = 0, pperm_cap = 0;
    float* d_alpha = nullptr;
    int64_t alpha_cap = 0;
    int64_t last_draws[3] = {0, 0, 0};   // entries of d_eperm / d_pperm / d_alpha the last epoch consumed
    float* d_feat_all = nullptr;   // replicated data-parallel mode: all ranks' next_obs_feat rows
    int64_t feat_all_cap = 0;
    const float* gather_src = nullptr;   // what d_feat_all currently holds: rows pointer, row count and version of the rollout
    int64_t gather_rows = 0;             // it was gathered from (0 = nothing reusable)
    uint64_t gather_version = 0;
    int64_t n_gathers = 0;               // all-gathers issued so far (test hook)
    int64_t opt_t = 0;             // completed Adam steps (mirrors SgOptState::t0 on the device)
    hipGraphExec_t epoch_graph = nullptr;   // one epoch of update steps, captured once and replayed
    bool graph_refused = false;             // a capture with collectives failed once: stay on direct launches
    uint64_t epoch_graph_key[12] = {0};
    float* d_wT = nullptr;         // weight images W1 | W2 | W2^T | W1^T of k_disc_chain4, maintained by k_disc_wgrad
    float *d_erows = nullptr, *d_prows = nullptr;   // the epoch's expert / policy rows in consumption order
    int64_t erows_cap = 0, prows_cap = 0;
    float* d_rows = nullptr;       // sg_disc_update_rows: caller-assembled policy rows
    int64_t rows_cap = 0;
    double* d_loss_acc = nullptr;
    double* d_scal = nullptr;      // device-resident learner scalars: ret_rms {mean, var, count} | sum(1 - masks) | r_sa
    int last_n_d = 0;              // steps of the last epoch (its loss sums in d_loss_acc are divided by this)
    float* d_returns = nullptr;    // Discriminator.returns [n]
    int returns_n = 0;
    bool returns_none = true;
    uint64_t rng_calls = 0;
    long long* d_dbg_step4 = nullptr;   // k_disc_step4's stamps (SG_STEP4_STAMPS builds, tools/step4_times.py)
    long long* d_dbg = nullptr;    // phase-timestamp buffer (test hook)
};

// Device-side optimizer scalars: kept in device memory so queued / graph-replayed launches never depend on
// host-side kernel arguments that change between steps.  Adam step t = t0 + (the launch's 1-based index within
// its epoch / update); its bias corrections (torch computes them in Python doubles) sit in slot t & 1, written by
// one spare lane of an earlier kernel (sg_opt_prepare) so nothing on a step's critical path evaluates pow().
struct SgOptState {
    float lr;
    float reserved[3];
    float step_size2[2];   // lr / (1 - beta1^t)
    float bc2_sqrt2[2];    // sqrt(1 - beta2^t)
    int t0;                // Adam steps completed before the current epoch / update (k_opt_commit adds to it)
    int pad[3];
};

#ifdef __HIPCC__
// tiny clears stay on the library's own launch path (the runtime's fill path is a separate blit kernel)
__attribute__((unused)) static __global__ void k_zero_f64(double* p, int n) {
    if ((int)threadIdx.x < n) p[threadIdx.x] = 0.0;
}
// End of an epoch / update: the steps just taken become part of the base count.
__attribute__((unused)) static __global__ void k_opt_commit(SgOptState* st, int n_steps) { st->t0 += n_steps; }
// bias-correction scalars of Adam step t (1-based) into slot t & 1
__device__ __forceinline__ void sg_opt_prepare(SgOptState* st, int t) {
    const double bc1 = 1.0 - pow(0.9, (double)t), bc2 = 1.0 - pow(0.999, (double)t);
    st->step_size2[t & 1] = (float)((double)st->lr / bc1);
    st->bc2_sqrt2[t & 1] = (float)sqrt(bc2);
}
#endif

// Capture `enqueue()` (kernel launches and, with a communicator, RCCL collectives on ctx->stream) into a graph and
// instantiate it.  Whether a capture goes through is only known at run time (an RCCL build / topology may refuse it; so may
// the runtime when other host threads are busy with it): any failure ends the capture, is reported ONCE on stderr with the
// HIP error that caused it, and returns 1 ("run it eagerly instead" -- the caller then stays on direct launches for that
// object); 0 = *exec is ready.  Captures are serialised within the process and use the relaxed capture mode: contexts
// driven from several host threads (tests/test_gpu_world.py) otherwise invalidate each other's captures with their own
// allocations and copies.
#include <stdio.h>

template <typename F>
static inline int sg_try_capture(sg_ctx* ctx, hipGraphExec_t* exec, F&& enqueue) {
    std::lock_guard<std::recursive_mutex> lock(sg_capture_mutex());
    hipGraph_t graph = nullptr;
    auto refuse = [&](const char* what, hipError_t e, int rc) {
        static bool told = false;
        if (!told) {
            fprintf(stderr, "[simgan_hip] graph capture not used (%s: %s%s); falling back to direct launches\n", what,
                    e != hipSuccess ? hipGetErrorString(e) : "no HIP error", rc != 0 ? ", enqueue reported an error" : "");
            told = true;
        }
        (void)hipGetLastError();
        return 1;
    };
    hipError_t e = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed);
    if (e != hipSuccess) return refuse("hipStreamBeginCapture", e, 0);
    const int rc = enqueue();
    e = hipStreamEndCapture(ctx->stream, &graph);
    if (rc != 0 || e != hipSuccess || !graph) {
        if (graph) (void)hipGraphDestroy(graph);
        return refuse("hipStreamEndCapture", e, rc);
    }
    e = hipGraphInstantiate(exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) { *exec = nullptr; return refuse("hipGraphInstantiate", e, 0); }
    return 0;
}

// collectives (sg_comm.cpp): RCCL, or the one-host loopback transport
int sg_comm_graph_ok(const sg_ctx* ctx);   // 1: the collectives are stream operations that a hipGraph capture records
void sg_comm_destroy(sg_ctx* ctx);
// sum(1 - masks) over the rollout's T+1 slots (all ranks) into a device double, no host synchronisation (sg_rollout.hip)
int sg_rollout_count_dones_dev(sg_rollout* r, double* d_out);
int sg_comm_allreduce_f32(sg_ctx* ctx, float* dev, int64_t n);
int sg_comm_allreduce_f64(sg_ctx* ctx, double* dev, int64_t n);
int sg_comm_allgather_f32(sg_ctx* ctx, const float* dev_in, float* dev_out, int64_t n_per_rank);
