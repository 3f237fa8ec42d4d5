// sg_policy.hip -- MLP / split policy forward (act / get_value / evaluate_actions) as gfx950 kernels.
// The PPO update lives in sg_ppo.hip.
//
// Replaces (reference, a2c/ = third_party/a2c_ppo_acktr/):
//   Policy.act / get_value / evaluate_actions          a2c/model.py:89-114, a2c/model_split.py:70-95
//   MLPBase.forward, SplitPolicyBaseNew.forward        a2c/model.py:255-264, a2c/model_split.py:187-198
//   DiagGaussian / StateDiagGaussianNew / FixedNormal  a2c/distributions.py:51-59,91-118, a2c/model_split.py:201-238
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "sg_common.h"
#include "sg_rng.hpp"

#define HALF_LOG_2PI 0.91893853320467274178f
#define SG_ACT_STREAM 0x41435400ull

// --------------------------------------------------------------------------------- forward

struct FwdArgs {
    SgPolicyDesc d;
    const float* params;
    const float* obs;     // [*, O]
    const int64_t* idx;   // optional row gather
    int n;
    float* heads;         // [n_trunks][n][hld]
    int hld;
    int wbuf_floats;
};

// One trunk forward on R = 16*MT rows held in LDS.  W = the trunk's parameter block: its LDS image, or (GW) the block in
// global memory itself -- the general-shape path for trunks that do not fit a CU's LDS (any hidden size the reference's
// --hidden-size accepts, a2c/arguments.py:107-109; weights come through L2 one MFMA fragment at a time, sg_gemm.hpp).
template <int MT, bool GW = false>
__device__ __forceinline__ void trunk_forward(const SgPolicyDesc& d, const SgTrunk& tr, const float* W,
                                              const float* X, float* H1, float* H2, float* OUT,
                                              int ldP) {
    const int ldO = d.ldO, ldH = tr.ldH, Hp = tr.Hp;   // this trunk's own width (the critic's may differ from the actors')
    const float* b1 = W + tr.b1;
    const float* b2 = W + tr.b2;
    const float* bh = W + tr.bh;
    sg_layer_nt<MT, GW>(X, ldO, W + tr.w1, ldO, d.Op, Hp,
                        [&](int r, int c, float v) { H1[r * ldH + c] = sg_tanh(v + b1[c]); });
    __syncthreads();
    sg_layer_nt<MT, GW>(H1, ldH, W + tr.w2, ldH, Hp, Hp,
                        [&](int r, int c, float v) { H2[r * ldH + c] = sg_tanh(v + b2[c]); });
    __syncthreads();
    sg_layer_nt<MT, GW>(H2, ldH, W + tr.wh, ldH, Hp, tr.Pp,
                        [&](int r, int c, float v) { OUT[r * ldP + c] = v + bh[c]; });
}

template <int MT, bool GW = false>
__global__ __launch_bounds__(256) void k_policy_forward(FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int R = 16 * MT;
    const SgPolicyDesc& d = a.d;
    const SgTrunk tr = d.trunk[blockIdx.y];
    const float* W = GW ? a.params + tr.off : smem;
    float* X = smem + (GW ? 0 : a.wbuf_floats);
    float* H1 = X + R * d.ldO;
    float* H2 = H1 + R * d.ldH;
    float* OUT = H2 + R * d.ldH;
    const int ldP = tr.ldP;
    if (!GW) sg_stage(smem, a.params + tr.off, tr.size / 4);
    float* heads = a.heads + (size_t)blockIdx.y * a.n * a.hld;
    for (int base = blockIdx.x * R; base < a.n; base += gridDim.x * R) {
        __syncthreads();
        for (int i = threadIdx.x; i < R * d.Op; i += blockDim.x) {
            const int r = i / d.Op, c = i - r * d.Op;
            const int row = base + r;
            float v = 0.f;
            if (row < a.n && c < d.O) {
                const int64_t src = a.idx ? a.idx[row] : (int64_t)row;
                v = a.obs[src * d.O + c];
            }
            X[r * d.ldO + c] = v;
        }
        __syncthreads();
        trunk_forward<MT, GW>(d, tr, W, X, H1, H2, OUT, ldP);
        __syncthreads();
        for (int i = threadIdx.x; i < R * tr.Pp; i += blockDim.x) {
            const int r = i / tr.Pp, c = i - r * tr.Pp;
            if (base + r < a.n && c < a.hld) heads[(size_t)(base + r) * a.hld + c] = OUT[r * ldP + c];
        }
    }
}

// Gaussian head: value, action (sampled / mode / given), log-prob, per-row entropy.
struct HeadArgs {
    SgPolicyDesc d;
    const float* params;
    const float* heads;
    int hld, n;
    int mode;             // 0: sample with noise (or RNG if noise NULL); 1: deterministic; 2: evaluate given action
    const float* noise;   // [n, A] or NULL
    uint64_t seed;
    uint64_t stream;      // RNG stream of this call: 'ACT\0' + data-parallel rank (every rank draws its own noise)
    const float* action_in;  // mode 2
    float *value, *action, *logp, *ent;  // any may be NULL
};

__global__ void k_gauss_head(HeadArgs a) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= a.n) return;
    const SgPolicyDesc& d = a.d;
    const size_t hs = (size_t)a.n * a.hld;
    const float* h0 = a.heads + (size_t)row * a.hld;
    const float* h1 = h0 + hs;
    const float* hc = a.heads + (size_t)(d.n_trunks - 1) * hs + (size_t)row * a.hld;
    if (a.value) a.value[row] = hc[0];
    if (!a.logp && !a.action && !a.ent) return;
    float logp = 0.f, ent = 0.f;
    for (int k = 0; k < d.A; ++k) {
        float mean, ls;
        if (d.kind == SG_POLICY_MLP) {
            mean = h0[k];
            ls = a.params[d.trunk[0].off + d.trunk[0].ex + k];
        } else if (k < d.nc) {
            mean = h0[k];
            ls = h0[d.nc + k];
        } else {
            mean = h1[k - d.nc];
            ls = h1[d.na + k - d.nc];
        }
        const float sigma = expf(ls);
        float act;
        if (a.mode == 2) act = a.action_in[(size_t)row * d.A + k];
        else if (a.mode == 1) act = mean;
        else {
            const float z = a.noise ? a.noise[(size_t)row * d.A + k]
                                    : sg_normal(a.seed, a.stream, (uint64_t)row * d.A + k);
            act = z * sigma + mean;
        }
        if (a.action) a.action[(size_t)row * d.A + k] = act;
        const float diff = act - mean, var = sigma * sigma;
        logp += -(diff * diff) / (2.0f * var) - logf(sigma) - HALF_LOG_2PI;
        ent += 0.5f + HALF_LOG_2PI + logf(sigma);
    }
    if (a.logp) a.logp[row] = logp;
    if (a.ent) a.ent[row] = ent;
}

// LDS floats needed by the forward kernel / the PPO gradient kernel for a given MT.
static int max_trunk_size(const SgPolicyDesc& d, int t0, int nt) {
    int m = 0;
    for (int t = t0; t < t0 + nt; ++t) m = d.trunk[t].size > m ? d.trunk[t].size : m;
    return m;
}
static int max_ldP(const SgPolicyDesc& d) {
    int m = 0;
    for (int t = 0; t < d.n_trunks; ++t) m = d.trunk[t].ldP > m ? d.trunk[t].ldP : m;
    return m;
}
static size_t fwd_lds_bytes(const SgPolicyDesc& d, int MT, bool gw = false) {
    const int R = 16 * MT;
    return sizeof(float) * (size_t)((gw ? 0 : max_trunk_size(d, 0, d.n_trunks)) + R * d.ldO + 2 * R * d.ldH + R * max_ldP(d));
}
// The LDS-resident kernels need a trunk's whole parameter block beside one 16-row tile; a policy that does not fit takes
// the global-weight instances (SG_POLICY_GW=1 forces them for any shape: tests).
bool sg_policy_needs_gw(const sg_ctx* ctx, const SgPolicyDesc& d) {
    const char* e = getenv("SG_POLICY_GW");
    if (e && e[0] == '1') return true;
    return fwd_lds_bytes(d, 1) > (size_t)ctx->lds_bytes - 1024;
}

// Runs all trunks forward over n rows (device obs) into ctx scratch `heads`; returns hld.
static int policy_forward_dev(sg_policy* p, const float* d_obs, const int64_t* d_idx, int n, float* d_heads, int hld) {
    sg_ctx* ctx = p->ctx;
    FwdArgs a;
    a.d = p->desc; a.params = p->d_params; a.obs = d_obs; a.idx = d_idx; a.n = n; a.heads = d_heads; a.hld = hld;
    const bool gw = sg_policy_needs_gw(ctx, p->desc);
    a.wbuf_floats = gw ? 0 : max_trunk_size(p->desc, 0, p->desc.n_trunks);
    int MT = 4;
    while (MT > 1 && (fwd_lds_bytes(p->desc, MT, gw) > (size_t)ctx->lds_bytes - 1024 || 16 * (MT / 2) >= n)) MT /= 2;
    SG_REQUIRE(fwd_lds_bytes(p->desc, MT, gw) <= (size_t)ctx->lds_bytes,
               "policy forward: one 16-row activation tile (%zu B) does not fit LDS", fwd_lds_bytes(p->desc, MT, gw));
    const int R = 16 * MT;
    int gx = (n + R - 1) / R;
    if (gx > 4 * ctx->num_cu) gx = 4 * ctx->num_cu;
    dim3 grid(gx, p->desc.n_trunks);
    const size_t lds = fwd_lds_bytes(p->desc, MT, gw);
    if (gw) {
        if (MT == 4) hipLaunchKernelGGL((k_policy_forward<4, true>), grid, dim3(256), lds, ctx->stream, a);
        else if (MT == 2) hipLaunchKernelGGL((k_policy_forward<2, true>), grid, dim3(256), lds, ctx->stream, a);
        else hipLaunchKernelGGL((k_policy_forward<1, true>), grid, dim3(256), lds, ctx->stream, a);
    } else if (MT == 4) hipLaunchKernelGGL(k_policy_forward<4>, grid, dim3(256), lds, ctx->stream, a);
    else if (MT == 2) hipLaunchKernelGGL(k_policy_forward<2>, grid, dim3(256), lds, ctx->stream, a);
    else hipLaunchKernelGGL(k_policy_forward<1>, grid, dim3(256), lds, ctx->stream, a);
    SG_CHECK(hipGetLastError());
    return 0;
}

int sg_policy_heads_ld(const SgPolicyDesc& d) {
    int m = 0;
    for (int t = 0; t < d.n_trunks; ++t) m = d.trunk[t].Pp > m ? d.trunk[t].Pp : m;
    return m;
}

// Device-side entry used by the rollout code (compute_returns_policy, synthetic fill).
int sg_policy_forward_device(sg_policy* p, const float* d_obs, int n, int mode, const float* d_noise,
                             uint64_t seed, const float* d_action_in, float* d_value, float* d_action,
                             float* d_logp, float* d_ent) {
    sg_ctx* ctx = p->ctx;
    const int hld = sg_policy_heads_ld(p->desc);
    float* scratch = nullptr;
    const size_t heads_bytes = sizeof(float) * (size_t)p->desc.n_trunks * n * hld;
    SG_TRY(sg_ctx_scratch(ctx, heads_bytes, &scratch));
    SG_TRY(policy_forward_dev(p, d_obs, nullptr, n, scratch, hld));
    HeadArgs h;
    h.d = p->desc; h.params = p->d_params; h.heads = scratch; h.hld = hld; h.n = n; h.mode = mode;
    h.noise = d_noise; h.seed = seed; h.stream = SG_ACT_STREAM + (uint64_t)ctx->rank; h.action_in = d_action_in;
    h.value = d_value; h.action = d_action; h.logp = d_logp; h.ent = d_ent;
    hipLaunchKernelGGL(k_gauss_head, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, h);
    SG_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------ policy API

extern "C" int sg_policy_create(sg_ctx* ctx, int kind, int obs_dim, int act_dim, int hidden, int num_feet,
                                sg_policy** out) {
    return sg_policy_create2(ctx, kind, obs_dim, act_dim, hidden, num_feet, 0, out);
}

extern "C" int sg_policy_create2(sg_ctx* ctx, int kind, int obs_dim, int act_dim, int hidden, int num_feet, int critic_hidden,
                                 sg_policy** out) {
    SG_DEVICE_WIDE();
    SG_REQUIRE(ctx && out, "sg_policy_create: NULL argument");
    SG_REQUIRE(critic_hidden >= 0, "sg_policy_create2: bad critic_hidden");
    SG_REQUIRE(kind == SG_POLICY_MLP || kind == SG_POLICY_SPLIT, "sg_policy_create: unknown kind %d", kind);
    SG_REQUIRE(obs_dim > 0 && act_dim > 0 && hidden > 0, "sg_policy_create: bad dims");
    if (kind == SG_POLICY_SPLIT)
        SG_REQUIRE(num_feet > 0 && act_dim == 7 * num_feet,
                   "SplitPolicy: num_outputs (%d) must equal (4+3)*num_feet (%d)", act_dim, 7 * num_feet);
    SG_CHECK(hipSetDevice(ctx->device));
    sg_policy* p = new sg_policy();
    p->ctx = ctx;
    p->desc = sg_make_policy_desc(kind, obs_dim, act_dim, hidden, num_feet, critic_hidden);
    // any width the reference's constructor accepts (a2c/arguments.py:107-109, a2c/model.py:233-253): a trunk that fits a
    // CU's LDS runs on the LDS-resident kernels, a larger one on the global-weight instances; only one 16-row activation
    // tile has to fit
    SG_REQUIRE(fwd_lds_bytes(p->desc, 1, true) <= (size_t)ctx->lds_bytes - 1024,
               "sg_policy_create: one 16-row activation tile of this policy (obs %d, hidden %d: %zu bytes) does not fit the %d-byte LDS",
               obs_dim, hidden, fwd_lds_bytes(p->desc, 1, true), ctx->lds_bytes);
    SG_CHECK(sg_dev_malloc((void**)&p->d_params, sizeof(float) * p->desc.total));
    SG_CHECK(hipMemsetAsync(p->d_params, 0, sizeof(float) * p->desc.total, ctx->stream));
    *out = p;
    return 0;
}

extern "C" int sg_policy_destroy(sg_policy* p) {
    SG_DEVICE_WIDE();
    if (!p) return 0;
    (void)hipStreamSynchronize(p->ctx->stream);
    if (p->d_params) (void)sg_dev_free(p->d_params);
    if (p->d_io) (void)sg_dev_free(p->d_io);
    delete p;
    return 0;
}

extern "C" int sg_policy_num_params(const sg_policy* p, int64_t* n) {
    SG_REQUIRE(p && n, "sg_policy_num_params: NULL argument");
    *n = sg_policy_flat_count(p->desc);
    return 0;
}

extern "C" int sg_policy_set_params(sg_policy* p, const float* flat, int64_t n) {
    SG_REQUIRE(p && flat, "sg_policy_set_params: NULL argument");
    SG_REQUIRE(n == sg_policy_flat_count(p->desc), "sg_policy_set_params: expected %lld floats, got %lld",
               (long long)sg_policy_flat_count(p->desc), (long long)n);
    std::vector<float> padded(p->desc.total, 0.f);
    sg_policy_pad(p->desc, flat, padded.data());
    SG_CHECK(hipStreamSynchronize(p->ctx->stream));
    SG_COPY_SYNC(p->ctx, p->d_params, padded.data(), sizeof(float) * padded.size(), hipMemcpyHostToDevice);
    return 0;
}

extern "C" int sg_policy_get_params(sg_policy* p, float* flat, int64_t n) {
    SG_REQUIRE(p && flat, "sg_policy_get_params: NULL argument");
    SG_REQUIRE(n == sg_policy_flat_count(p->desc), "sg_policy_get_params: expected %lld floats, got %lld",
               (long long)sg_policy_flat_count(p->desc), (long long)n);
    std::vector<float> padded(p->desc.total);
    SG_CHECK(hipStreamSynchronize(p->ctx->stream));
    SG_COPY_SYNC(p->ctx, padded.data(), p->d_params, sizeof(float) * padded.size(), hipMemcpyDeviceToHost);
    sg_policy_unpad(p->desc, padded.data(), flat);
    return 0;
}

// host-pointer front end shared by act / get_value / evaluate
static int policy_host_call(sg_policy* p, const float* obs, int n, int mode, const float* noise, uint64_t seed,
                            const float* action_in, float* value, float* action, float* logp, float* ent_rows) {
    sg_ctx* ctx = p->ctx;
    SG_REQUIRE(n > 0, "policy: n must be positive");
    const int O = p->desc.O, A = p->desc.A;
    SG_CHECK(hipSetDevice(ctx->device));
    // device staging: obs | noise/action_in | action | value | logp | ent
    const size_t f_obs = (size_t)n * O, f_na = (size_t)n * A;
    const size_t need = sizeof(float) * (f_obs + 2 * f_na + 3 * (size_t)n);
    if (need > p->io_bytes) {
        SG_CHECK(hipStreamSynchronize(ctx->stream));
        if (p->d_io) SG_CHECK(sg_dev_free(p->d_io));
        SG_CHECK(sg_dev_malloc((void**)&p->d_io, need + need / 2));
        p->io_bytes = need + need / 2;
    }
    float* d_io = p->d_io;
    float* d_obs = d_io;
    float* d_in = d_obs + f_obs;
    float* d_action = d_in + f_na;
    float* d_value = d_action + f_na;
    float* d_logp = d_value + n;
    float* d_ent = d_logp + n;
    int rc = 0;
    do {
        if (hipMemcpyAsync(d_obs, obs, sizeof(float) * f_obs, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { rc = -1; break; }
        const float* src_in = mode == 2 ? action_in : noise;
        if (src_in && hipMemcpyAsync(d_in, src_in, sizeof(float) * f_na, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { rc = -1; break; }
        rc = sg_policy_forward_device(p, d_obs, n, mode, (mode == 0 && noise) ? d_in : nullptr, seed,
                                      mode == 2 ? d_in : nullptr, d_value, d_action, d_logp, d_ent);
        if (rc) break;
        if (value && hipMemcpyAsync(value, d_value, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { rc = -1; break; }
        if (action && hipMemcpyAsync(action, d_action, sizeof(float) * f_na, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { rc = -1; break; }
        if (logp && hipMemcpyAsync(logp, d_logp, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { rc = -1; break; }
        if (ent_rows && hipMemcpyAsync(ent_rows, d_ent, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { rc = -1; break; }
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = -1; break; }
    } while (0);
    if (rc == -1) sg_set_error("policy call: HIP error: %s", hipGetErrorString(hipGetLastError()));
    return rc;
}

extern "C" int sg_policy_act(sg_policy* p, const float* obs, int n, const float* noise, uint64_t seed,
                             int deterministic, float* value, float* action, float* logp) {
    SG_REQUIRE(p && obs && value && action && logp, "sg_policy_act: NULL argument");
    return policy_host_call(p, obs, n, deterministic ? 1 : 0, noise, seed, nullptr, value, action, logp, nullptr);
}

extern "C" int sg_policy_get_value(sg_policy* p, const float* obs, int n, float* value) {
    SG_REQUIRE(p && obs && value, "sg_policy_get_value: NULL argument");
    return policy_host_call(p, obs, n, 1, nullptr, 0, nullptr, value, nullptr, nullptr, nullptr);
}


// ------------------------------------------------------------------- ensemble act
// One launch for "row i acts with policy idx[i]" over K resident weight sets of one shape.  Block (tile, k):
//   1. ranks the rows that drew policy k (a block-wide count + exclusive scan over idx, stable in row order) and
//      keeps the R = 16*MT of them that belong to its tile -- blocks past the end of policy k's rows exit;
//   2. per trunk: stages policy k's parameter block in LDS, runs the LDS-tile MFMA forward on the gathered rows;
//   3. Gaussian head per row (sample / mode), written back at the ORIGINAL row positions.
// Deterministic (no atomics); the noise of row i is indexed by i, so the result does not depend on the grouping.
struct EnsArgs {
    SgPolicyDesc d;
    const float* params[SG_ENSEMBLE_MAX];
    int K;
    const int32_t* idx;   // [n]
    const float* obs;     // [n, O]
    int n;
    int mode;             // 0 sample, 1 deterministic
    const float* noise;   // [n, A] or NULL
    uint64_t seed, stream;
    float *value, *action, *logp;
    int wbuf_floats, ldPmax;
};

template <int MT, bool GW = false>
__global__ __launch_bounds__(256) void k_policy_act_ensemble(EnsArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int R = 16 * MT;
    const SgPolicyDesc& d = a.d;
    const int tid = threadIdx.x, k = blockIdx.y, tile = blockIdx.x;
    float* X = smem + a.wbuf_floats;
    float* H1 = X + R * d.ldO;
    float* H2 = H1 + R * d.ldH;
    float* OUT = H2 + R * d.ldH;                       // [n_trunks][R][ldPmax]
    int* rows = reinterpret_cast<int*>(OUT + d.n_trunks * R * a.ldPmax);   // [R]
    int* cnt = rows + R;                               // [256] + total
    // 1. rows of policy k, ranked in row order
    const int chunk = (a.n + 255) / 256, lo = tid * chunk, hi = min(a.n, lo + chunk);
    int c = 0;
    for (int i = lo; i < hi; ++i) c += (a.idx[i] == k);
    cnt[tid] = c;
    if (tid < R) rows[tid] = -1;
    __syncthreads();
    if (tid < 64) {   // exclusive scan of the 256 counts by one wave: 4 per lane + a wave scan
        int v[4], s = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) { v[u] = cnt[4 * tid + u]; s += v[u]; }
        int inc = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (tid >= o) inc += t; }
        int base = inc - s;
#pragma unroll
        for (int u = 0; u < 4; ++u) { cnt[4 * tid + u] = base; base += v[u]; }
        if (tid == 63) cnt[256] = inc;
    }
    __syncthreads();
    const int total = cnt[256], first = tile * R;
    if (first >= total) return;
    const int nrows = min(R, total - first);
    {
        int rank = cnt[tid];
        for (int i = lo; i < hi; ++i)
            if (a.idx[i] == k) { if (rank >= first && rank < first + R) rows[rank - first] = i; ++rank; }
    }
    __syncthreads();
    for (int i = tid; i < R * d.Op; i += blockDim.x) {
        const int r = i / d.Op, cc = i - r * d.Op;
        const int row = rows[r];
        X[r * d.ldO + cc] = (row >= 0 && cc < d.O) ? a.obs[(size_t)row * d.O + cc] : 0.f;
    }
    // 2. trunks
    for (int t = 0; t < d.n_trunks; ++t) {
        const SgTrunk tr = d.trunk[t];
        __syncthreads();
        if (!GW) sg_stage(smem, a.params[k] + tr.off, tr.size / 4);
        __syncthreads();
        trunk_forward<MT, GW>(d, tr, GW ? a.params[k] + tr.off : smem, X, H1, H2, OUT + t * R * a.ldPmax, tr.ldP);
    }
    __syncthreads();
    // 3. heads
    if (tid < nrows) {
        const int row = rows[tid];
        const float* h0 = OUT + tid * d.trunk[0].ldP;
        const float* h1 = OUT + R * a.ldPmax + tid * d.trunk[1].ldP;
        const float* hc = OUT + (d.n_trunks - 1) * R * a.ldPmax + tid * d.trunk[d.n_trunks - 1].ldP;
        a.value[row] = hc[0];
        float logp = 0.f;
        for (int j = 0; j < d.A; ++j) {
            float mean, ls;
            if (d.kind == SG_POLICY_MLP) { mean = h0[j]; ls = a.params[k][d.trunk[0].off + d.trunk[0].ex + j]; }
            else if (j < d.nc) { mean = h0[j]; ls = h0[d.nc + j]; }
            else { mean = h1[j - d.nc]; ls = h1[d.na + j - d.nc]; }
            const float sigma = expf(ls);
            float act = mean;
            if (a.mode == 0) {
                const float z = a.noise ? a.noise[(size_t)row * d.A + j] : sg_normal(a.seed, a.stream, (uint64_t)row * d.A + j);
                act = z * sigma + mean;
            }
            a.action[(size_t)row * d.A + j] = act;
            const float diff = act - mean, var = sigma * sigma;
            logp += -(diff * diff) / (2.0f * var) - logf(sigma) - HALF_LOG_2PI;
        }
        a.logp[row] = logp;
    }
}

static size_t ens_lds_bytes(const SgPolicyDesc& d, int MT, bool gw = false) {
    const int R = 16 * MT;
    return sizeof(float) * (size_t)((gw ? 0 : max_trunk_size(d, 0, d.n_trunks)) + R * d.ldO + 2 * R * d.ldH + d.n_trunks * R * max_ldP(d)) +
           sizeof(int) * (size_t)(R + 256 + 4);
}

extern "C" int sg_policy_act_ensemble(sg_policy* const* policies, int n_policies, const int32_t* idx, const float* obs,
                                      int n, const float* noise, uint64_t seed, int deterministic, float* value,
                                      float* action, float* logp) {
    SG_REQUIRE(policies && idx && obs && value && action && logp, "sg_policy_act_ensemble: NULL argument");
    SG_REQUIRE(n_policies > 0 && n_policies <= SG_ENSEMBLE_MAX, "sg_policy_act_ensemble: 1..%d policies, got %d", SG_ENSEMBLE_MAX, n_policies);
    SG_REQUIRE(n > 0, "sg_policy_act_ensemble: n must be positive");
    sg_policy* p0 = policies[0];
    SG_REQUIRE(p0, "sg_policy_act_ensemble: NULL policy");
    sg_ctx* ctx = p0->ctx;
    const SgPolicyDesc& d = p0->desc;
    EnsArgs a;
    for (int k = 0; k < n_policies; ++k) {
        SG_REQUIRE(policies[k] && policies[k]->ctx == ctx, "sg_policy_act_ensemble: member %d is NULL or lives on another context", k);
        const SgPolicyDesc& e = policies[k]->desc;
        SG_REQUIRE(e.kind == d.kind && e.O == d.O && e.A == d.A && e.H == d.H && e.Hc == d.Hc && e.num_feet == d.num_feet,
                   "sg_policy_act_ensemble: member %d has a different shape", k);
        a.params[k] = policies[k]->d_params;
    }
    for (int i = 0; i < n; ++i)
        SG_REQUIRE(idx[i] >= 0 && idx[i] < n_policies, "sg_policy_act_ensemble: idx[%d] = %d outside [0, %d)", i, idx[i], n_policies);
    SG_CHECK(hipSetDevice(ctx->device));
    const int O = d.O, A = d.A;
    // device staging in the first member's io buffer: obs | noise | action | value | logp | idx
    const size_t f_obs = (size_t)n * O, f_na = (size_t)n * A;
    const size_t need = sizeof(float) * (f_obs + 2 * f_na + 3 * (size_t)n);
    if (need > p0->io_bytes) {
        SG_CHECK(hipStreamSynchronize(ctx->stream));
        if (p0->d_io) SG_CHECK(sg_dev_free(p0->d_io));
        SG_CHECK(sg_dev_malloc((void**)&p0->d_io, need + need / 2));
        p0->io_bytes = need + need / 2;
    }
    float* d_obs = p0->d_io;
    float* d_noise = d_obs + f_obs;
    float* d_action = d_noise + f_na;
    float* d_value = d_action + f_na;
    float* d_logp = d_value + n;
    int32_t* d_idx = reinterpret_cast<int32_t*>(d_logp + n);
    SG_CHECK(hipMemcpyAsync(d_obs, obs, sizeof(float) * f_obs, hipMemcpyHostToDevice, ctx->stream));
    SG_CHECK(hipMemcpyAsync(d_idx, idx, sizeof(int32_t) * n, hipMemcpyHostToDevice, ctx->stream));
    if (noise && !deterministic) SG_CHECK(hipMemcpyAsync(d_noise, noise, sizeof(float) * f_na, hipMemcpyHostToDevice, ctx->stream));
    a.d = d; a.K = n_policies; a.idx = d_idx; a.obs = d_obs; a.n = n; a.mode = deterministic ? 1 : 0;
    a.noise = (noise && !deterministic) ? d_noise : nullptr; a.seed = seed; a.stream = SG_ACT_STREAM + (uint64_t)ctx->rank;
    a.value = d_value; a.action = d_action; a.logp = d_logp;
    const bool gw = sg_policy_needs_gw(ctx, d) || ens_lds_bytes(d, 1) > (size_t)ctx->lds_bytes - 1024;
    a.wbuf_floats = gw ? 0 : max_trunk_size(d, 0, d.n_trunks); a.ldPmax = max_ldP(d);
    int MT = 2;   // 32-row tiles unless the pool is small or LDS is short
    while (MT > 1 && (ens_lds_bytes(d, MT, gw) > (size_t)ctx->lds_bytes - 1024 || 16 * MT * n_policies > 2 * n)) MT /= 2;
    SG_REQUIRE(ens_lds_bytes(d, MT, gw) <= (size_t)ctx->lds_bytes, "sg_policy_act_ensemble: an activation tile does not fit LDS (%zu B)", ens_lds_bytes(d, MT, gw));
    const int R = 16 * MT;
    const dim3 grid((n + R - 1) / R, n_policies);
    const size_t lds_e = ens_lds_bytes(d, MT, gw);
    if (gw) {
        if (MT == 2) hipLaunchKernelGGL((k_policy_act_ensemble<2, true>), grid, dim3(256), lds_e, ctx->stream, a);
        else hipLaunchKernelGGL((k_policy_act_ensemble<1, true>), grid, dim3(256), lds_e, ctx->stream, a);
    } else if (MT == 2) hipLaunchKernelGGL(k_policy_act_ensemble<2>, grid, dim3(256), lds_e, ctx->stream, a);
    else hipLaunchKernelGGL(k_policy_act_ensemble<1>, grid, dim3(256), lds_e, ctx->stream, a);
    SG_CHECK(hipGetLastError());
    SG_CHECK(hipMemcpyAsync(value, d_value, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream));
    SG_CHECK(hipMemcpyAsync(action, d_action, sizeof(float) * f_na, hipMemcpyDeviceToHost, ctx->stream));
    SG_CHECK(hipMemcpyAsync(logp, d_logp, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream));
    SG_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int sg_policy_evaluate(sg_policy* p, const float* obs, const float* action, int n, float* value,
                                  float* logp, float* entropy) {
    SG_REQUIRE(p && obs && action && value && logp && entropy, "sg_policy_evaluate: NULL argument");
    std::vector<float> ent(n);
    SG_TRY(policy_host_call(p, obs, n, 2, nullptr, 0, action, value, nullptr, logp, ent.data()));
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += ent[i];
    *entropy = (float)(s / n);
    return 0;
}
