// sg_policy.hip -- MLP / split policy forward, PPO clipped-surrogate gradient, gradient-norm
// clipping and Adam as gfx950 kernels.
//
// Replaces (reference, a2c/ = third_party/a2c_ppo_acktr/):
//   Policy.act / get_value / evaluate_actions          a2c/model.py:89-114, a2c/model_split.py:70-95
//   MLPBase.forward, SplitPolicyBaseNew.forward        a2c/model.py:255-264, a2c/model_split.py:187-198
//   DiagGaussian / StateDiagGaussianNew / FixedNormal  a2c/distributions.py:51-59,91-118, a2c/model_split.py:201-238
//   PPO.update                                         a2c/algo/ppo.py:65-157
//
// Kernel structure of one PPO optimizer step (E_p * M of them per update, all queued on one
// stream with no host synchronisation in between):
//   k_ppo_grad    grid (row groups, 2 parts): part 0 = actor trunk(s), part 1 = critic trunk.
//                 A workgroup keeps one trunk's parameter block (its exact HBM image) in LDS,
//                 gathers its minibatch rows by permutation index, runs forward + loss +
//                 backward entirely on LDS tiles with fp32 MFMA, and writes its partial
//                 gradient sums to a private slab (deterministic, no atomics).
//   k_ppo_reduce  sums the slabs per parameter, emits per-block sum-of-squares, bumps Adam's t.
//   k_ppo_adam    global-norm clip (max_norm/(norm+1e-6), clamped to 1) + Adam.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "sg_common.h"
#include "sg_rng.hpp"

#define HALF_LOG_2PI 0.91893853320467274178f

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

// One trunk forward on R = 16*MT rows held in LDS.  W = the trunk's parameter block in LDS.
template <int MT>
__device__ __forceinline__ void trunk_forward(const SgPolicyDesc& d, const SgTrunk& tr, const float* W,
                                              const float* X, float* H1, float* H2, float* OUT,
                                              int ldP) {
    const int ldO = d.ldO, ldH = d.ldH;
    const float* b1 = W + tr.b1;
    const float* b2 = W + tr.b2;
    const float* bh = W + tr.bh;
    sg_layer_nt<MT>(X, ldO, W + tr.w1, ldO, d.Op, d.Hp,
                    [&](int r, int c, float v) { H1[r * ldH + c] = sg_tanh(v + b1[c]); });
    __syncthreads();
    sg_layer_nt<MT>(H1, ldH, W + tr.w2, ldH, d.Hp, d.Hp,
                    [&](int r, int c, float v) { H2[r * ldH + c] = sg_tanh(v + b2[c]); });
    __syncthreads();
    sg_layer_nt<MT>(H2, ldH, W + tr.wh, ldH, d.Hp, tr.Pp,
                    [&](int r, int c, float v) { OUT[r * ldP + c] = v + bh[c]; });
}

template <int MT>
__global__ __launch_bounds__(256) void k_policy_forward(FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int R = 16 * MT;
    const SgPolicyDesc& d = a.d;
    const SgTrunk tr = d.trunk[blockIdx.y];
    float* W = smem;
    float* X = W + a.wbuf_floats;
    float* H1 = X + R * d.ldO;
    float* H2 = H1 + R * d.ldH;
    float* OUT = H2 + R * d.ldH;
    const int ldP = tr.ldP;
    sg_stage(W, a.params + tr.off, tr.size / 4);
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
        trunk_forward<MT>(d, tr, W, X, H1, H2, OUT, ldP);
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
                                    : sg_normal(a.seed, 0x41435400ull, (uint64_t)row * d.A + k);
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
static size_t fwd_lds_bytes(const SgPolicyDesc& d, int MT) {
    const int R = 16 * MT;
    return sizeof(float) * (size_t)(max_trunk_size(d, 0, d.n_trunks) + R * d.ldO + 2 * R * d.ldH + R * max_ldP(d));
}

// Runs all trunks forward over n rows (device obs) into ctx scratch `heads`; returns hld.
static int policy_forward_dev(sg_policy* p, const float* d_obs, const int64_t* d_idx, int n, float* d_heads, int hld) {
    sg_ctx* ctx = p->ctx;
    FwdArgs a;
    a.d = p->desc; a.params = p->d_params; a.obs = d_obs; a.idx = d_idx; a.n = n; a.heads = d_heads; a.hld = hld;
    a.wbuf_floats = max_trunk_size(p->desc, 0, p->desc.n_trunks);
    int MT = 4;
    while (MT > 1 && (fwd_lds_bytes(p->desc, MT) > (size_t)ctx->lds_bytes - 1024 || 16 * (MT / 2) >= n)) MT /= 2;
    SG_REQUIRE(fwd_lds_bytes(p->desc, MT) <= (size_t)ctx->lds_bytes,
               "policy forward: parameter block (%zu B) does not fit LDS", fwd_lds_bytes(p->desc, MT));
    const int R = 16 * MT;
    int gx = (n + R - 1) / R;
    if (gx > 4 * ctx->num_cu) gx = 4 * ctx->num_cu;
    dim3 grid(gx, p->desc.n_trunks);
    const size_t lds = fwd_lds_bytes(p->desc, MT);
    if (MT == 4) hipLaunchKernelGGL(k_policy_forward<4>, grid, dim3(256), lds, ctx->stream, a);
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
    h.noise = d_noise; h.seed = seed; h.action_in = d_action_in;
    h.value = d_value; h.action = d_action; h.logp = d_logp; h.ent = d_ent;
    hipLaunchKernelGGL(k_gauss_head, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, h);
    SG_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------ policy API

extern "C" int sg_policy_create(sg_ctx* ctx, int kind, int obs_dim, int act_dim, int hidden, int num_feet,
                                sg_policy** out) {
    SG_REQUIRE(ctx && out, "sg_policy_create: NULL argument");
    SG_REQUIRE(kind == SG_POLICY_MLP || kind == SG_POLICY_SPLIT, "sg_policy_create: unknown kind %d", kind);
    SG_REQUIRE(obs_dim > 0 && act_dim > 0 && hidden > 0, "sg_policy_create: bad dims");
    if (kind == SG_POLICY_SPLIT)
        SG_REQUIRE(num_feet > 0 && act_dim == 7 * num_feet,
                   "SplitPolicy: num_outputs (%d) must equal (4+3)*num_feet (%d)", act_dim, 7 * num_feet);
    SG_CHECK(hipSetDevice(ctx->device));
    sg_policy* p = new sg_policy();
    p->ctx = ctx;
    p->desc = sg_make_policy_desc(kind, obs_dim, act_dim, hidden, num_feet);
    SG_REQUIRE(fwd_lds_bytes(p->desc, 1) <= (size_t)ctx->lds_bytes,
               "sg_policy_create: a trunk's parameter block (%d floats) does not fit the %d-byte LDS",
               max_trunk_size(p->desc, 0, p->desc.n_trunks), ctx->lds_bytes);
    SG_CHECK(hipMalloc((void**)&p->d_params, sizeof(float) * p->desc.total));
    SG_CHECK(hipMemsetAsync(p->d_params, 0, sizeof(float) * p->desc.total, ctx->stream));
    *out = p;
    return 0;
}

extern "C" int sg_policy_destroy(sg_policy* p) {
    if (!p) return 0;
    (void)hipStreamSynchronize(p->ctx->stream);
    if (p->d_params) (void)hipFree(p->d_params);
    if (p->d_io) (void)hipFree(p->d_io);
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
    SG_CHECK(hipMemcpy(p->d_params, padded.data(), sizeof(float) * padded.size(), hipMemcpyHostToDevice));
    return 0;
}

extern "C" int sg_policy_get_params(sg_policy* p, float* flat, int64_t n) {
    SG_REQUIRE(p && flat, "sg_policy_get_params: NULL argument");
    SG_REQUIRE(n == sg_policy_flat_count(p->desc), "sg_policy_get_params: expected %lld floats, got %lld",
               (long long)sg_policy_flat_count(p->desc), (long long)n);
    std::vector<float> padded(p->desc.total);
    SG_CHECK(hipStreamSynchronize(p->ctx->stream));
    SG_CHECK(hipMemcpy(padded.data(), p->d_params, sizeof(float) * padded.size(), hipMemcpyDeviceToHost));
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
        if (p->d_io) SG_CHECK(hipFree(p->d_io));
        SG_CHECK(hipMalloc((void**)&p->d_io, need + need / 2));
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

// ------------------------------------------------------------------------------ PPO kernels

struct PpoArgs {
    SgPolicyDesc d;
    const float* params;
    const float *obs, *actions, *old_logp, *adv, *vpred, *ret;
    const int64_t* perm;  // minibatch rows: perm[0 .. mb)
    int mb;               // local minibatch rows
    int rows_per_wg;
    float inv_B;          // 1 / global minibatch rows
    float clip, vcoef, ecoef;
    int use_clipped;
    float* slabs;
    int slab_stride;      // floats per slab (total + 8)
    int wbuf_floats, ldPmax;
};

// KO = pad16(obs)/16, KH = pad16(hidden)/16 as compile-time constants (0 = run-time shape): fixes
// every GEMM extent so the tile engine's K/N dispatch folds away (see k_disc_grad).
template <int MT, int KO, int KH>
__global__ __launch_bounds__(256) void k_ppo_grad(PpoArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int R = 16 * MT;
    SgPolicyDesc d = a.d;
    if (KO > 0 && KH > 0) { d.Op = 16 * KO; d.ldO = d.Op + 4; d.Hp = 16 * KH; d.ldH = d.Hp + 4; }
    const int part = blockIdx.y;
    const int t0 = part == 0 ? 0 : d.n_trunks - 1;
    const int nt = part == 0 ? d.n_trunks - 1 : 1;
    const int ldO = d.ldO, ldH = d.ldH, ldP = a.ldPmax, A = d.A;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    float* W = smem;
    float* X = W + a.wbuf_floats;
    float* p = X + R * ldO;
    float *H1[2], *H2[2], *OUT[2];
    for (int t = 0; t < 2; ++t) {
        H1[t] = p; p += R * ldH;
        H2[t] = p; p += R * ldH;
        OUT[t] = p; p += R * ldP;
    }
    float* ACT = p; p += (R * A + 3) & ~3;
    float* OLDLP = p; p += R;
    float* ADV = p; p += R;
    float* VPRED = p; p += R;
    float* RET = p; p += R;
    int* VALID = reinterpret_cast<int*>(p); p += R;
    int* IDX = reinterpret_cast<int*>(p); p += R;
    float* ROWL = p; p += 2 * R;   // per-row loss terms

    float* slab = a.slabs + (size_t)blockIdx.x * a.slab_stride;
    const int row_begin = blockIdx.x * a.rows_per_wg;
    int resident = -1;
    float acc_lv = 0.f, acc_la = 0.f, acc_le = 0.f;  // meaningful in wave 0 lane 0

    int chunk = 0;
    for (int off = 0; off < a.rows_per_wg; off += R, ++chunk) {
        const int base = row_begin + off;
        const bool accumulate = chunk > 0;
        __syncthreads();
        if (tid < R) {
            const int rr = base + tid;
            const bool valid = rr < a.mb && off + tid < a.rows_per_wg;
            const int idx = valid ? (int)a.perm[rr] : 0;
            VALID[tid] = valid;
            IDX[tid] = idx;
            OLDLP[tid] = valid ? a.old_logp[idx] : 0.f;
            ADV[tid] = valid ? a.adv[idx] : 0.f;
            VPRED[tid] = valid ? a.vpred[idx] : 0.f;
            RET[tid] = valid ? a.ret[idx] : 0.f;
        }
        __syncthreads();
        for (int i = tid; i < R * d.Op; i += blockDim.x) {
            const int r = i / d.Op, c = i - r * d.Op;
            X[r * ldO + c] = (VALID[r] && c < d.O) ? a.obs[(size_t)IDX[r] * d.O + c] : 0.f;
        }
        if (part == 0)
            for (int i = tid; i < R * A; i += blockDim.x) {
                const int r = i / A, c = i - r * A;
                ACT[i] = VALID[r] ? a.actions[(size_t)IDX[r] * A + c] : 0.f;
            }
        // ---- forward of every trunk of this part
        for (int ti = 0; ti < nt; ++ti) {
            const SgTrunk tr = d.trunk[t0 + ti];
            if (resident != t0 + ti) {
                __syncthreads();
                sg_stage(W, a.params + tr.off, tr.size / 4);
                resident = t0 + ti;
            }
            __syncthreads();
            trunk_forward<MT>(d, tr, W, X, H1[ti], H2[ti], OUT[ti], ldP);
        }
        __syncthreads();
        // ---- loss and d(loss)/d(head outputs)  (a2c/algo/ppo.py:92-106)
        if (part == 1) {
            // critic: one lane per row
            if (tid < R) {
                const int r = tid;
                const float v = OUT[0][r * ldP];
                float dv = 0.f, lv = 0.f;
                if (VALID[r]) {
                    const float Rt = RET[r], vo = VPRED[r];
                    if (a.use_clipped) {
                        const float dvv = v - vo;
                        const float vc = vo + fminf(fmaxf(dvv, -a.clip), a.clip);
                        const float u = (v - Rt) * (v - Rt), w = (vc - Rt) * (vc - Rt);
                        const float m1 = u > w ? 1.f : (u < w ? 0.f : 0.5f);
                        const float pass = (dvv >= -a.clip && dvv <= a.clip) ? 1.f : 0.f;
                        dv = 0.5f * a.inv_B * (m1 * 2.f * (v - Rt) + (1.f - m1) * 2.f * (vc - Rt) * pass);
                        lv = 0.5f * fmaxf(u, w);
                    } else {
                        dv = 0.5f * a.inv_B * (-2.f) * (Rt - v);
                        lv = 0.5f * (Rt - v) * (Rt - v);
                    }
                    dv *= a.vcoef;
                }
                OUT[0][r * ldP] = dv;
                ROWL[r] = lv;
            }
        } else {
            // actor(s): 32 lanes per row, one action dimension per lane; log-prob / entropy summed
            // across the row's lanes with shuffles, every lane then forms its own d/dmean, d/dlogstd
            const SgTrunk tra = d.trunk[0];
            const bool mlp = d.kind == SG_POLICY_MLP;
            const int rows_per_pass = blockDim.x >> 5;
            for (int r = tid >> 5; r < R; r += rows_per_pass) {
                float* o0 = OUT[0] + r * ldP;
                float* o1 = OUT[1] + r * ldP;
                float logp = 0.f, ent = 0.f;
                for (int k0 = 0; k0 < A; k0 += 32) {       // A <= 32 for every shipped policy: one trip
                    const int k = k0 + (tid & 31);
                    float lp = 0.f, en = 0.f;
                    if (k < A) {
                        float mean, ls;
                        if (mlp) { mean = o0[k]; ls = W[tra.ex + k]; }
                        else if (k < d.nc) { mean = o0[k]; ls = o0[d.nc + k]; }
                        else { mean = o1[k - d.nc]; ls = o1[d.na + k - d.nc]; }
                        const float sigma = expf(ls), diff = ACT[r * A + k] - mean, lsig = logf(sigma);
                        lp = -(diff * diff) / (2.f * sigma * sigma) - lsig - HALF_LOG_2PI;
                        en = 0.5f + HALF_LOG_2PI + lsig;
                    }
#pragma unroll
                    for (int o = 1; o < 32; o <<= 1) { lp += __shfl_xor(lp, o); en += __shfl_xor(en, o); }
                    logp += lp;
                    ent += en;
                }
                const bool valid = VALID[r];
                float dlogp = 0.f, la = 0.f;
                if (valid) {
                    const float adv = ADV[r];
                    const float ratio = expf(logp - OLDLP[r]);
                    const float surr1 = ratio * adv;
                    const float surr2 = fminf(fmaxf(ratio, 1.f - a.clip), 1.f + a.clip) * adv;
                    const float w1 = surr1 < surr2 ? 1.f : (surr1 > surr2 ? 0.f : 0.5f);
                    const float inr = (ratio >= 1.f - a.clip && ratio <= 1.f + a.clip) ? 1.f : 0.f;
                    dlogp = -a.inv_B * (w1 * adv + (1.f - w1) * adv * inr) * ratio;
                    la = -fminf(surr1, surr2);
                }
                const float dent = valid ? a.ecoef * a.inv_B : 0.f;
                for (int k0 = 0; k0 < A; k0 += 32) {
                    const int k = k0 + (tid & 31);
                    if (k < A) {
                        float mean, ls, *pm, *pl;
                        if (mlp) { mean = o0[k]; ls = W[tra.ex + k]; pm = o0 + k; pl = o1 + k; }
                        else if (k < d.nc) { pm = o0 + k; pl = o0 + d.nc + k; mean = *pm; ls = *pl; }
                        else { pm = o1 + (k - d.nc); pl = o1 + (d.na + k - d.nc); mean = *pm; ls = *pl; }
                        const float sigma = expf(ls), var = sigma * sigma, diff = ACT[r * A + k] - mean;
                        *pm = dlogp * diff / var;
                        *pl = dlogp * (diff * diff / var - 1.f) - dent;
                    } else if (mlp && k < tra.Pp) {
                        o1[k] = 0.f;   // o1 doubles as the per-row d/d logstd tile: clear its padding columns
                    }
                }
                if ((tid & 31) == 0) { ROWL[r] = la; ROWL[R + r] = valid ? ent : 0.f; }
            }
        }
        __syncthreads();
        if (tid == 0) {
            if (part == 1) { for (int r = 0; r < R; ++r) acc_lv += ROWL[r]; }
            else { for (int r = 0; r < R; ++r) { acc_la += ROWL[r]; acc_le += ROWL[R + r]; } }
        }
        // ---- backward, trunks in reverse so the last-staged trunk is still resident
        for (int ti = nt - 1; ti >= 0; --ti) {
            const SgTrunk tr = d.trunk[t0 + ti];
            if (resident != t0 + ti) {
                sg_stage(W, a.params + tr.off, tr.size / 4);
                resident = t0 + ti;
                __syncthreads();
            }
            float* g = slab + tr.off;
            float* h1 = H1[ti];
            float* h2 = H2[ti];
            float* dout = OUT[ti];
            // head weight / bias gradients
            sg_grad_tn<MT>(dout, ldP, h2, ldH, tr.Pp, d.Hp, g + tr.wh, ldH, accumulate);
            sg_colsum(dout, ldP, R, tr.Pp, g + tr.bh, accumulate);
            if (tr.EX) sg_colsum(OUT[1], ldP, R, SG_PAD16(tr.EX), g + tr.ex, accumulate);
            __syncthreads();
            // dZ = (dY W) * (1 - h^2) in place over h; the bias gradient (column sum of dZ) falls out
            // of the epilogue registers
            auto dz_epilogue = [&](float* h, float* gb) {
                return [=](int tn, f32x4 (&acc)[MT][1]) {
                    const int c = tn * 16 + (tid & 15), lq = (tid & 63) >> 4;
                    float z[MT][4];
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float* ph = h + (i * 16 + 4 * lq + r) * ldH + c;
                            const float hv = *ph;
                            z[i][r] = acc[i][0][r] * (1.f - hv * hv);
                            *ph = z[i][r];
                        }
                    const float sb = sg_tile_colsum<MT>(z);
                    if (lq == 0) gb[c] = accumulate ? gb[c] + sb : sb;
                };
            };
            sg_layer_nn_t<MT>(dout, ldP, W + tr.wh, ldH, tr.Pp, d.Hp, dz_epilogue(h2, g + tr.b2));
            __syncthreads();
            sg_grad_tn<MT>(h2, ldH, h1, ldH, d.Hp, d.Hp, g + tr.w2, ldH, accumulate);
            __syncthreads();
            sg_layer_nn_t<MT>(h2, ldH, W + tr.w2, ldH, d.Hp, d.Hp, dz_epilogue(h1, g + tr.b1));
            __syncthreads();
            sg_grad_tn<MT>(h1, ldH, X, ldO, d.Hp, d.Op, g + tr.w1, ldO, accumulate);
            __syncthreads();
        }
    }
    if (tid == 0) {
        float* ls = slab + d.total;
        if (part == 1) ls[0] = acc_lv;
        else { ls[1] = acc_la; ls[2] = acc_le; }
    }
}

// shape-specialised instances for the shipped configurations at the
// tile height the launch heuristics pick for them, plus run-time-shape fallbacks
static void launch_ppo_grad(sg_ctx* ctx, int MT, const SgPolicyDesc& d, dim3 grid, size_t lds, const PpoArgs& pa) {
    const int ko = d.Op / 16, kh = d.Hp / 16;
    const dim3 block(256);
#define SG_PPO_CASE(mt, o, h) \
    if (MT == mt && ko == o && kh == h) { SG_LAUNCH(ctx, SG_PROF_PPO_GRAD, (k_ppo_grad<mt, o, h>), grid, block, lds, pa); return; }
    SG_PPO_CASE(2, 3, 4)   // north-star synthetic: obs 47, h64
    SG_PPO_CASE(4, 3, 4)
    SG_PPO_CASE(2, 1, 7)   // HopperCombined: obs 14, h100
    SG_PPO_CASE(1, 4, 7)   // LaikagoCombined: obs 64, h100
    SG_PPO_CASE(2, 7, 4)   // Laikago refinement: obs 111, h64
#undef SG_PPO_CASE
    if (MT == 4) SG_LAUNCH(ctx, SG_PROF_PPO_GRAD, (k_ppo_grad<4, 0, 0>), grid, block, lds, pa);
    else if (MT == 2) SG_LAUNCH(ctx, SG_PROF_PPO_GRAD, (k_ppo_grad<2, 0, 0>), grid, block, lds, pa);
    else SG_LAUNCH(ctx, SG_PROF_PPO_GRAD, (k_ppo_grad<1, 0, 0>), grid, block, lds, pa);
}

static size_t ppo_lds_bytes(const SgPolicyDesc& d, int MT) {
    const int R = 16 * MT;
    const int nt_actor = d.n_trunks - 1;
    const int wbuf = max_trunk_size(d, 0, d.n_trunks);
    (void)nt_actor;
    size_t f = (size_t)wbuf + R * d.ldO + 2 * (2 * R * d.ldH + R * max_ldP(d)) + ((R * d.A + 3) & ~3) + 8 * R;
    return sizeof(float) * f;
}

// grad[i] = sum over slabs; part[block] = sum of squares of this block's grads; bumps Adam's t.
__global__ __launch_bounds__(256) void k_ppo_reduce(const float* slabs, int n_slabs, int slab_stride, int total,
                                                    float* grad, float* part, SgOptState* st) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float g = 0.f;
    if (i < total + 8) {
        // 8 independent partial sums keep 8 slab loads in flight; combined in a fixed order
        float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int s = 0;
        for (; s + 8 <= n_slabs; s += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) p[u] += slabs[(size_t)(s + u) * slab_stride + i];
        }
        for (; s < n_slabs; ++s) p[0] += slabs[(size_t)s * slab_stride + i];
        g = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
        grad[i] = g;
    }
    float sq = (i < total) ? g * g : 0.f;
    sq = sg_wave_sum(sq);
    __shared__ float ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
        if (blockIdx.x == 0) sg_opt_advance(st);
    }
}

// sum of squares only (data-parallel mode: recomputed after the all-reduce)
__global__ __launch_bounds__(256) void k_sumsq(const float* grad, int total, float* part) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float sq = (i < total) ? grad[i] * grad[i] : 0.f;
    sq = sg_wave_sum(sq);
    __shared__ float ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// clip_grad_norm_ + Adam (a2c/algo/ppo.py:143-145; torch.optim.Adam single-tensor math).
__global__ __launch_bounds__(256) void k_ppo_adam(float* params, float* m, float* v, const float* grad,
                                                  const float* part, int n_part, int total,
                                                  const SgOptState* st, float eps, float max_norm,
                                                  float inv_mb, double* loss_acc) {
    __shared__ float s_coef;
    const float s_step_size = st->step_size, s_bc2_sqrt = st->bc2_sqrt;
    if (threadIdx.x < 64) {
        float s = 0.f;
        for (int j = threadIdx.x; j < n_part; j += 64) s += part[j];
        s = sg_wave_sum(s);
        if (threadIdx.x == 0) {
            const float norm = sqrtf(s);
            float coef = max_norm / (norm + 1e-6f);
            s_coef = coef > 1.f ? 1.f : coef;
        }
    }
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        const float g = grad[i] * s_coef;
        float mi = m[i], vi = v[i];
        mi = mi + (g - mi) * (float)(1.0 - 0.9);
        vi = vi * (float)0.999 + (float)(1.0 - 0.999) * g * g;
        const float denom = sqrtf(vi) / s_bc2_sqrt + eps;
        params[i] = params[i] - s_step_size * (mi / denom);
        m[i] = mi;
        v[i] = vi;
    }
    if (blockIdx.x == 0 && threadIdx.x < 3)  // value_loss.item() etc. are float32, summed in Python doubles
        loss_acc[threadIdx.x] += (double)(grad[total + threadIdx.x] * inv_mb);
}

// adv = returns[:-1] - value_preds[:-1]; sums for mean / unbiased std (a2c/algo/ppo.py:66-68)
__global__ __launch_bounds__(1024) void k_adv_stats(const float* ret, const float* vpred, int64_t n, float* adv,
                                                    double* stats /* [0]=sum, [1]=sumsq-about-mean, [2]=n */,
                                                    int pass, int finalize) {
    __shared__ double ws[16];
    __shared__ double s_mean;
    const int tid = threadIdx.x;
    if (pass == 0) {
        double s = 0.0;
        for (int64_t i = tid; i < n; i += blockDim.x) {
            const float a = ret[i] - vpred[i];
            adv[i] = a;
            s += (double)a;
        }
        s = sg_wave_sum(s);
        if ((tid & 63) == 0) ws[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += ws[w];
            stats[0] = t;
            stats[2] = (double)n;
        }
    } else if (pass == 1) {
        if (tid == 0) s_mean = (double)(float)(stats[0] / stats[2]);
        __syncthreads();
        const double mean = s_mean;
        double s = 0.0;
        for (int64_t i = tid; i < n; i += blockDim.x) {
            const double dd = (double)adv[i] - mean;
            s += dd * dd;
        }
        s = sg_wave_sum(s);
        if ((tid & 63) == 0) ws[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += ws[w];
            stats[1] = t;
        }
    } else {
        const float mean = (float)(stats[0] / stats[2]);
        const float sd = (float)sqrt(stats[1] / (stats[2] - 1.0));
        for (int64_t i = tid + (int64_t)blockIdx.x * blockDim.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
            adv[i] = (adv[i] - mean) / (sd + 1e-5f);
    }
    (void)finalize;
}

__global__ void k_fill_perm(int64_t* perm, int64_t n, int half_bits, uint64_t key) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) perm[i] = sg_perm_at(i, n, half_bits, key);
}

int sg_fill_perm(sg_ctx* ctx, int64_t* d_perm, int64_t n, uint64_t seed, uint64_t stream_id) {
    const uint64_t key = sg_key(seed, 0x5045524Dull, stream_id);
    hipLaunchKernelGGL(k_fill_perm, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_perm, n,
                       sg_perm_half_bits((uint64_t)n), key);
    SG_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------- PPO API

static int ppo_pick_mt(sg_ctx* ctx, const SgPolicyDesc& d, int rows_per_wg_hint) {
    int MT = 4;
    while (MT > 1 && (ppo_lds_bytes(d, MT) > (size_t)ctx->lds_bytes - 512 || 16 * (MT / 2) >= rows_per_wg_hint)) MT /= 2;
    return MT;
}

extern "C" int sg_ppo_create(sg_ctx* ctx, sg_policy* p, const sg_ppo_config* cfg, sg_ppo** out) {
    SG_REQUIRE(ctx && p && cfg && out, "sg_ppo_create: NULL argument");
    SG_REQUIRE(cfg->ppo_epoch > 0 && cfg->num_mini_batch > 0, "sg_ppo_create: ppo_epoch and num_mini_batch must be positive");
    SG_REQUIRE(ppo_lds_bytes(p->desc, 1) <= (size_t)ctx->lds_bytes,
               "sg_ppo_create: policy too large for the LDS-resident PPO kernel (%zu > %d bytes)",
               ppo_lds_bytes(p->desc, 1), ctx->lds_bytes);
    SG_CHECK(hipSetDevice(ctx->device));
    sg_ppo* a = new sg_ppo();
    a->ctx = ctx; a->policy = p; a->cfg = *cfg;
    const size_t tot = (size_t)p->desc.total + 8;
    SG_CHECK(hipMalloc((void**)&a->d_m, sizeof(float) * tot));
    SG_CHECK(hipMalloc((void**)&a->d_v, sizeof(float) * tot));
    SG_CHECK(hipMalloc((void**)&a->d_grad, sizeof(float) * tot));
    SG_CHECK(hipMalloc((void**)&a->d_state, sizeof(SgOptState)));
    SG_CHECK(hipMalloc((void**)&a->d_loss_acc, sizeof(double) * 8));
    SG_CHECK(hipMalloc((void**)&a->d_part, sizeof(float) * ((tot + 255) / 256 + 8)));
    SG_CHECK(hipMemsetAsync(a->d_m, 0, sizeof(float) * tot, ctx->stream));
    SG_CHECK(hipMemsetAsync(a->d_v, 0, sizeof(float) * tot, ctx->stream));
    SG_CHECK(hipMemsetAsync(a->d_loss_acc, 0, sizeof(double) * 8, ctx->stream));
    SgOptState st;
    memset(&st, 0, sizeof st);
    st.lr = cfg->lr;
    SG_CHECK(hipMemcpyAsync(a->d_state, &st, sizeof st, hipMemcpyHostToDevice, ctx->stream));
    SG_CHECK(hipStreamSynchronize(ctx->stream));
    *out = a;
    return 0;
}

extern "C" int sg_ppo_destroy(sg_ppo* a) {
    if (!a) return 0;
    (void)hipStreamSynchronize(a->ctx->stream);
    float* ptrs[] = {a->d_m, a->d_v, a->d_grad, a->d_slabs, a->d_state, a->d_part};
    for (float* q : ptrs) if (q) (void)hipFree(q);
    if (a->d_perms) (void)hipFree(a->d_perms);
    if (a->d_loss_acc) (void)hipFree(a->d_loss_acc);
    delete a;
    return 0;
}

extern "C" int sg_ppo_set_lr(sg_ppo* a, float lr) {
    SG_REQUIRE(a, "sg_ppo_set_lr: NULL argument");
    a->cfg.lr = lr;
    SG_CHECK(hipMemcpyAsync(&reinterpret_cast<SgOptState*>(a->d_state)->lr, &a->cfg.lr, sizeof(float),
                            hipMemcpyHostToDevice, a->ctx->stream));
    SG_CHECK(hipStreamSynchronize(a->ctx->stream));
    return 0;
}

extern "C" int sg_ppo_get_adam(sg_ppo* a, float* m, float* v, int64_t n, int64_t* step) {
    SG_REQUIRE(a && m && v && step, "sg_ppo_get_adam: NULL argument");
    const SgPolicyDesc& d = a->policy->desc;
    SG_REQUIRE(n == sg_policy_flat_count(d), "sg_ppo_get_adam: bad length");
    std::vector<float> pm(d.total), pv(d.total);
    SgOptState st;
    SG_CHECK(hipStreamSynchronize(a->ctx->stream));
    SG_CHECK(hipMemcpy(pm.data(), a->d_m, sizeof(float) * d.total, hipMemcpyDeviceToHost));
    SG_CHECK(hipMemcpy(pv.data(), a->d_v, sizeof(float) * d.total, hipMemcpyDeviceToHost));
    SG_CHECK(hipMemcpy(&st, a->d_state, sizeof st, hipMemcpyDeviceToHost));
    sg_policy_unpad(d, pm.data(), m);
    sg_policy_unpad(d, pv.data(), v);
    *step = (int64_t)st.step;
    return 0;
}

extern "C" int sg_ppo_set_adam(sg_ppo* a, const float* m, const float* v, int64_t n, int64_t step) {
    SG_REQUIRE(a && m && v, "sg_ppo_set_adam: NULL argument");
    const SgPolicyDesc& d = a->policy->desc;
    SG_REQUIRE(n == sg_policy_flat_count(d), "sg_ppo_set_adam: bad length");
    std::vector<float> pm(d.total, 0.f), pv(d.total, 0.f);
    sg_policy_pad(d, m, pm.data());
    sg_policy_pad(d, v, pv.data());
    const float fs = (float)step;
    SG_CHECK(hipStreamSynchronize(a->ctx->stream));
    SG_CHECK(hipMemcpy(a->d_m, pm.data(), sizeof(float) * d.total, hipMemcpyHostToDevice));
    SG_CHECK(hipMemcpy(a->d_v, pv.data(), sizeof(float) * d.total, hipMemcpyHostToDevice));
    SG_CHECK(hipMemcpy(&reinterpret_cast<SgOptState*>(a->d_state)->step, &fs, sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

extern "C" int sg_ppo_update(sg_ppo* a, sg_rollout* r, const int64_t* perms, uint64_t seed, float out3[3]) {
    SG_REQUIRE(a && r && out3, "sg_ppo_update: NULL argument");
    sg_ctx* ctx = a->ctx;
    const SgPolicyDesc& d = a->policy->desc;
    SG_REQUIRE(r->O == d.O && r->A == d.A, "sg_ppo_update: rollout dims (obs %d, act %d) do not match the policy (%d, %d)",
               r->O, r->A, d.O, d.A);
    const int64_t TN = (int64_t)r->T * r->N;
    const int M = a->cfg.num_mini_batch, E = a->cfg.ppo_epoch;
    // a2c/storage.py:152-157
    SG_REQUIRE(TN >= M, "PPO requires the number of processes (%d) * number of steps (%d) = %lld to be greater than "
               "or equal to the number of PPO mini batches (%d).", r->N, r->T, (long long)TN, M);
    SG_CHECK(hipSetDevice(ctx->device));
    const int mb = (int)(TN / M);
    const int world = ctx->world;

    // advantages (global mean / unbiased std)
    float* adv = r->d_field[SG_F_ADVANTAGES];
    double* stats = a->d_loss_acc + 4;
    hipLaunchKernelGGL(k_adv_stats, dim3(1), dim3(1024), 0, ctx->stream, r->d_field[SG_F_RETURNS],
                       r->d_field[SG_F_VALUE_PREDS], TN, adv, stats, 0, 0);
    if (ctx->use_comm) {
        // stats[0] = sum, stats[2] = n are linear: all-reduce, then every rank uses the global mean
        SG_TRY(sg_comm_allreduce_f64(ctx, stats, 3));
    }
    hipLaunchKernelGGL(k_adv_stats, dim3(1), dim3(1024), 0, ctx->stream, r->d_field[SG_F_RETURNS],
                       r->d_field[SG_F_VALUE_PREDS], TN, adv, stats, 1, 0);
    if (ctx->use_comm) {
        // only stats[1] (sum of squares about the global mean) must be reduced now; keep sum and n
        SG_TRY(sg_comm_allreduce_f64(ctx, stats + 1, 1));
    }
    hipLaunchKernelGGL(k_adv_stats, dim3(64), dim3(1024), 0, ctx->stream, r->d_field[SG_F_RETURNS],
                       r->d_field[SG_F_VALUE_PREDS], TN, adv, stats, 2, 0);
    SG_CHECK(hipGetLastError());

    // permutations
    if (a->perms_cap < (int64_t)E * TN) {
        if (a->d_perms) SG_CHECK(hipFree(a->d_perms));
        SG_CHECK(hipMalloc((void**)&a->d_perms, sizeof(int64_t) * (size_t)E * TN));
        a->perms_cap = (int64_t)E * TN;
    }
    if (perms) {
        SG_CHECK(hipMemcpyAsync(a->d_perms, perms, sizeof(int64_t) * (size_t)E * TN, hipMemcpyHostToDevice, ctx->stream));
    } else {
        for (int e = 0; e < E; ++e)
            SG_TRY(sg_fill_perm(ctx, a->d_perms + (size_t)e * TN, TN, seed, (uint64_t)e * 2654435761ull + (uint64_t)ctx->rank));
    }

    // launch geometry
    int rows_per_wg = (mb + (ctx->num_cu / 2) - 1) / (ctx->num_cu / 2);
    if (const char* e = getenv("SG_PPO_ROWS")) {   // tuning knob: rows of the minibatch per workgroup
        const int v = atoi(e);
        if (v >= 16) rows_per_wg = v;
    }
    int MT = ppo_pick_mt(ctx, d, rows_per_wg);
    const int R = 16 * MT;
    rows_per_wg = ((rows_per_wg + R - 1) / R) * R;
    const int G = (mb + rows_per_wg - 1) / rows_per_wg;
    const int slab_stride = d.total + 8;
    if (a->n_slabs < G) {
        if (a->d_slabs) SG_CHECK(hipFree(a->d_slabs));
        SG_CHECK(hipMalloc((void**)&a->d_slabs, sizeof(float) * (size_t)G * slab_stride));
        a->n_slabs = G;
    }
    // ld padding columns of the slabs are never written by the kernels: they must read as zero
    SG_CHECK(hipMemsetAsync(a->d_slabs, 0, sizeof(float) * (size_t)G * slab_stride, ctx->stream));
    SG_CHECK(hipMemsetAsync(a->d_loss_acc, 0, sizeof(double) * 3, ctx->stream));

    PpoArgs pa;
    pa.d = d; pa.params = a->policy->d_params;
    pa.obs = r->d_field[SG_F_OBS]; pa.actions = r->d_field[SG_F_ACTIONS]; pa.old_logp = r->d_field[SG_F_LOGP];
    pa.adv = adv; pa.vpred = r->d_field[SG_F_VALUE_PREDS]; pa.ret = r->d_field[SG_F_RETURNS];
    pa.mb = mb; pa.rows_per_wg = rows_per_wg; pa.inv_B = 1.0f / (float)((int64_t)mb * world);
    pa.clip = a->cfg.clip_param; pa.vcoef = a->cfg.value_loss_coef; pa.ecoef = a->cfg.entropy_coef;
    pa.use_clipped = a->cfg.use_clipped_value_loss;
    pa.slabs = a->d_slabs; pa.slab_stride = slab_stride;
    pa.wbuf_floats = max_trunk_size(d, 0, d.n_trunks); pa.ldPmax = max_ldP(d);
    const size_t lds = ppo_lds_bytes(d, MT);
    const int nblk = (d.total + 8 + 255) / 256;
    SgOptState* st = reinterpret_cast<SgOptState*>(a->d_state);
    const float inv_mb = pa.inv_B;

    for (int e = 0; e < E; ++e)
        for (int k = 0; k < M; ++k) {
            pa.perm = a->d_perms + (size_t)e * TN + (size_t)k * mb;
            launch_ppo_grad(ctx, MT, d, dim3(G, 2), lds, pa);
            SG_LAUNCH(ctx, SG_PROF_PPO_REDUCE, k_ppo_reduce, dim3(nblk), dim3(256), 0, a->d_slabs, G, slab_stride,
                      d.total, a->d_grad, a->d_part, st);
            if (ctx->use_comm) {
                SG_TRY(sg_comm_allreduce_f32(ctx, a->d_grad, d.total + 8));
                hipLaunchKernelGGL(k_sumsq, dim3(nblk), dim3(256), 0, ctx->stream, a->d_grad, d.total, a->d_part);
            }
            SG_LAUNCH(ctx, SG_PROF_PPO_ADAM, k_ppo_adam, dim3(nblk), dim3(256), 0, a->policy->d_params, a->d_m,
                      a->d_v, a->d_grad, a->d_part, nblk, d.total, st, a->cfg.eps, a->cfg.max_grad_norm,
                      inv_mb, a->d_loss_acc);
        }
    SG_CHECK(hipGetLastError());
    double acc[3];
    SG_CHECK(hipMemcpyAsync(acc, a->d_loss_acc, sizeof acc, hipMemcpyDeviceToHost, ctx->stream));
    SG_CHECK(hipStreamSynchronize(ctx->stream));
    const double nu = (double)E * M;
    for (int i = 0; i < 3; ++i) out3[i] = (float)(acc[i] / nu);
    return 0;
}
