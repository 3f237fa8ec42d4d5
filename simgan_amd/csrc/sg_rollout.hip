// sg_rollout.hip -- device-resident RolloutStorage and the GAE return scan.
//
// Replaces (reference): RolloutStorage.__init__/insert/after_update/compute_returns
// a2c/storage.py:32-142.  Buffers keep the reference's (t, n, feature) row-major order, so the
// flattened row id of a2c/storage.py:168-185 is t*N + n and "next" rows are at +N.
#include <string.h>

#include <vector>

#include "sg_common.h"
#include "sg_rng.hpp"

int sg_policy_forward_device(sg_policy* p, const float* d_obs, int n, int mode, const float* d_noise,
                             uint64_t seed, const float* d_action_in, float* d_value, float* d_action,
                             float* d_logp, float* d_ent);

extern "C" int sg_rollout_create(sg_ctx* ctx, int T, int N, int obs_dim, int act_dim, int feat_dim,
                                 sg_rollout** out) {
    SG_DEVICE_WIDE();
    SG_REQUIRE(ctx && out, "sg_rollout_create: NULL argument");
    SG_REQUIRE(T > 0 && N > 0 && obs_dim > 0 && act_dim > 0 && feat_dim >= 0, "sg_rollout_create: bad dims");
    SG_CHECK(hipSetDevice(ctx->device));
    sg_rollout* r = new sg_rollout();
    r->ctx = ctx; r->T = T; r->N = N; r->O = obs_dim; r->A = act_dim; r->F = feat_dim;
    const int slots[SG_F_COUNT] = {T + 1, T + 1, T, T, T + 1, T + 1, T, T + 1, T + 1, T};
    const int width[SG_F_COUNT] = {obs_dim, feat_dim, act_dim, 1, 1, 1, 1, 1, 1, 1};
    for (int f = 0; f < SG_F_COUNT; ++f) {
        r->field_slots[f] = slots[f];
        r->field_width[f] = width[f];
        r->field_count[f] = (int64_t)slots[f] * N * width[f];
        const size_t bytes = sizeof(float) * (size_t)(r->field_count[f] > 0 ? r->field_count[f] : 1);
        SG_CHECK(sg_dev_malloc((void**)&r->d_field[f], bytes));
        SG_CHECK(hipMemsetAsync(r->d_field[f], 0, bytes, ctx->stream));
    }
    // masks / bad_masks start at one (a2c/storage.py:50-54)
    std::vector<float> ones((size_t)(T + 1) * N, 1.0f);
    SG_CHECK(hipMemcpyAsync(r->d_field[SG_F_MASKS], ones.data(), sizeof(float) * ones.size(), hipMemcpyHostToDevice, ctx->stream));
    SG_CHECK(hipMemcpyAsync(r->d_field[SG_F_BAD_MASKS], ones.data(), sizeof(float) * ones.size(), hipMemcpyHostToDevice, ctx->stream));
    SG_CHECK(hipStreamSynchronize(ctx->stream));
    *out = r;
    return 0;
}

extern "C" int sg_rollout_destroy(sg_rollout* r) {
    SG_DEVICE_WIDE();
    if (!r) return 0;
    (void)hipStreamSynchronize(r->ctx->stream);
    for (int f = 0; f < SG_F_COUNT; ++f)
        if (r->d_field[f]) (void)sg_dev_free(r->d_field[f]);
    if (r->d_perm) (void)sg_dev_free(r->d_perm);
    delete r;
    return 0;
}

static int check_field(const sg_rollout* r, int field, const char* who) {
    SG_REQUIRE(r, "%s: rollout is NULL", who);
    SG_REQUIRE(field >= 0 && field < SG_F_COUNT, "%s: unknown field %d", who, field);
    return 0;
}

extern "C" int sg_rollout_upload(sg_rollout* r, int field, const float* host, int64_t count) {
    SG_TRY(check_field(r, field, "sg_rollout_upload"));
    SG_REQUIRE(host, "sg_rollout_upload: host is NULL");
    SG_REQUIRE(count == r->field_count[field], "sg_rollout_upload: field %d holds %lld floats, got %lld", field,
               (long long)r->field_count[field], (long long)count);
    SG_REQUIRE(field != SG_F_ADVANTAGES, "sg_rollout_upload: advantages are read-only");
    if (count == 0) return 0;
    SG_CHECK(hipMemcpyAsync(r->d_field[field], host, sizeof(float) * count, hipMemcpyHostToDevice, r->ctx->stream));
    SG_CHECK(hipStreamSynchronize(r->ctx->stream));
    if (field == SG_F_OBS_FEAT) r->feat_version = sg_next_feat_version();
    return 0;
}

extern "C" int sg_rollout_download(sg_rollout* r, int field, float* host, int64_t count) {
    SG_TRY(check_field(r, field, "sg_rollout_download"));
    SG_REQUIRE(host, "sg_rollout_download: host is NULL");
    SG_REQUIRE(count == r->field_count[field], "sg_rollout_download: field %d holds %lld floats, got %lld", field,
               (long long)r->field_count[field], (long long)count);
    if (count == 0) return 0;
    SG_CHECK(hipMemcpyAsync(host, r->d_field[field], sizeof(float) * count, hipMemcpyDeviceToHost, r->ctx->stream));
    SG_CHECK(hipStreamSynchronize(r->ctx->stream));
    return 0;
}

extern "C" int sg_rollout_upload_step(sg_rollout* r, int field, int t, const float* host, int64_t count) {
    SG_TRY(check_field(r, field, "sg_rollout_upload_step"));
    SG_REQUIRE(host, "sg_rollout_upload_step: host is NULL");
    const int64_t per = (int64_t)r->N * r->field_width[field];
    SG_REQUIRE(t >= 0 && t < r->field_slots[field], "sg_rollout_upload_step: slot %d out of range [0,%d)", t, r->field_slots[field]);
    SG_REQUIRE(count == per, "sg_rollout_upload_step: a slot of field %d holds %lld floats, got %lld", field, (long long)per, (long long)count);
    if (count == 0) return 0;
    SG_CHECK(hipMemcpyAsync(r->d_field[field] + (size_t)t * per, host, sizeof(float) * per, hipMemcpyHostToDevice, r->ctx->stream));
    SG_CHECK(hipStreamSynchronize(r->ctx->stream));
    if (field == SG_F_OBS_FEAT) r->feat_version = sg_next_feat_version();
    return 0;
}

extern "C" int sg_rollout_download_step(sg_rollout* r, int field, int t, float* host, int64_t count) {
    SG_TRY(check_field(r, field, "sg_rollout_download_step"));
    SG_REQUIRE(host, "sg_rollout_download_step: host is NULL");
    const int64_t per = (int64_t)r->N * r->field_width[field];
    SG_REQUIRE(t >= 0 && t < r->field_slots[field], "sg_rollout_download_step: slot %d out of range [0,%d)", t, r->field_slots[field]);
    SG_REQUIRE(count == per, "sg_rollout_download_step: a slot of field %d holds %lld floats, got %lld", field, (long long)per, (long long)count);
    if (count == 0) return 0;
    SG_CHECK(hipMemcpyAsync(host, r->d_field[field] + (size_t)t * per, sizeof(float) * per, hipMemcpyDeviceToHost, r->ctx->stream));
    SG_CHECK(hipStreamSynchronize(r->ctx->stream));
    return 0;
}

// slot T -> slot 0 of obs, obs_feat, masks, bad_masks in one launch (four small device-to-device copies
// through the runtime's copy path cost more than the copy and occasionally stall for milliseconds)
struct AfterUpdateArgs {
    float* base[4];
    int64_t per[4];   // floats per time slot
    int T;
};
__global__ __launch_bounds__(256) void k_after_update(AfterUpdateArgs a) {
    const int f = blockIdx.y;
    float* p = a.base[f];
    const int64_t per = a.per[f];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (int64_t)gridDim.x * blockDim.x)
        p[i] = p[(int64_t)a.T * per + i];
}

extern "C" int sg_rollout_after_update(sg_rollout* r) {
    SG_REQUIRE(r, "sg_rollout_after_update: NULL argument");
    const int fields[4] = {SG_F_OBS, SG_F_OBS_FEAT, SG_F_MASKS, SG_F_BAD_MASKS};
    AfterUpdateArgs a;
    a.T = r->T;
    for (int i = 0; i < 4; ++i) {
        a.base[i] = r->d_field[fields[i]];
        a.per[i] = (int64_t)r->N * r->field_width[fields[i]];
    }
    hipLaunchKernelGGL(k_after_update, dim3(32, 4), dim3(256), 0, r->ctx->stream, a);
    r->feat_version = sg_next_feat_version();
    SG_CHECK(hipGetLastError());
    return 0;
}

// One thread per environment column; reverse scan over T (a2c/storage.py:109-142).
// The recurrence is serial in t, but its inputs are not: each chunk of 8 time steps is fetched with 32 independent
// loads before the 8 dependent updates, so a column pays one memory round trip per chunk instead of one per step.
__global__ void k_compute_returns(int T, int N, const float* __restrict__ rewards, float* value_preds,
                                  float* returns, const float* __restrict__ masks,
                                  const float* __restrict__ bad_masks, const float* __restrict__ next_value,
                                  int use_gae, float gamma, float lam, int proper) {
    constexpr int C = 8;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float nv = next_value[n];
    float gae = 0.f, v_next = nv, ret = nv;
    if (use_gae) value_preds[(size_t)T * N + n] = nv;
    else returns[(size_t)T * N + n] = nv;
    for (int t1 = T - 1; t1 >= 0; t1 -= C) {
        float rw[C], vp[C], mk[C], bm[C];
#pragma unroll
        for (int u = 0; u < C; ++u) {
            const int t = t1 - u >= 0 ? t1 - u : 0;
            const size_t i = (size_t)t * N + n, j = i + N;
            rw[u] = rewards[i]; vp[u] = value_preds[i]; mk[u] = masks[j]; bm[u] = bad_masks[j];
        }
#pragma unroll
        for (int u = 0; u < C; ++u) {
            const int t = t1 - u;
            if (t < 0) break;
            const size_t i = (size_t)t * N + n;
            if (use_gae) {
                const float delta = rw[u] + gamma * v_next * mk[u] - vp[u];
                gae = delta + gamma * lam * mk[u] * gae;
                if (proper) gae = gae * bm[u];
                returns[i] = gae + vp[u];
                v_next = vp[u];
            } else {
                if (proper) ret = (ret * gamma * mk[u] + rw[u]) * bm[u] + (1.f - bm[u]) * vp[u];
                else ret = ret * gamma * mk[u] + rw[u];
                returns[i] = ret;
            }
        }
    }
}

static int compute_returns_dev(sg_rollout* r, const float* d_next_value, int use_gae, float gamma, float lam, int proper) {
    hipLaunchKernelGGL(k_compute_returns, dim3((r->N + 63) / 64), dim3(64), 0, r->ctx->stream, r->T, r->N,
                       r->d_field[SG_F_REWARDS], r->d_field[SG_F_VALUE_PREDS], r->d_field[SG_F_RETURNS],
                       r->d_field[SG_F_MASKS], r->d_field[SG_F_BAD_MASKS], d_next_value, use_gae, gamma, lam, proper);
    SG_CHECK(hipGetLastError());
    return 0;
}

extern "C" int sg_rollout_compute_returns(sg_rollout* r, const float* next_value, int use_gae, float gamma,
                                          float gae_lambda, int use_proper_time_limits) {
    SG_REQUIRE(r && next_value, "sg_rollout_compute_returns: NULL argument");
    float* d_nv = nullptr;
    SG_TRY(sg_ctx_scratch(r->ctx, sizeof(float) * r->N, &d_nv));
    SG_CHECK(hipMemcpyAsync(d_nv, next_value, sizeof(float) * r->N, hipMemcpyHostToDevice, r->ctx->stream));
    SG_TRY(compute_returns_dev(r, d_nv, use_gae, gamma, gae_lambda, use_proper_time_limits));
    SG_CHECK(hipStreamSynchronize(r->ctx->stream));
    return 0;
}

extern "C" int sg_rollout_compute_returns_policy(sg_rollout* r, sg_policy* p, int use_gae, float gamma,
                                                 float gae_lambda, int use_proper_time_limits) {
    SG_REQUIRE(r && p, "sg_rollout_compute_returns_policy: NULL argument");
    SG_REQUIRE(p->desc.O == r->O, "sg_rollout_compute_returns_policy: obs dim mismatch");
    // next_value = get_value(obs[T]) lands in returns[T] (a slot the GAE branch never reads or writes)
    float* d_nv = r->d_field[SG_F_RETURNS] + (size_t)r->T * r->N;
    SG_TRY(sg_policy_forward_device(p, r->d_field[SG_F_OBS] + (size_t)r->T * r->N * r->O, r->N, 1, nullptr, 0, nullptr,
                                    d_nv, nullptr, nullptr, nullptr));
    SG_TRY(compute_returns_dev(r, d_nv, use_gae, gamma, gae_lambda, use_proper_time_limits));
    return 0;
}

__global__ void k_count_dones(const float* masks, int64_t n, double* out) {
    __shared__ double ws[16];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += 1.0 - (double)masks[i];
    s = sg_wave_sum(s);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += ws[w];
        *out = t;
    }
}

int sg_rollout_count_dones_dev(sg_rollout* r, double* d_out) {
    hipLaunchKernelGGL(k_count_dones, dim3(1), dim3(1024), 0, r->ctx->stream, r->d_field[SG_F_MASKS],
                       (int64_t)(r->T + 1) * r->N, d_out);
    SG_CHECK(hipGetLastError());
    if (r->ctx->use_comm) SG_TRY(sg_comm_allreduce_f64(r->ctx, d_out, 1));
    return 0;
}

extern "C" int sg_rollout_count_dones(sg_rollout* r, double* dones) {
    SG_REQUIRE(r && dones, "sg_rollout_count_dones: NULL argument");
    float* scratch = nullptr;
    SG_TRY(sg_ctx_scratch(r->ctx, 64, &scratch));
    double* d_out = reinterpret_cast<double*>(scratch);
    hipLaunchKernelGGL(k_count_dones, dim3(1), dim3(1024), 0, r->ctx->stream, r->d_field[SG_F_MASKS],
                       (int64_t)(r->T + 1) * r->N, d_out);
    SG_CHECK(hipGetLastError());
    if (r->ctx->use_comm) SG_TRY(sg_comm_allreduce_f64(r->ctx, d_out, 1));
    SG_TRY(sg_ctx_fetch_f64(r->ctx, d_out, dones, 1));
    return 0;
}

// ------------------------------------------------------------------- synthetic rollout source
__global__ void k_fill_normal(float* x, int64_t n, uint64_t seed, uint64_t stream) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = sg_normal(seed, stream, (uint64_t)i);
}
__global__ void k_fill_masks(float* x, int64_t n, float p_done, uint64_t seed, uint64_t stream) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = sg_uniform(seed, stream, (uint64_t)i) < p_done ? 0.f : 1.f;
}
__global__ void k_fill_const(float* x, int64_t n, float v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = v;
}

extern "C" int sg_rollout_fill_synthetic(sg_rollout* r, sg_policy* p, uint64_t seed, float p_done) {
    SG_REQUIRE(r && p, "sg_rollout_fill_synthetic: NULL argument");
    SG_REQUIRE(p->desc.O == r->O && p->desc.A == r->A, "sg_rollout_fill_synthetic: policy/rollout dims differ");
    sg_ctx* ctx = r->ctx;
    auto blocks = [](int64_t n) { return dim3((unsigned)((n + 255) / 256)); };
    const uint64_t s = seed * 1000003ull + (uint64_t)ctx->rank;
    r->feat_version = sg_next_feat_version();
    hipLaunchKernelGGL(k_fill_normal, blocks(r->field_count[SG_F_OBS]), dim3(256), 0, ctx->stream, r->d_field[SG_F_OBS], r->field_count[SG_F_OBS], s, 1ull);
    if (r->field_count[SG_F_OBS_FEAT])
        hipLaunchKernelGGL(k_fill_normal, blocks(r->field_count[SG_F_OBS_FEAT]), dim3(256), 0, ctx->stream, r->d_field[SG_F_OBS_FEAT], r->field_count[SG_F_OBS_FEAT], s, 2ull);
    hipLaunchKernelGGL(k_fill_normal, blocks(r->field_count[SG_F_REWARDS]), dim3(256), 0, ctx->stream, r->d_field[SG_F_REWARDS], r->field_count[SG_F_REWARDS], s, 3ull);
    hipLaunchKernelGGL(k_fill_masks, blocks(r->field_count[SG_F_MASKS]), dim3(256), 0, ctx->stream, r->d_field[SG_F_MASKS], r->field_count[SG_F_MASKS], p_done, s, 4ull);
    hipLaunchKernelGGL(k_fill_const, blocks(r->field_count[SG_F_BAD_MASKS]), dim3(256), 0, ctx->stream, r->d_field[SG_F_BAD_MASKS], r->field_count[SG_F_BAD_MASKS], 1.0f);
    SG_CHECK(hipGetLastError());
    // actions / log-probs / values from the policy itself so PPO ratios start at 1
    const int64_t TN = (int64_t)r->T * r->N;
    SG_TRY(sg_policy_forward_device(p, r->d_field[SG_F_OBS], (int)TN, 0, nullptr, s ^ 0xACull, nullptr,
                                    r->d_field[SG_F_VALUE_PREDS], r->d_field[SG_F_ACTIONS], r->d_field[SG_F_LOGP], nullptr));
    SG_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}
