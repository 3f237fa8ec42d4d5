// sg_ppo.hip -- PPO.update on gfx950: advantage normalisation, then per optimizer step the
// clipped-surrogate / clipped-value / entropy loss, its gradient, global-norm clipping and Adam.
//
// Replaces (reference, a2c/ = third_party/a2c_ppo_acktr/): PPO.__init__/update a2c/algo/ppo.py:29-157,
// the minibatch gather of RolloutStorage.feed_forward_generator a2c/storage.py:144-192 and
// nn.utils.clip_grad_norm_ + optim.Adam (a2c/algo/ppo.py:143-145).  Kernels: sg_ppo_kernels.hpp.
//
// Launch sequence of one update (all on the library's stream, one host synchronisation at the end):
//   k_adv_stats x3                                   advantages, global mean / unbiased std
//   per epoch:   k_ppo_epoch_gather                  permuted copy of the rollout
//   per step:    k_ppo_fwd -> k_ppo_bwd -> k_ppo_reduce [-> all-reduce -> k_sumsq] -> k_ppo_adam
//   then         k_opt_commit                        Adam's step base += E*M
// Everything from the first epoch gather on is captured into a hipGraph once and replayed per update (single GPU;
// with a communicator the RCCL calls are issued directly between the kernels).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "sg_common.h"
#include "sg_rng.hpp"
#include "sg_ppo_kernels.hpp"

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

// ---------------------------------------------------------------------------------- launch
static int max_trunk_floats(const SgPolicyDesc& d) {
    int m = 0;
    for (int t = 0; t < d.n_trunks; ++t) m = d.trunk[t].size > m ? d.trunk[t].size : m;
    return m;
}
static int max_bwd_floats(const SgPolicyDesc& d) {
    int m = 0;
    for (int t = 0; t < d.n_trunks; ++t) { const int s = d.trunk[t].size - d.trunk[t].w2; m = s > m ? s : m; }
    return m;
}
static int stack_ldP(const SgPolicyDesc& d) {
    int m = 0;
    for (int t = 0; t < d.n_trunks; ++t) m = d.trunk[t].ldP > m ? d.trunk[t].ldP : m;
    return m;
}
// gw: the global-weight instances (trunks larger than a CU's LDS, sg_policy_needs_gw): LDS holds the row tiles only
static size_t ppo_fwd_lds(const SgPolicyDesc& d, int MT, bool gw) {
    const int R = 16 * MT;
    return sizeof(float) * ((gw ? 0 : (size_t)max_trunk_floats(d)) + R * d.ldO + 2 * R * d.ldH);
}
static bool ppo_fused(const SgPolicyDesc& d, int MT) {
    const char* e = getenv("SG_PPO_FUSED");
    return d.kind == SG_POLICY_MLP && MT <= 2 && !(e && !strcmp(e, "0"));
}
static size_t ppo_bwd_lds(const SgPolicyDesc& d, int MT, bool gw) {
    const int R = 16 * MT;
    return sizeof(float) * ((gw ? 0 : (size_t)(ppo_fused(d, MT) ? max_trunk_floats(d) : max_bwd_floats(d))) + R * d.ldO + 2 * R * d.ldH + 2 * R * stack_ldP(d) +
                            ((R * d.A + 3) & ~3) + 7 * R);
}

// shape-specialised instances for the shipped configurations at the row-group
// size the launch heuristic picks for them, plus run-time-shape fallbacks
#define SG_PPO_SHAPES(X) X(1, 3, 4) X(2, 3, 4) /* north-star: obs 47, h64 */ X(2, 1, 7) /* HopperCombined: obs 14, h100 */ \
                         X(2, 4, 7) /* LaikagoCombined: obs 64, h100 */ X(2, 7, 4) /* Laikago refinement: obs 111, h64 */

// 8 waves once a workgroup has two row tiles to deal out (SG_PPO_WAVES=4: tuning knob, 4 waves always)
static int ppo_block_threads(int MT) {
    const char* we = getenv("SG_PPO_WAVES");
    return (MT >= 2 && !(we && atoi(we) == 4)) ? 512 : 256;
}

// (the shape-specialised instances fold ONE hidden width into the code: a policy whose critic was rebuilt at another width
// -- Policy.reset_critic, d.Hc != d.H -- takes the run-time-shape instances, kh = 0 matches none of them)
static void launch_ppo_fwd(sg_ctx* ctx, int MT, const SgPolicyDesc& d, dim3 grid, size_t lds, const PpoArgs& pa, bool gw) {
    const int ko = d.Op / 16, kh = d.Hc == d.H ? d.Hp / 16 : 0;
    const dim3 block(ppo_block_threads(MT));
    if (gw) {   // general-shape instances: run-time extents, weights through L2
        if (MT == 2) SG_LAUNCH(ctx, SG_PROF_PPO_FWD, (k_ppo_fwd<2, 0, 0, true>), grid, block, lds, pa);
        else SG_LAUNCH(ctx, SG_PROF_PPO_FWD, (k_ppo_fwd<1, 0, 0, true>), grid, block, lds, pa);
        return;
    }
#define SG_CASE(mt, o, h) \
    if (MT == mt && ko == o && kh == h) { SG_LAUNCH(ctx, SG_PROF_PPO_FWD, (k_ppo_fwd<mt, o, h>), grid, block, lds, pa); return; }
    SG_PPO_SHAPES(SG_CASE)
#undef SG_CASE
    if (MT == 4) SG_LAUNCH(ctx, SG_PROF_PPO_FWD, (k_ppo_fwd<4, 0, 0>), grid, block, lds, pa);
    else if (MT == 2) SG_LAUNCH(ctx, SG_PROF_PPO_FWD, (k_ppo_fwd<2, 0, 0>), grid, block, lds, pa);
    else SG_LAUNCH(ctx, SG_PROF_PPO_FWD, (k_ppo_fwd<1, 0, 0>), grid, block, lds, pa);
}
static void launch_ppo_fwd_critic(sg_ctx* ctx, int MT, const SgPolicyDesc& d, dim3 grid, size_t lds, const PpoArgs& pa) {
    const int ko = d.Op / 16, kh = d.Hc == d.H ? d.Hp / 16 : 0;
    const dim3 block(ppo_block_threads(MT));
    if (MT == 2 && ko == 4 && kh == 7) { SG_LAUNCH(ctx, SG_PROF_PPO_FWD, (k_ppo_fwd_critic<2, 4, 7>), grid, block, lds, pa); return; }   // LaikagoCombined
    if (MT == 2 && ko == 1 && kh == 7) { SG_LAUNCH(ctx, SG_PROF_PPO_FWD, (k_ppo_fwd_critic<2, 1, 7>), grid, block, lds, pa); return; }   // HopperCombined
    if (MT == 2) SG_LAUNCH(ctx, SG_PROF_PPO_FWD, (k_ppo_fwd_critic<2, 0, 0>), grid, block, lds, pa);
    else SG_LAUNCH(ctx, SG_PROF_PPO_FWD, (k_ppo_fwd_critic<1, 0, 0>), grid, block, lds, pa);
}
// SplitPolicy, one launch per step (k_ppo_pair): grid = 3 G workgroups, trunk index fastest
static void launch_ppo_pair(sg_ctx* ctx, int MT, const SgPolicyDesc& d, int G, size_t lds, const PpoArgs& pa) {
    const int ko = d.Op / 16, kh = d.Hc == d.H ? d.Hp / 16 : 0;
    const dim3 block(ppo_block_threads(MT)), grid(3 * G);
    if (MT == 2 && ko == 1 && kh == 7) { SG_LAUNCH(ctx, SG_PROF_PPO_BWD, (k_ppo_pair<2, 1, 7>), grid, block, lds, pa); return; }   // HopperCombined
    if (MT == 2 && ko == 4 && kh == 7) { SG_LAUNCH(ctx, SG_PROF_PPO_BWD, (k_ppo_pair<2, 4, 7>), grid, block, lds, pa); return; }   // LaikagoCombined at <= 2720-row minibatches
    if (MT == 2) SG_LAUNCH(ctx, SG_PROF_PPO_BWD, (k_ppo_pair<2, 0, 0>), grid, block, lds, pa);
    else SG_LAUNCH(ctx, SG_PROF_PPO_BWD, (k_ppo_pair<1, 0, 0>), grid, block, lds, pa);
}
static void launch_ppo_bwd(sg_ctx* ctx, int MT, const SgPolicyDesc& d, dim3 grid, size_t lds, const PpoArgs& pa, bool fused, bool gw) {
    const int ko = d.Op / 16, kh = d.Hc == d.H ? d.Hp / 16 : 0;
    const dim3 block(ppo_block_threads(MT));
    if (gw) {
        if (fused && MT == 2) SG_LAUNCH(ctx, SG_PROF_PPO_BWD, (k_ppo_bwd<2, 0, 0, true, true>), grid, block, lds, pa);
        else if (fused) SG_LAUNCH(ctx, SG_PROF_PPO_BWD, (k_ppo_bwd<1, 0, 0, true, true>), grid, block, lds, pa);
        else if (MT == 2) SG_LAUNCH(ctx, SG_PROF_PPO_BWD, (k_ppo_bwd<2, 0, 0, false, true>), grid, block, lds, pa);
        else SG_LAUNCH(ctx, SG_PROF_PPO_BWD, (k_ppo_bwd<1, 0, 0, false, true>), grid, block, lds, pa);
        return;
    }
    if (fused) {   // Policy (independent actor / critic trunks): forward recomputed inside, no k_ppo_fwd launch
        if (MT == 1 && ko == 3 && kh == 4) { SG_LAUNCH(ctx, SG_PROF_PPO_BWD, (k_ppo_bwd<1, 3, 4, true>), grid, block, lds, pa); return; }
        if (MT == 2 && ko == 3 && kh == 4) { SG_LAUNCH(ctx, SG_PROF_PPO_BWD, (k_ppo_bwd<2, 3, 4, true>), grid, block, lds, pa); return; }
        if (MT == 2 && ko == 7 && kh == 4) { SG_LAUNCH(ctx, SG_PROF_PPO_BWD, (k_ppo_bwd<2, 7, 4, true>), grid, block, lds, pa); return; }   // Laikago refinement: obs 111, h64
        if (MT == 2) SG_LAUNCH(ctx, SG_PROF_PPO_BWD, (k_ppo_bwd<2, 0, 0, true>), grid, block, lds, pa);
        else SG_LAUNCH(ctx, SG_PROF_PPO_BWD, (k_ppo_bwd<1, 0, 0, true>), grid, block, lds, pa);
        return;
    }
#define SG_CASE(mt, o, h) \
    if (MT == mt && ko == o && kh == h) { SG_LAUNCH(ctx, SG_PROF_PPO_BWD, (k_ppo_bwd<mt, o, h>), grid, block, lds, pa); return; }
    SG_PPO_SHAPES(SG_CASE)
#undef SG_CASE
    if (MT == 4) SG_LAUNCH(ctx, SG_PROF_PPO_BWD, (k_ppo_bwd<4, 0, 0>), grid, block, lds, pa);
    else if (MT == 2) SG_LAUNCH(ctx, SG_PROF_PPO_BWD, (k_ppo_bwd<2, 0, 0>), grid, block, lds, pa);
    else SG_LAUNCH(ctx, SG_PROF_PPO_BWD, (k_ppo_bwd<1, 0, 0>), grid, block, lds, pa);
}

// ---------------------------------------------------------------------------------- PPO API
extern "C" int sg_ppo_create(sg_ctx* ctx, sg_policy* p, const sg_ppo_config* cfg, sg_ppo** out) {
    SG_DEVICE_WIDE();
    SG_REQUIRE(ctx && p && cfg && out, "sg_ppo_create: NULL argument");
    SG_REQUIRE(cfg->ppo_epoch > 0 && cfg->num_mini_batch > 0, "sg_ppo_create: ppo_epoch and num_mini_batch must be positive");
    // a policy whose trunk does not fit a CU's LDS runs on the global-weight instances; only the 16-row tiles must fit
    SG_REQUIRE(ppo_fwd_lds(p->desc, 1, true) <= (size_t)ctx->lds_bytes && ppo_bwd_lds(p->desc, 1, true) <= (size_t)ctx->lds_bytes,
               "sg_ppo_create: the 16-row activation tiles of this policy do not fit LDS (%zu / %zu > %d bytes)",
               ppo_fwd_lds(p->desc, 1, true), ppo_bwd_lds(p->desc, 1, true), ctx->lds_bytes);
    SG_CHECK(hipSetDevice(ctx->device));
    sg_ppo* a = new sg_ppo();
    a->ctx = ctx; a->policy = p; a->cfg = *cfg;
    const size_t tot = (size_t)p->desc.total + 8;
    SG_CHECK(sg_dev_malloc((void**)&a->d_m, sizeof(float) * tot));
    SG_CHECK(sg_dev_malloc((void**)&a->d_v, sizeof(float) * tot));
    SG_CHECK(sg_dev_malloc((void**)&a->d_grad, sizeof(float) * tot));
    SG_CHECK(sg_dev_malloc((void**)&a->d_state, sizeof(SgOptState)));
    SG_CHECK(sg_dev_malloc((void**)&a->d_loss_acc, sizeof(double) * 8));
    SG_CHECK(sg_dev_malloc((void**)&a->d_part, sizeof(float) * ((tot + 8 + SG_PPO_REDUCE_PARAMS - 1) / SG_PPO_REDUCE_PARAMS + 8)));
    SG_CHECK(hipMemsetAsync(a->d_m, 0, sizeof(float) * tot, ctx->stream));
    SG_CHECK(hipMemsetAsync(a->d_v, 0, sizeof(float) * tot, ctx->stream));
    SG_CHECK(hipMemsetAsync(a->d_loss_acc, 0, sizeof(double) * 8, ctx->stream));
    SG_CHECK(sg_dev_malloc((void**)&a->d_pair, SG_PAIR_BYTES));            // k_ppo_pair: error word
    SG_CHECK(hipMemsetAsync(a->d_pair, 0, SG_PAIR_BYTES, ctx->stream));
    SgOptState st;
    memset(&st, 0, sizeof st);
    st.lr = cfg->lr;
    SG_CHECK(hipMemcpyAsync(a->d_state, &st, sizeof st, hipMemcpyHostToDevice, ctx->stream));
    SG_CHECK(hipStreamSynchronize(ctx->stream));
    sg_ctx_learner_born(ctx);
    *out = a;
    return 0;
}

extern "C" int sg_ppo_destroy(sg_ppo* a) {
    SG_DEVICE_WIDE();
    if (!a) return 0;
    (void)hipStreamSynchronize(a->ctx->stream);
    sg_ctx_learner_gone(a->ctx);
    for (auto& q : a->ctx->res_a) if (q == a) q = nullptr;
    float* ptrs[] = {a->d_m, a->d_v, a->d_grad, a->d_slabs, a->d_state, a->d_part, a->d_stacks};
    for (float* q : ptrs) if (q) (void)sg_dev_free(q);
    if (a->d_perms) (void)sg_dev_free(a->d_perms);
    if (a->d_loss_acc) (void)sg_dev_free(a->d_loss_acc);
    if (a->d_dbg) (void)sg_dev_free(a->d_dbg);
    if (a->d_pair) (void)sg_dev_free(a->d_pair);
    if (a->steps_graph) (void)hipGraphExecDestroy(a->steps_graph);
    delete a;
    return 0;
}

__global__ void k_set_lr(SgOptState* st, float lr) { st->lr = lr; }

extern "C" int sg_ppo_set_lr(sg_ppo* a, float lr) {
    SG_REQUIRE(a, "sg_ppo_set_lr: NULL argument");
    SG_CHECK(hipSetDevice(a->ctx->device));
    a->cfg.lr = lr;
    // by kernel argument, in stream order: no host synchronisation (the schedule writes it before every update)
    hipLaunchKernelGGL(k_set_lr, dim3(1), dim3(1), 0, a->ctx->stream, reinterpret_cast<SgOptState*>(a->d_state), lr);
    SG_CHECK(hipGetLastError());
    return 0;
}

extern "C" int sg_ppo_last_perms(sg_ppo* a, int64_t* perms, int64_t count) {
    SG_REQUIRE(a && perms, "sg_ppo_last_perms: NULL argument");
    SG_REQUIRE(a->last_perm_count > 0, "sg_ppo_last_perms: no update has run on this agent");
    SG_REQUIRE(count == a->last_perm_count, "sg_ppo_last_perms: the last update used %lld indices, asked for %lld",
               (long long)a->last_perm_count, (long long)count);
    SG_CHECK(hipSetDevice(a->ctx->device));
    SG_CHECK(hipStreamSynchronize(a->ctx->stream));
    SG_COPY_SYNC(a->ctx, perms, a->d_perms, sizeof(int64_t) * count, hipMemcpyDeviceToHost);
    return 0;
}

extern "C" int sg_ppo_get_adam(sg_ppo* a, float* m, float* v, int64_t n, int64_t* step) {
    SG_REQUIRE(a && m && v && step, "sg_ppo_get_adam: NULL argument");
    const SgPolicyDesc& d = a->policy->desc;
    SG_REQUIRE(n == sg_policy_flat_count(d), "sg_ppo_get_adam: bad length");
    std::vector<float> pm(d.total), pv(d.total);
    SG_CHECK(hipStreamSynchronize(a->ctx->stream));
    SG_COPY_SYNC(a->ctx, pm.data(), a->d_m, sizeof(float) * d.total, hipMemcpyDeviceToHost);
    SG_COPY_SYNC(a->ctx, pv.data(), a->d_v, sizeof(float) * d.total, hipMemcpyDeviceToHost);
    sg_policy_unpad(d, pm.data(), m);
    sg_policy_unpad(d, pv.data(), v);
    *step = a->opt_t;
    return 0;
}

extern "C" int sg_ppo_set_adam(sg_ppo* a, const float* m, const float* v, int64_t n, int64_t step) {
    SG_REQUIRE(a && m && v, "sg_ppo_set_adam: NULL argument");
    const SgPolicyDesc& d = a->policy->desc;
    SG_REQUIRE(n == sg_policy_flat_count(d), "sg_ppo_set_adam: bad length");
    std::vector<float> pm(d.total, 0.f), pv(d.total, 0.f);
    sg_policy_pad(d, m, pm.data());
    sg_policy_pad(d, v, pv.data());
    SG_REQUIRE(step >= 0 && step < (1ll << 30), "sg_ppo_set_adam: step out of range");
    const int t0 = (int)step;
    SG_CHECK(hipStreamSynchronize(a->ctx->stream));
    SG_COPY_SYNC(a->ctx, a->d_m, pm.data(), sizeof(float) * d.total, hipMemcpyHostToDevice);
    SG_COPY_SYNC(a->ctx, a->d_v, pv.data(), sizeof(float) * d.total, hipMemcpyHostToDevice);
    SG_COPY_SYNC(a->ctx, &reinterpret_cast<SgOptState*>(a->d_state)->t0, &t0, sizeof t0, hipMemcpyHostToDevice);
    // the words k_ppo_pair's actor workgroups swap carry Adam step numbers: a step count set from outside may repeat old ones,
    // so the row stacks they live in are cleared before the next update
    a->scratch_key = 0;
    a->pair_primed = false;
    a->opt_t = step;
    return 0;
}

extern "C" int sg_ppo_update(sg_ppo* a, sg_rollout* r, const int64_t* perms, int64_t n_perms, uint64_t seed, float out3[3]) {
    SG_REQUIRE(a && r, "sg_ppo_update: NULL argument");
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
    const int world = ctx->world;
    // Minibatch geometry.  Single rank, or the library's own generator: every rank permutes its own TN rows and gives
    // TN / M of them to each step.  World > 1 with INJECTED permutations ("owned" mode, the parity form of SURVEY.md 8(e)):
    // the permutations are the reference's own draws at num_processes = world * N -- [E][TN * world] ids in its numbering
    // t * (N * world) + rank * N + n (a2c/storage.py:159-185) -- and rank r takes, of every minibatch, the rows it owns,
    // so the counts per step are uneven; row groups past a step's count are masked inside the kernels.
    const bool owned = perms != nullptr && ctx->use_comm && world > 1;
    const int64_t TN_perm = owned ? TN * world : TN;
    if (perms) {
        SG_REQUIRE(n_perms == (int64_t)E * TN_perm, "sg_ppo_update: perms holds %lld indices, %d epochs x %lld rows need %lld",
                   (long long)n_perms, E, (long long)TN_perm, (long long)E * TN_perm);
        for (int64_t i = 0; i < n_perms; ++i)
            SG_REQUIRE(perms[i] >= 0 && perms[i] < TN_perm, "sg_ppo_update: perms[%lld] = %lld is outside [0, %lld)", (long long)i,
                       (long long)perms[i], (long long)TN_perm);
    }
    std::vector<int> step_cnt((size_t)E * M), step_off((size_t)E * M);   // rows [off, off + cnt) of the epoch's permuted copy
    std::vector<int64_t> own_perm;                                       // owned mode: [E][TN] local row ids, compacted per epoch
    std::vector<int64_t> epoch_rows(E, TN);                              // rows the epoch's gather copies
    int mb = (int)(TN / M);
    int64_t mb_global = (int64_t)mb * (ctx->use_comm ? world : 1);
    if (owned) {
        const int64_t Ng = (int64_t)r->N * world;
        mb_global = TN_perm / M;
        own_perm.assign((size_t)E * TN, 0);
        mb = 1;
        std::vector<uint8_t> seen((size_t)TN_perm);
        for (int e = 0; e < E; ++e) {
            // every epoch's row must be a true permutation (a2c/storage.py:159-162 draws one): with repeated ids a rank could
            // be handed more than its T*N rows, past the end of own_perm and of the device's epoch copy
            std::fill(seen.begin(), seen.end(), (uint8_t)0);
            for (int64_t i = 0; i < TN_perm; ++i) {
                const int64_t g = perms[(size_t)e * TN_perm + i];
                SG_REQUIRE(!seen[(size_t)g], "sg_ppo_update: perms of epoch %d repeat row id %lld: not a permutation", e, (long long)g);
                seen[(size_t)g] = 1;
            }
            int64_t n_own = 0;
            for (int k = 0; k < M; ++k) {
                step_off[(size_t)e * M + k] = (int)n_own;
                for (int64_t i = 0; i < mb_global; ++i) {
                    const int64_t g = perms[(size_t)e * TN_perm + (size_t)k * mb_global + i];
                    const int64_t t = g / Ng, c = g - t * Ng, owner = c / r->N;
                    if (owner == ctx->rank) {
                        SG_REQUIRE(n_own < TN, "sg_ppo_update: epoch %d hands this rank more than its %lld rows", e, (long long)TN);
                        own_perm[(size_t)e * TN + n_own++] = t * r->N + (c - owner * r->N);
                    }
                }
                const int cnt = (int)n_own - step_off[(size_t)e * M + k];
                step_cnt[(size_t)e * M + k] = cnt;
                if (cnt > mb) mb = cnt;
            }
            epoch_rows[e] = n_own;
        }
    } else {
        for (int e = 0; e < E; ++e)
            for (int k = 0; k < M; ++k) { step_cnt[(size_t)e * M + k] = mb; step_off[(size_t)e * M + k] = k * mb; }
    }

    // advantages (global mean / unbiased std)
    float* adv = r->d_field[SG_F_ADVANTAGES];
    double* stats = a->d_loss_acc + 4;
    hipLaunchKernelGGL(k_adv_stats, dim3(1), dim3(1024), 0, ctx->stream, r->d_field[SG_F_RETURNS],
                       r->d_field[SG_F_VALUE_PREDS], TN, adv, stats, 0);
    if (ctx->use_comm) SG_TRY(sg_comm_allreduce_f64(ctx, stats, 3));       // sum and n are linear
    hipLaunchKernelGGL(k_adv_stats, dim3(1), dim3(1024), 0, ctx->stream, r->d_field[SG_F_RETURNS],
                       r->d_field[SG_F_VALUE_PREDS], TN, adv, stats, 1);
    if (ctx->use_comm) SG_TRY(sg_comm_allreduce_f64(ctx, stats + 1, 1));   // squares about the global mean
    hipLaunchKernelGGL(k_adv_stats, dim3(64), dim3(1024), 0, ctx->stream, r->d_field[SG_F_RETURNS],
                       r->d_field[SG_F_VALUE_PREDS], TN, adv, stats, 2);
    SG_CHECK(hipGetLastError());

    // permutations
    if (a->perms_cap < (int64_t)E * TN) {
        SG_CHECK(hipStreamSynchronize(ctx->stream));
        if (a->d_perms) SG_CHECK(sg_dev_free(a->d_perms));
        SG_CHECK(sg_dev_malloc((void**)&a->d_perms, sizeof(int64_t) * (size_t)E * TN));
        a->perms_cap = (int64_t)E * TN;
    }
    a->last_perm_count = owned ? 0 : (int64_t)E * TN;   // owned mode: the device holds this rank's share only
    if (owned) {
        SG_CHECK(hipMemcpyAsync(a->d_perms, own_perm.data(), sizeof(int64_t) * (size_t)E * TN, hipMemcpyHostToDevice, ctx->stream));
        SG_CHECK(hipStreamSynchronize(ctx->stream));   // own_perm goes out of scope with this call
    } else if (perms) {
        SG_CHECK(hipMemcpyAsync(a->d_perms, perms, sizeof(int64_t) * (size_t)E * TN, hipMemcpyHostToDevice, ctx->stream));
        // a queued update (out3 == NULL) returns without a host wait: the caller's array must have been read by then
        if (!out3) SG_CHECK(hipStreamSynchronize(ctx->stream));
    } else {
        for (int e = 0; e < E; ++e)
            SG_TRY(sg_fill_perm(ctx, a->d_perms + (size_t)e * TN, TN, seed, (uint64_t)e * 2654435761ull + (uint64_t)ctx->rank));
    }

    // launch geometry: 32-row groups as long as they still give every CU a workgroup (half the gradient slabs for
    // k_ppo_reduce to stream: -1.8 us per step at the north-star shape against +0.6 us in k_ppo_bwd), or when the
    // 16-row slabs would exceed 24 MB; otherwise 16-row groups (more workgroups in flight hide the phases' latencies)
    int MT = ((size_t)((mb + 15) / 16) * (size_t)(d.total + 8) * sizeof(float) > ((size_t)24 << 20) ||
              ((mb + 31) / 32) * d.n_trunks >= ctx->num_cu) ? 2 : 1;
    if (const char* e = getenv("SG_PPO_ROWS")) {   // tuning knob
        const int v = atoi(e);
        if (v == 16 || v == 32 || v == 64) MT = v / 16;
    }
    // global-weight instances when a trunk (+ one 16-row tile) does not fit LDS: in the forward, or in the backward launch
    const bool gw = sg_policy_needs_gw(ctx, d) || ppo_fwd_lds(d, 1, false) > (size_t)ctx->lds_bytes ||
                    ppo_bwd_lds(d, 1, false) > (size_t)ctx->lds_bytes;
    if (gw && MT > 2) MT = 2;
    while (MT > 1 && (ppo_fwd_lds(d, MT, gw) > (size_t)ctx->lds_bytes || ppo_bwd_lds(d, MT, gw) > (size_t)ctx->lds_bytes)) MT /= 2;
    const int R = 16 * MT;
    const int G = (mb + R - 1) / R;
    const int mbp = G * R;
    const int ldP = stack_ldP(d);
    // 64-float (256-byte) multiple: k_ppo_reduce's waves read 64 consecutive floats of every slab, and with a stride that is
    // not a multiple of the 128-byte line each such read straddles three lines instead of two (round 3: FETCH_SIZE of the
    // reduce 1.4-1.95x the slab bytes before)
    const int slab_stride = (d.total + 8 + 63) & ~63;
    // slack rows: the last row tile may read past the last minibatch (owned mode: a short step still reads mbp rows)
    const int TNp = (int)TN + 64 + (owned ? mbp : 0);

    // scratch: slabs | epoch copy (X, ACT, SC) | per-trunk row stacks (H1, H2, OUT)
    const size_t slab_f = (size_t)G * slab_stride;
    const size_t epoch_f = (size_t)TNp * (d.ldO + d.A + 4);
    const size_t stack_f = (size_t)d.n_trunks * mbp * (2 * (size_t)d.ldH + ldP);
    if (a->slabs_cap < slab_f || a->stacks_cap < epoch_f + stack_f) {
        SG_CHECK(hipStreamSynchronize(ctx->stream));
        if (a->d_slabs) SG_CHECK(sg_dev_free(a->d_slabs));
        if (a->d_stacks) SG_CHECK(sg_dev_free(a->d_stacks));
        SG_CHECK(sg_dev_malloc((void**)&a->d_slabs, sizeof(float) * slab_f));
        SG_CHECK(sg_dev_malloc((void**)&a->d_stacks, sizeof(float) * (epoch_f + stack_f)));
        a->slabs_cap = slab_f;
        a->stacks_cap = epoch_f + stack_f;
        a->scratch_key = 0;
    }
    // ld-padding columns of the slabs are never written by the kernels and must read as zero; the epoch copy's
    // slack rows must be finite.  Both hold for as long as the scratch layout is unchanged, so the 35 MB are
    // cleared when the layout changes, not on every update.
    const uint64_t key = ((uint64_t)G << 40) ^ ((uint64_t)slab_stride << 20) ^ ((uint64_t)mbp << 8) ^ (uint64_t)TNp ^ ((uint64_t)MT << 60);
    if (a->scratch_key != key) {
        SG_CHECK(hipMemsetAsync(a->d_slabs, 0, sizeof(float) * slab_f, ctx->stream));
        SG_CHECK(hipMemsetAsync(a->d_stacks, 0, sizeof(float) * (epoch_f + stack_f), ctx->stream));
        a->scratch_key = key;
    }
    hipLaunchKernelGGL(k_zero_f64, dim3(1), dim3(64), 0, ctx->stream, a->d_loss_acc, 3);

    float* epX = a->d_stacks;
    float* epACT = epX + (size_t)TNp * d.ldO;
    float* epSC = epACT + (size_t)TNp * d.A;
    float* stk = epSC + (size_t)TNp * 4;

    EpochGatherArgs ga;
    ga.obs = r->d_field[SG_F_OBS]; ga.actions = r->d_field[SG_F_ACTIONS]; ga.old_logp = r->d_field[SG_F_LOGP];
    ga.adv = adv; ga.vpred = r->d_field[SG_F_VALUE_PREDS]; ga.ret = r->d_field[SG_F_RETURNS];
    ga.TN = TN; ga.O = d.O; ga.Op = d.Op; ga.ldO = d.ldO; ga.A = d.A; ga.sc_stride = TNp;
    ga.X = epX; ga.ACT = epACT; ga.SC = epSC;

    PpoArgs pa;
    pa.d = d; pa.params = a->policy->d_params;
    pa.sc_stride = TNp; pa.mb = mb; pa.mbp = mbp; pa.inv_B = 1.0f / (float)mb_global;
    pa.clip = a->cfg.clip_param; pa.vcoef = a->cfg.value_loss_coef; pa.ecoef = a->cfg.entropy_coef;
    pa.use_clipped = a->cfg.use_clipped_value_loss;
    pa.slabs = a->d_slabs; pa.slab_stride = slab_stride; pa.ldP = ldP; pa.dbg = a->d_dbg;
    pa.st = reinterpret_cast<SgOptState*>(a->d_state); pa.G = G; pa.k1 = 0;
    pa.pair = a->d_pair;
    for (int t = 0; t < 3; ++t) {
        const bool on = t < d.n_trunks;
        pa.H1[t] = on ? stk : nullptr; if (on) stk += (size_t)mbp * d.ldH;
        pa.H2[t] = on ? stk : nullptr; if (on) stk += (size_t)mbp * d.ldH;
        pa.OUT[t] = on ? stk : nullptr; if (on) stk += (size_t)mbp * ldP;
    }
    const int wb_f = gw ? 0 : max_trunk_floats(d), wb_b = gw ? 0 : max_bwd_floats(d);
    const bool fused = ppo_fused(d, MT);
    // SplitPolicy with more (row group, trunk) workgroups than CUs: the critic's whole fused forward + backward rides in the
    // forward launch (k_ppo_fwd_critic), the backward launch covers the two actor trunks.  SG_PPO_CRITIC_FIRST=0/1 forces it.
    const size_t lds_fc = sizeof(float) * ((size_t)wb_f + R * d.ldO + 2 * R * d.ldH + 2 * R * ldP + ((R * d.A + 3) & ~3) + 7 * R);
    const char* cfenv = getenv("SG_PPO_CRITIC_FIRST");
    const bool crit_first = !gw && !fused && d.kind == SG_POLICY_SPLIT && MT <= 2 && lds_fc <= (size_t)ctx->lds_bytes &&
                            (cfenv ? cfenv[0] == '1' : G * d.n_trunks > ctx->num_cu);
    // SplitPolicy whose 3 G (row group, trunk) workgroups are all resident at once: ONE launch per step, every trunk fused,
    // the two actor workgroups of a row group exchanging their head outputs inside it (k_ppo_pair).  SG_PPO_PAIR=0: two launches.
    const char* penv = getenv("SG_PPO_PAIR");
    const bool pair = !gw && !fused && !crit_first && d.kind == SG_POLICY_SPLIT && d.n_trunks == 3 && MT <= 2 &&
                      lds_fc <= (size_t)ctx->lds_bytes && 3 * G <= ctx->num_cu && 2 * ldP + 2 <= std::min(d.trunk[0].ldH, d.trunk[1].ldH) && !a->d_dbg &&
                      !a->self_wait_failed && (penv ? strcmp(penv, "0") != 0 : sg_ctx_exclusive(ctx));   // (=1 forces it on a shared device: tests)
    // (the tagged words live in the ACTOR trunks' H1 rows, indexed with the trunk's own ldH: d.ldH is the widest trunk's, which a
    // critic rebuilt wider than the actors -- sg_policy_create2 -- would make too generous a bound)
    // the words the actor pairs swap live in the H1 row stacks: an update that ran the two-launch step left activations there,
    // and a bit pattern must never be mistaken for a tagged word -- clear them whenever the mode is (re-)entered
    if (pair && !a->pair_primed) SG_CHECK(hipMemsetAsync(a->d_stacks, 0, sizeof(float) * (epoch_f + stack_f), ctx->stream));
    a->pair_primed = pair;
    const size_t lds_f = ppo_fwd_lds(d, MT, gw), lds_b = ppo_bwd_lds(d, MT, gw);
    const int nblk = (d.total + 8 + 255) / 256;
    const int nblk_r = (d.total + 8 + SG_PPO_REDUCE_PARAMS - 1) / SG_PPO_REDUCE_PARAMS;
    SgOptState* st = reinterpret_cast<SgOptState*>(a->d_state);

    // The E*M optimizer steps (+ one row gather per epoch) depend only on buffer addresses, the minibatch geometry
    // and the PPO coefficients: the learning rate and Adam's step count live on the device.  The sequence is
    // captured into a hipGraph once and replayed per update, so the host issues one call instead of ~650.
    auto enqueue_steps = [&]() -> int {
        hipLaunchKernelGGL(k_opt_prepare_first, dim3(1), dim3(1), 0, ctx->stream, st);
        for (int e = 0; e < E; ++e) {
            ga.perm = a->d_perms + (size_t)e * TN;
            ga.TN = epoch_rows[e];
            if (ga.TN) hipLaunchKernelGGL(k_ppo_epoch_gather, dim3((unsigned)((ga.TN + 63) / 64)), dim3(256), 0, ctx->stream, ga);
            for (int k = 0; k < M; ++k) {
                const size_t rb = (size_t)step_off[(size_t)e * M + k];
                pa.mb = step_cnt[(size_t)e * M + k];
                pa.X = epX + rb * d.ldO; pa.ACT = epACT + rb * d.A; pa.SC = epSC + rb;
                pa.k1 = e * M + k + 1;
                if (pair) {
                    pa.wbuf_floats = wb_f;
                    launch_ppo_pair(ctx, MT, d, G, lds_fc, pa);
                } else if (crit_first) {
                    pa.wbuf_floats = wb_f;
                    launch_ppo_fwd_critic(ctx, MT, d, dim3(G, d.n_trunks), lds_fc > lds_f ? lds_fc : lds_f, pa);
                    pa.wbuf_floats = wb_b;
                    launch_ppo_bwd(ctx, MT, d, dim3(G, d.n_trunks - 1), lds_b, pa, false, false);   // trunks 0, 1: the actors
                } else {
                    if (!fused) {
                        pa.wbuf_floats = wb_f;
                        launch_ppo_fwd(ctx, MT, d, dim3(G, d.n_trunks), lds_f, pa, gw);
                    }
                    pa.wbuf_floats = fused ? wb_f : wb_b;
                    launch_ppo_bwd(ctx, MT, d, dim3(G, d.n_trunks), lds_b, pa, fused, gw);
                }
                SG_LAUNCH(ctx, SG_PROF_PPO_REDUCE, k_ppo_reduce, dim3(nblk_r), dim3(256), 0, a->d_slabs, G, slab_stride,
                          d.total, a->d_grad, a->d_part);
                if (ctx->use_comm) {
                    SG_TRY(sg_comm_allreduce_f32(ctx, a->d_grad, d.total + 8));
                    hipLaunchKernelGGL(k_sumsq, dim3(nblk), dim3(256), 0, ctx->stream, a->d_grad, d.total, a->d_part);
                }
                SG_LAUNCH(ctx, SG_PROF_PPO_ADAM, k_ppo_adam, dim3(nblk), dim3(256), 0, a->policy->d_params, a->d_m, a->d_v,
                          a->d_grad, a->d_part, ctx->use_comm ? nblk : nblk_r, d.total, st, e * M + k + 1, a->cfg.eps, a->cfg.max_grad_norm,
                          pa.inv_B, a->d_loss_acc);
            }
        }
        hipLaunchKernelGGL(k_opt_commit, dim3(1), dim3(1), 0, ctx->stream, st, E * M);
        return 0;
    };
    // With a communicator the RCCL all-reduces are part of the captured sequence (RCCL enqueues them on the capturing
    // stream like any kernel), so N > 1 keeps the one-call-per-update property; SG_PPO_GRAPH_COMM=0 or a capture the
    // RCCL build refuses falls back to direct launches.
    const char* genv = getenv("SG_PPO_GRAPH");
    const char* gcenv = getenv("SG_PPO_GRAPH_COMM");
    const bool comm_ok = !a->graph_refused && (!ctx->use_comm || (sg_comm_graph_ok(ctx) && !(gcenv && !strcmp(gcenv, "0"))));
    bool use_graph = comm_ok && !owned && !ctx->profile && !a->d_dbg && !(genv && !strcmp(genv, "0"));
    if (use_graph) {
        uint32_t fbits[6];
        const float fv[6] = {a->cfg.clip_param, a->cfg.value_loss_coef, a->cfg.entropy_coef, a->cfg.eps, a->cfg.max_grad_norm, pa.inv_B};
        memcpy(fbits, fv, sizeof fbits);
        const uint64_t key[16] = {(uint64_t)(uintptr_t)a->d_slabs, (uint64_t)(uintptr_t)a->d_stacks, (uint64_t)(uintptr_t)a->d_perms,
                                  (uint64_t)(uintptr_t)r->d_field[SG_F_OBS], (uint64_t)(uintptr_t)r->d_field[SG_F_ACTIONS],
                                  (uint64_t)(uintptr_t)r->d_field[SG_F_RETURNS], (uint64_t)(uintptr_t)a->policy->d_params,
                                  (uint64_t)TN, ((uint64_t)E << 32) | (uint64_t)M, ((uint64_t)MT << 32) | (uint64_t)G,
                                  ((uint64_t)fbits[0] << 32) | fbits[1], ((uint64_t)fbits[2] << 32) | fbits[3],
                                  ((uint64_t)fbits[4] << 32) | fbits[5], (uint64_t)a->cfg.use_clipped_value_loss,
                                  (uint64_t)(uintptr_t)r->d_field[SG_F_LOGP], 0x50504full + (fused ? 1 : 0) + (ctx->use_comm ? 2 : 0) + (crit_first ? 4 : 0) + (gw ? 8 : 0) + (pair ? 16 : 0) + (sg_comm_peer_on(ctx) ? 32 : 0) + ((uint64_t)sg_comm_peer_generation(ctx) << 32)};
        if (!a->steps_graph || memcmp(key, a->steps_graph_key, sizeof key) != 0) {
            if (a->steps_graph) { SG_CHECK(hipGraphExecDestroy(a->steps_graph)); a->steps_graph = nullptr; }
            if (sg_try_capture(ctx, &a->steps_graph, enqueue_steps) != 0) {
                a->graph_refused = true;   // reported once on stderr: direct launches from now on
                use_graph = false;
            } else {
                memcpy(a->steps_graph_key, key, sizeof key);
            }
        }
        if (use_graph) SG_CHECK(hipGraphLaunch(a->steps_graph, ctx->stream));
    }
    if (!use_graph) SG_TRY(enqueue_steps());
    SG_CHECK(hipGetLastError());
    a->opt_t += (int64_t)E * M;
    if (!out3) return 0;   // the caller reads the losses later (sg_results_publish): the update stays queued, no host wait
    double acc[3];
    SG_TRY(sg_ctx_fetch_f64(ctx, a->d_loss_acc, acc, 3));
    if (pair && (acc[0] != acc[0] || acc[1] != acc[1] || acc[2] != acc[2])) {   // NaN: either the data, or a workgroup of k_ppo_pair gave up waiting
        unsigned err = 0;
        SG_COPY_SYNC(ctx, &err, a->d_pair + SG_PAIR_ERR_WORD, sizeof err, hipMemcpyDeviceToHost);
        if (err) {
            SG_CHECK(hipMemsetAsync(a->d_pair + SG_PAIR_ERR_WORD, 0, sizeof err, ctx->stream));
            a->self_wait_failed = true;
            SG_REQUIRE(false, "sg_ppo_update: a workgroup of k_ppo_pair waited %d s for its partner's words and gave "
                       "up (the policy's state is undefined; SG_PPO_PAIR=0 runs the multi-launch step)", (int)(SG_PAIR_TIMEOUT_TICKS / 100000000ll));
        }
    }
    const double nu = (double)E * M;
    for (int i = 0; i < 3; ++i) out3[i] = (float)(acc[i] / nu);
    return 0;
}

// ------------------------------------------------------------------------------- results ring
// One update's scalars, published without a host synchronisation: slot layout (doubles)
//   [0..2] discriminator loss sums of the last epoch (divide by [11])   [3..5] ret_rms mean / var / count   [6] sum(1 - masks)
//   [7] r_sa   [8..10] PPO loss sums (divide by [12])   [11] n_d   [12] ppo_epoch * num_mini_batch
__global__ void k_publish(double* dst, const double* d_acc, const double* d_scal, const double* p_acc, double n_d, double n_p,
                          const unsigned* d_err, const unsigned* a_err, const unsigned* peer_err) {
    const int t = threadIdx.x;
    if (t < 3) dst[t] = d_acc ? d_acc[t] : 0.0;
    else if (t < 8) dst[t] = d_scal ? d_scal[t - 3] : 0.0;
    else if (t < 11) dst[t] = p_acc ? p_acc[t - 8] : 0.0;
    else if (t == 11) dst[t] = n_d;
    else if (t == 12) dst[t] = n_p;
    // the sticky error words of the launches that wait inside themselves (k_disc_step4, k_ppo_pair): an update queued without
    // a host wait has nobody else to read them
    else if (t == 13) dst[t] = d_err ? (double)*d_err : 0.0;
    else if (t == 14) dst[t] = a_err ? (double)*a_err : 0.0;
    else if (t == 15) dst[t] = peer_err ? (double)__hip_atomic_load(peer_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;   // the peer mesh's (sg_comm.cpp)
}

extern "C" int sg_results_publish(sg_ctx* ctx, sg_disc* d, sg_ppo* a, int slot) {
    SG_REQUIRE(ctx && slot >= 0 && slot < SG_RESULT_SLOTS, "sg_results_publish: bad argument");
    SG_REQUIRE((!d || d->ctx == ctx) && (!a || a->ctx == ctx), "sg_results_publish: objects of another context");
    SG_CHECK(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, ctx->stream, ctx->results + 16 * slot, d ? d->d_loss_acc : nullptr,
                       d ? d->d_scal : nullptr, a ? a->d_loss_acc : nullptr, d ? (double)d->last_n_d : 1.0,
                       a ? (double)a->cfg.ppo_epoch * a->cfg.num_mini_batch : 1.0,
                       d ? sg_disc_err_word(d) : nullptr,
                       (a && a->d_pair) ? a->d_pair + SG_PAIR_ERR_WORD : nullptr, sg_comm_peer_err_word(ctx));
    SG_CHECK(hipGetLastError());
    SG_CHECK(hipEventRecord(ctx->res_ev[slot], ctx->stream));
    ctx->res_d[slot] = d;
    ctx->res_a[slot] = a;
    return 0;
}

// A hand-off time-out inside a queued update surfaces HERE (the synchronous calls report it themselves): the error is
// returned, the sticky device word is cleared and the object runs its multi-launch form from now on.
extern "C" int sg_results_fetch(sg_ctx* ctx, int slot, double out13[13]) {
    SG_REQUIRE(ctx && out13 && slot >= 0 && slot < SG_RESULT_SLOTS, "sg_results_fetch: bad argument");
    SG_CHECK(hipEventSynchronize(ctx->res_ev[slot]));
    const double* src = ctx->results + 16 * slot;
    memcpy(out13, src, sizeof(double) * 13);
    const bool d_err = src[13] != 0.0, a_err = src[14] != 0.0;
    if (src[15] != 0.0) {
        SG_CHECK(hipSetDevice(ctx->device));
        if (unsigned* w = sg_comm_peer_err_word(ctx)) SG_CHECK(hipMemsetAsync(w, 0, sizeof(unsigned), ctx->stream));
        SG_REQUIRE(false, "sg_results_fetch: rank %d: a peer-mesh all-reduce (SG_COMM_PEER) waited 20 s for a peer's flags and gave up "
                   "during or before the update this slot reports; that step's gradient was NaN on this rank", ctx->rank);
    }
    if (d_err || a_err) {
        SG_CHECK(hipSetDevice(ctx->device));
        sg_disc* d = ctx->res_d[slot];
        sg_ppo* a = ctx->res_a[slot];
        if (d_err && d) {
            d->self_wait_failed = true;
            SG_CHECK(hipMemsetAsync(sg_disc_err_word(d), 0, sizeof(unsigned), ctx->stream));
        }
        if (a_err && a && a->d_pair) {
            a->self_wait_failed = true;
            SG_CHECK(hipMemsetAsync(a->d_pair + SG_PAIR_ERR_WORD, 0, sizeof(unsigned), ctx->stream));
        }
        SG_REQUIRE(false, "sg_results_fetch: %s%s%s waited for workgroups of its own launch and gave up during the update this slot reports "
                   "(another process on this GPU?): the %s undefined and the slot's losses are NaN; later updates of the object run the "
                   "multi-launch form", d_err ? "k_disc_step4" : "", d_err && a_err ? " and " : "", a_err ? "k_ppo_pair" : "",
                   d_err && a_err ? "discriminator's and the policy's state are" : d_err ? "discriminator's state is" : "policy's state is");
    }
    return 0;
}
