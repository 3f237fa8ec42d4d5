/*
 * simgan_hip.h -- C ABI of libsimgan_hip.so: SimGAN's GAIL+PPO update path as
 * hand-written HIP kernels for gfx950 (MI355X).
 *
 * The reference (jyf588/SimGAN) has no FFI: its hot path sits behind Python
 * classes.  Each entry point below replaces the body of
 * one reference method; the Python shim in simgan_amd/ keeps the reference's
 * class/method surface and binds these symbols with ctypes.
 * a2c/ = third_party/a2c_ppo_acktr/ in the reference tree.
 *
 * Conventions
 *   - plain C types only; all tensors are float32, row-major, caller-owned HOST
 *     pointers unless a name says "dev"; indices are int64.
 *   - every function returns 0 on success, <0 on error; sg_last_error() returns
 *     a thread-local message.
 *   - handles are opaque; one sg_ctx per GPU/process; not thread-safe (the
 *     reference learner is single-threaded: a2c/main_gail_dyn_ppo.py:64).
 *   - every stochastic input is optional by pointer: pass the reference's
 *     RNG artefacts (permutations / alpha / noise) for parity, or NULL to use
 *     the library's own counter-based generator seeded by `seed`.
 *   - flat parameter vectors use torch state_dict order of the reference
 *     modules (see oracle/sg_oracle.c header for the exact order).
 */
#ifndef SIMGAN_HIP_H
#define SIMGAN_HIP_H

#include <stdint.h>

/* The library is built with -fvisibility=hidden: exactly the entry points declared here are exported
 * (tests/test_abi.py checks both directions against `nm -D`).  Test hooks and probe kernels live in a
 * separate libsimgan_hip_test.so (simgan_amd/csrc/sg_test_api.h), never in the product library. */
#define SG_API __attribute__((visibility("default")))

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sg_ctx sg_ctx;
typedef struct sg_policy sg_policy;
typedef struct sg_rollout sg_rollout;
typedef struct sg_ppo sg_ppo;
typedef struct sg_disc sg_disc;

/* ------------------------------------------------------------------ context */
SG_API const char *sg_last_error(void);
SG_API const char *sg_version(void);
/* Creates the context on HIP device `device` (one stream, scratch arenas). */
SG_API int sg_ctx_create(int device, sg_ctx **out);
SG_API int sg_ctx_destroy(sg_ctx *ctx);
SG_API int sg_ctx_synchronize(sg_ctx *ctx);
/* Device properties the bench reports: name, CU count, HBM bytes. */
SG_API int sg_ctx_device_info(sg_ctx *ctx, char *name, int name_len, int *num_cu, int64_t *hbm_bytes);

/* Data-parallel communicator over RCCL (one process per GPU, xGMI). Replaces nothing in the
 * reference (single process, a2c/main_gail_dyn_ppo.py:64); added per BASELINE.json north_star.
 * id is the 128-byte ncclUniqueId produced on rank 0 and broadcast by the host launcher. */
SG_API int sg_comm_unique_id(uint8_t id[128]);
/* An id for the LOOPBACK transport instead: `world` contexts of ONE host (threads or processes, any devices -- also all
 * on the same device) exchange through a shared-memory segment named by the id; sg_ctx_comm_init recognises the id's
 * tag.  Every world > 1 code path of the library then runs on a one-GPU box (tests, bench.py --loopback); it
 * synchronises the stream per collective and is not a performance path. */
SG_API int sg_comm_loopback_id(uint8_t id[128]);
SG_API int sg_ctx_comm_init(sg_ctx *ctx, const uint8_t id[128], int rank, int world);
/* *kind = 0 no communicator, 1 RCCL, 2 loopback; + 4 when the per-step float32 gradient all-reduces run as one kernel over
 * peer-mapped device memory instead of a library collective (SG_COMM_PEER=1 in the environment at sg_ctx_comm_init: opt-in,
 * built over either transport). */
SG_API int sg_ctx_comm_kind(sg_ctx *ctx, int *kind);
/* Collective over an initialised communicator (every rank, same argument): 1 builds the peer mesh for the per-step float32
 * gradient all-reduce (what SG_COMM_PEER=1 does at sg_ctx_comm_init), 0 returns to the base communicator.  The outcome is
 * collective: every rank leaves with the mesh or every rank gets the error.  No counterpart in the reference (single process,
 * a2c/main_gail_dyn_ppo.py:64); bench.py times both all-reduce forms with it on the first multi-GPU run. */
SG_API int sg_ctx_comm_set_peer(sg_ctx *ctx, int enable);
/* rank / world as the communicator itself reports them (ncclCommUserRank / ncclCommCount) once it exists. */
SG_API int sg_ctx_comm_info(sg_ctx *ctx, int *rank, int *world);
/* Discriminator data-parallel mode for world > 1: 0 = replicated (default: one all-gather of
 * the ranks' rows per epoch, identical full-batch steps everywhere), 1 = sharded (batch/world rows per rank and one
 * gradient all-reduce per step).  Initial value: SG_DISC_DP=sharded in the environment. */
SG_API int sg_ctx_set_disc_dp(sg_ctx *ctx, int sharded);

/* ------------------------------------------------------------------- policy */
enum { SG_POLICY_MLP = 0, SG_POLICY_SPLIT = 1 };
/* Policy(obs_shape, action_space, base_kwargs)            a2c/model.py:38-67   (kind MLP)
 * SplitPolicy(obs_shape, action_space, base_kwargs)       a2c/model_split.py:40-52 (kind SPLIT;
 * requires act_dim == 7*num_feet, a2c/model_split.py:205). Parameters start at zero. */
SG_API int sg_policy_create(sg_ctx *ctx, int kind, int obs_dim, int act_dim, int hidden, int num_feet,
                     sg_policy **out);
/* The same with a critic trunk of its own width (critic_hidden; 0 = hidden): Policy.reset_critic(obs_shape) rebuilds a
 * 64-unit critic beside an actor of ANY hidden size (a2c/model.py:80-87; a2c/main.py:85 calls it on every warm start).
 * state_dict order and entry points are unchanged; base.critic.* and base.critic_linear take the critic's width. */
SG_API int sg_policy_create2(sg_ctx *ctx, int kind, int obs_dim, int act_dim, int hidden, int num_feet, int critic_hidden,
                             sg_policy **out);
SG_API int sg_policy_destroy(sg_policy *p);
SG_API int sg_policy_num_params(const sg_policy *p, int64_t *n);
/* nn.Module.load_state_dict / state_dict, flattened. */
SG_API int sg_policy_set_params(sg_policy *p, const float *flat, int64_t n);
SG_API int sg_policy_get_params(sg_policy *p, float *flat, int64_t n);
/* Policy.act a2c/model.py:89-101, SplitPolicy.act a2c/model_split.py:70-82.
 * obs[n,O]; noise[n,A] standard normal draws or NULL (-> library RNG with `seed`);
 * deterministic != 0 -> dist.mode().  Outputs value[n], action[n,A], logp[n]. */
SG_API int sg_policy_act(sg_policy *p, const float *obs, int n, const float *noise, uint64_t seed,
                  int deterministic, float *value, float *action, float *logp);
/* Policy.get_value a2c/model.py:103-105 */
SG_API int sg_policy_get_value(sg_policy *p, const float *obs, int n, float *value);
/* Policy.evaluate_actions a2c/model.py:107-114: value[n], logp[n], *entropy = dist.entropy().mean() */
SG_API int sg_policy_evaluate(sg_policy *p, const float *obs, const float *action, int n, float *value,
                       float *logp, float *entropy);
/* The policies that live INSIDE the reference's hybrid-sim environments, batched over the N environments of a
 * pool: every worker calls a batch-1 actor_critic.act per step, in refinement mode on one of five saved dynamics
 * policies drawn per step (my_pybullet_envs/hopper_env_combined_policy.py:113-140,211-216,
 * laikago_env_combined_policy.py:126-153,263-268).  ONE launch: row i of obs[n,O] goes through
 * policies[idx[i]] (0 <= idx[i] < n_policies <= SG_ENSEMBLE_MAX, all members the same kind and shape, weights
 * resident).  noise / seed / deterministic and the outputs are as in sg_policy_act. */
#define SG_ENSEMBLE_MAX 8
SG_API int sg_policy_act_ensemble(sg_policy *const *policies, int n_policies, const int32_t *idx, const float *obs,
                           int n, const float *noise, uint64_t seed, int deterministic, float *value,
                           float *action, float *logp);

/* ------------------------------------------------------------------ rollout */
/* RolloutStorage(num_steps, num_processes, obs_shape, action_space, rhs, feat_len) a2c/storage.py:32-56.
 * Device-resident buffers, same shapes and row-major (t, n, ·) order as the reference. */
enum {
    SG_F_OBS = 0,          /* [T+1, N, O] */
    SG_F_OBS_FEAT = 1,     /* [T+1, N, F] */
    SG_F_ACTIONS = 2,      /* [T,   N, A] */
    SG_F_REWARDS = 3,      /* [T,   N]    */
    SG_F_VALUE_PREDS = 4,  /* [T+1, N]    */
    SG_F_RETURNS = 5,      /* [T+1, N]    */
    SG_F_LOGP = 6,         /* [T,   N]  action_log_probs */
    SG_F_MASKS = 7,        /* [T+1, N]    */
    SG_F_BAD_MASKS = 8,    /* [T+1, N]    */
    SG_F_ADVANTAGES = 9,   /* [T,   N]  normalised advantages of the last sg_ppo_update (read-only) */
    SG_F_COUNT = 10
};
SG_API int sg_rollout_create(sg_ctx *ctx, int T, int N, int obs_dim, int act_dim, int feat_dim,
                      sg_rollout **out);
SG_API int sg_rollout_destroy(sg_rollout *r);
/* Whole-field host<->device copies; count = number of floats (must equal the field size). */
SG_API int sg_rollout_upload(sg_rollout *r, int field, const float *host, int64_t count);
SG_API int sg_rollout_download(sg_rollout *r, int field, float *host, int64_t count);
/* One time-slot [t] of a field (N * width floats): the reference's rollouts.obs[step].copy_(..). */
SG_API int sg_rollout_upload_step(sg_rollout *r, int field, int t, const float *host, int64_t count);
SG_API int sg_rollout_download_step(sg_rollout *r, int field, int t, float *host, int64_t count);
/* RolloutStorage.after_update a2c/storage.py:96-101: slot T -> slot 0 for obs, obs_feat, masks, bad_masks */
SG_API int sg_rollout_after_update(sg_rollout *r);
/* RolloutStorage.compute_returns a2c/storage.py:103-142 (all four branches); next_value[N] host. */
SG_API int sg_rollout_compute_returns(sg_rollout *r, const float *next_value, int use_gae, float gamma,
                               float gae_lambda, int use_proper_time_limits);
/* Same, with next_value = policy.get_value(obs[T]) computed on device (a2c/main_gail_dyn_ppo.py:239-242,299). */
SG_API int sg_rollout_compute_returns_policy(sg_rollout *r, sg_policy *p, int use_gae, float gamma,
                                      float gae_lambda, int use_proper_time_limits);
/* Synthetic rollout fill on device for benchmarks: obs, obs_feat ~ N(0,1);
 * actions/logp/value_preds from policy.act; masks ~ Bernoulli(1-p_done); bad_masks = 1. */
SG_API int sg_rollout_fill_synthetic(sg_rollout *r, sg_policy *p, uint64_t seed, float p_done);

/* ---------------------------------------------------------------------- PPO */
typedef struct {
    float clip_param;
    int ppo_epoch;
    int num_mini_batch;
    float value_loss_coef;
    float entropy_coef;
    float lr;
    float eps;
    float max_grad_norm;
    int use_clipped_value_loss;
} sg_ppo_config;
/* PPO(actor_critic, clip_param, ppo_epoch, num_mini_batch, value_loss_coef, entropy_coef, lr, eps,
 *     max_grad_norm, use_clipped_value_loss)  a2c/algo/ppo.py:30-63 (Adam state starts at zero). */
SG_API int sg_ppo_create(sg_ctx *ctx, sg_policy *p, const sg_ppo_config *cfg, sg_ppo **out);
SG_API int sg_ppo_destroy(sg_ppo *a);
/* optimizer.param_groups[i]['lr'] = lr  (a2c/utils.py:68-72 update_linear_schedule) */
SG_API int sg_ppo_set_lr(sg_ppo *a, float lr);
/* PPO.update(rollouts) a2c/algo/ppo.py:65-157 -> out3 = {value_loss, action_loss, dist_entropy}.
 * perms: [ppo_epoch][T*N] int64 = the permutation each epoch's sampler draws (a2c/storage.py:159-162) with n_perms =
 * its element count (checked, as is every index), or NULL (n_perms ignored) -> device-generated from `seed`.
 * With a communicator of world > 1 an injected permutation is the reference's draw at num_processes = world * N:
 * [ppo_epoch][T*N*world] ids in its numbering t*(N*world) + rank*N + n, the same array on every rank; each rank takes
 * the rows of every minibatch it owns (NULL: every rank permutes its own rows and gives T*N/num_mini_batch per step). */
SG_API int sg_ppo_update(sg_ppo *a, sg_rollout *r, const int64_t *perms, int64_t n_perms, uint64_t seed, float out3[3]);
/* out3 may be NULL (as for sg_disc_update_gail_dyn): the update is queued and the call returns without waiting for it.
 * An outer iteration made of such calls -- sg_disc_update_gail_dyn x gail_epoch, sg_disc_relabel_rewards_auto,
 * sg_rollout_compute_returns_policy, sg_ppo_update, sg_rollout_after_update -- never makes the host wait; its scalars (the
 * reference main's log line, a2c/main_gail_dyn_ppo.py:322-338) are published into one of SG_RESULT_SLOTS = 8 pinned
 * host slots by sg_results_publish (d and/or a may be NULL) and read with sg_results_fetch when the caller wants them:
 * out13 = {D loss sums x3 of the last epoch, ret_rms mean / var / count, sum(1 - masks), r_sa, PPO loss sums x3,
 * n_d, ppo_epoch * num_mini_batch}; losses = sums / their step count (float32, as the synchronous calls return them). */
SG_API int sg_results_publish(sg_ctx *ctx, sg_disc *d, sg_ppo *a, int slot);
SG_API int sg_results_fetch(sg_ctx *ctx, int slot, double out13[13]);
/* Adam state access for checkpoint/parity: m, v flat [n] in state_dict order; *step = t. */
/* The permutations the last sg_ppo_update consumed ([ppo_epoch][T*N], injected or library-drawn), so a run made
 * with the library's generator can be replayed elsewhere (the role torch.manual_seed plays for the reference). */
SG_API int sg_ppo_last_perms(sg_ppo *a, int64_t *perms, int64_t count);
SG_API int sg_ppo_get_adam(sg_ppo *a, float *m, float *v, int64_t n, int64_t *step);
SG_API int sg_ppo_set_adam(sg_ppo *a, const float *m, const float *v, int64_t n, int64_t step);

/* ------------------------------------------------------------ discriminator */
/* Discriminator(input_dim, hidden_dim, device) a2c/algo/gail.py:35-51; Adam(lr 1e-3, eps 1e-8). */
SG_API int sg_disc_create(sg_ctx *ctx, int input_dim, int hidden_dim, sg_disc **out);
SG_API int sg_disc_destroy(sg_disc *d);
SG_API int sg_disc_num_params(const sg_disc *d, int64_t *n);
SG_API int sg_disc_set_params(sg_disc *d, const float *flat, int64_t n);
SG_API int sg_disc_get_params(sg_disc *d, float *flat, int64_t n);
SG_API int sg_disc_get_adam(sg_disc *d, float *m, float *v, int64_t n, int64_t *step);
SG_API int sg_disc_set_adam(sg_disc *d, const float *m, const float *v, int64_t n, int64_t step);
/* The expert matrix [n_rows, input_dim] (a2c/main_gail_dyn_ppo.py:163-165) stays resident in HBM. */
SG_API int sg_disc_set_expert(sg_disc *d, const float *expert, int64_t n_rows);
/* Discriminator.update_gail_dyn(expert_loader, rollouts) a2c/algo/gail.py:154-193, one epoch.
 * batch_size = expert_loader.batch_size; n_d = min(n_expert/batch, T*N/batch) steps.
 * expert_perm[n_expert] (DataLoader shuffle), policy_perm[T*N] (feed_forward_generator),
 * alpha[n_d*batch] (torch.rand per step, a2c/algo/gail.py:72) -- each may be NULL -> `seed`; each comes with its
 * element count (n_expert_perm == n_expert, n_policy_perm == rows the permutation ranges over, n_alpha >= n_d*batch;
 * ignored for a NULL pointer), and lengths and index ranges are checked before anything is copied to the device.
 * With a communicator of world > 1 the injected policy_perm is the reference's draw at num_processes = world*N
 * ([T*N*world] ids in its numbering t*(N*world) + rank*N + n), the same arrays on every rank, in both data-parallel
 * modes; n_d = min(n_expert/batch, T*N*world/batch).  Sharded mode takes the three arrays together or none.
 * out3 = {mean(gail_loss+grad_pen), mean expert_loss, mean policy_loss}; *n_steps = n_d.  out3 may be NULL when the caller
 * has no use for this epoch's losses (the reference's main keeps only the last epoch's, a2c/main_gail_dyn_ppo.py:255-256):
 * the call then returns as soon as the epoch is queued instead of waiting for it.
 * n_expert < batch_size is an error (the reference raises on the size mismatch). */
SG_API int sg_disc_update_gail_dyn(sg_disc *d, sg_rollout *r, int batch_size, const int64_t *expert_perm, int64_t n_expert_perm,
                            const int64_t *policy_perm, int64_t n_policy_perm, const float *alpha, int64_t n_alpha,
                            uint64_t seed, float out3[3], int *n_steps);
/* Discriminator.update(expert_loader, rollouts, obsfilt, is_gail_dyn, a_dim) a2c/algo/gail.py:91-152, one
 * epoch: the same step on caller-assembled policy rows [n_rows, input_dim] (host): (state | action)
 * rows for classic GAIL, (obs_feat | action | next_obs_feat) for is_gail_dyn (a2c/algo/gail.py:102-109).
 * The expert matrix (sg_disc_set_expert) is assembled the same way.  Other arguments as above;
 * policy_perm ranges over n_rows (x world with a communicator).  n_cols = environment columns per time slot when the
 * rows are a rollout's (t, n) grid (needed to number the union of the ranks' rows as the reference would at
 * num_processes = world*N), 0 = unstructured rows (the union is numbered rank after rank). */
SG_API int sg_disc_update_rows(sg_disc *d, const float *policy_rows, int64_t n_rows, int n_cols, int batch_size,
                        const int64_t *expert_perm, int64_t n_expert_perm, const int64_t *policy_perm, int64_t n_policy_perm,
                        const float *alpha, int64_t n_alpha, uint64_t seed, float out3[3], int *n_steps);
/* Discriminator.predict_reward_combined(d_in, gamma, masks, offset) a2c/algo/gail.py:201-210.
 * x[n,F], masks[n] -> reward[n], returns[n]; Discriminator.returns persists inside the handle
 * (first call: returns = reward).  n must stay the same across calls. */
SG_API int sg_disc_predict_reward(sg_disc *d, const float *x, int n, float gamma, const float *masks,
                           float offset, float *reward, float *returns);
/* Discriminator.predict_prob_single_step(s, a, s_n) a2c/algo/gail.py:212-217: prob[n] = sigmoid(D(x)) on the
 * caller-concatenated rows x[n,F]. */
/* The T consecutive predict_reward_combined calls of the reference main's relabel loop (a2c/main_gail_dyn_ppo.py:275-280:
 * call t passes obs_feat[t + 1], masks[t]) in ONE pass over the rollout's device copy of obs_feat and masks: reward[t*N + n] and
 * returns[t*N + n] are bit for bit what the t-th sg_disc_predict_reward call returns, Discriminator.returns (a2c/algo/gail.py:206-209)
 * is left where the T-th call leaves it.  The host mirror (simgan_amd/algo/gail.py) uses it to serve the unchanged main's 128 calls
 * from one launch; the rollout's fields must be current on the device (sg_rollout_upload). */
SG_API int sg_disc_predict_reward_steps(sg_disc *d, sg_rollout *r, float gamma, float offset, float *reward, float *returns);
SG_API int sg_disc_predict_prob(sg_disc *d, const float *x, int n, float *prob);
/* Discriminator.compute_grad_pen_combined(expert_combined, policy_combined, lambda_) a2c/algo/gail.py:67-89 (and
 * compute_grad_pen :53-65 on caller-concatenated rows), the VALUE only: pen[i] = (||dD/dx(alpha_i e_i + (1 - alpha_i) p_i)||_2 - 1)^2
 * for the n row pairs expert_rows[n,F] / policy_rows[n,F]; the reference's scalar is lambda_ * mean(pen).  alpha[n] is the
 * reference's torch.rand(n, 1) draw (:72), NULL -> the library's generator with `seed`.  The differentiable term the reference adds
 * to its loss is formed inside the update kernels (sg_disc_update_gail_dyn / _rows); this call changes no state. */
SG_API int sg_disc_grad_pen(sg_disc *d, const float *expert_rows, const float *policy_rows, const float *alpha, int n,
                     uint64_t seed, float *pen);
/* The draws the last update epoch consumed: expert_perm[n_expert], policy_perm[n_policy_rows], alpha[n_alpha]
 * (injected or library-drawn); each pointer may be NULL, each count must match the epoch's size (returned in
 * counts3 = {n_expert, n_policy_rows, n_alpha} when counts3 != NULL). */
SG_API int sg_disc_last_draws(sg_disc *d, int64_t *expert_perm, int64_t *policy_perm, float *alpha, int64_t counts3[3]);
SG_API int sg_disc_reset_returns(sg_disc *d);
SG_API int sg_disc_get_returns(sg_disc *d, float *returns, int n, int *is_none);
SG_API int sg_disc_set_returns(sg_disc *d, const float *returns, int n);
/* Fused reward relabel, all T steps on device: a2c/main_gail_dyn_ppo.py:275-292
 *   for t: rewards[t], ret = predict_reward_combined(obs_feat[t+1], gamma, masks[t], offset)
 *          ret_rms.update(ret); rewards[t] = clip(rewards[t]/sqrt(ret_rms.var+1e-7), -10, 10)
 * rms_state = {mean, var, count} float64 (RunningMeanStd, a2c/baselines/common/running_mean_std.py:27-58),
 * updated in place.  Writes rollout field REWARDS. */
SG_API int sg_disc_relabel_rewards(sg_disc *d, sg_rollout *r, float gamma, float offset,
                            double rms_state[3]);
/* The same with the alive-bonus offset computed on the device as well (a2c/main_gail_dyn_ppo.py:258-271: r_sa from
 * sum(1 - masks), num_processes (all ranks), num_steps and gail_tar_length; no_alive_bonus != 0 -> 0) and ret_rms kept inside
 * the handle (initially RunningMeanStd(): {0, 1, 1e-4}): nothing is read back, the call only queues work.
 * sg_disc_set_rms overwrites the resident state; sg_disc_get_scalars reads {mean, var, count, sum(1 - masks), r_sa}. */
SG_API int sg_disc_relabel_rewards_auto(sg_disc *d, sg_rollout *r, float gamma, double gail_tar_length, int no_alive_bonus);
SG_API int sg_disc_set_rms(sg_disc *d, const double rms_state[3]);
SG_API int sg_disc_get_scalars(sg_disc *d, double out5[5]);
/* sum(1 - masks) over all T+1 slots (a2c/main_gail_dyn_ppo.py:258), for the alive-bonus offset. */
SG_API int sg_rollout_count_dones(sg_rollout *r, double *dones);

/* Page-locked host memory for buffers that cross the boundary every update (the rollout's host tensors: a2c/storage.py:37-53
 * allocates them with torch.zeros and `.to(device)` moves them, :57-68; here they stay on the host and sg_rollout_upload /
 * _download move their content).  From such memory a field crosses PCIe as one DMA at the link's speed; from pageable
 * memory the same calls work, through the runtime's staging path.  No context needed. */
SG_API int sg_host_alloc(int64_t bytes, void **out);
SG_API int sg_host_free(void *p);

/* ------------------------------------------------------------- measurement */
/* HIP-event timing of the dominant kernels on the library's stream since the last reset:
 * which = 0 k_disc_chain (discriminator step, serial part), 1 k_disc_wgrad (weight gradients + Adam),
 * 2 k_ppo_fwd, 3 k_ppo_bwd, 4 k_ppo_reduce, 5 relabel forward, 6 k_ppo_adam (clip + Adam), 7 k_disc_step4 (the one-launch
 * discriminator step), 8 the per-step float32 gradient all-reduce (N > 1: RCCL, loopback or the peer mesh, whichever runs).
 * Enabled by sg_ctx_profile(ctx, 1); adds start/stop events to every launch (do not enable inside
 * the timed bench region). */
SG_API int sg_ctx_profile(sg_ctx *ctx, int enable);
SG_API int sg_ctx_profile_read(sg_ctx *ctx, int which, double *total_ms, int64_t *launches);
SG_API int sg_ctx_profile_reset(sg_ctx *ctx);
/* A timestamp on the library's stream, taken by the device when it gets there (one HIP event; the host does not wait):
 * *id = the mark's number.  bench.py brackets every timed update with one and reads the per-update spread after the loop --
 * the reference times its updates with time.time() around the same calls (a2c/main_gail_dyn_ppo.py:318-321).
 * id = NULL: synchronise and forget every mark. */
SG_API int sg_ctx_mark(sg_ctx *ctx, int *id);
/* Milliseconds between two marks (waits for mark `to`). */
SG_API int sg_ctx_mark_elapsed(sg_ctx *ctx, int from, int to, double *ms);

#ifdef __cplusplus
}
#endif
#endif /* SIMGAN_HIP_H */
