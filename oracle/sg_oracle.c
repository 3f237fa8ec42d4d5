/*
 * sg_oracle.c -- CPU restatement of SimGAN's GAIL+PPO inner training loop.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle (and the timed
 * "port" CPU baseline in bench.py's cpu_baseline leg).  The product path
 * (simgan_amd/ + libsimgan_hip.so) never links, imports or calls it.
 *
 * Parity status: the reference has no tests or golden vectors for this path
 * (SURVEY.md section 4), so this restatement is pinned against outputs of the
 * reference itself, captured in the dev container by tools/gen_golden.py
 * (reference imported from /root/reference with gym/pybullet stand-in modules)
 * and committed as tests/golden/*.npz.  tests/test_oracle_golden.py checks
 * every function here against those captures.
 *
 * All arithmetic is float32 (like the reference on torch CPU) except where the
 * reference itself uses float64 (RunningMeanStd state; Python-scalar Adam bias
 * corrections).  Summation order inside dot products is plain left-to-right,
 * so results agree with torch to fp32 round-off (~1e-6 rel), not bit-for-bit.
 *
 * Abbreviation: a2c/ = /root/reference/third_party/a2c_ppo_acktr/
 *
 * Flat parameter order == torch state_dict order of the reference modules:
 *   Policy (kind 0)  a2c/model.py:37-114,233-264 ; a2c/distributions.py:91-118
 *     actor.0.{W[H,O],b[H]} actor.2.{W[H,H],b[H]} critic.0.{W,b} critic.2.{W,b}
 *     critic_linear.{W[1,H],b[1]} fc_mean.{W[A,H],b[A]} logstd[A]
 *   SplitPolicy (kind 1)  a2c/model_split.py:39-95,157-238
 *     actor_contact.{0,2}.{W,b} actor_actuator.{0,2}.{W,b} critic_full.{0,2}.{W,b}
 *     critic_full.4.{W[1,H],b[1]} contact_mean.{W[4f,H],b} actuator_mean.{W[3f,H],b}
 *     contact_logstd.{W[4f,H],b} actuator_logstd.{W[3f,H],b}
 *   Discriminator  a2c/algo/gail.py:40-43
 *     trunk.0.{W[Hd,F],b[Hd]} trunk.2.{W[Hd,Hd],b[Hd]} trunk.4.{W[1,Hd],b[1]}
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_KIND_MLP 0
#define ORC_KIND_SPLIT 1

typedef struct {
    int kind;     /* ORC_KIND_* */
    int O, A, H;  /* obs dim, action dim, hidden size */
    int num_feet; /* SplitPolicy only; A == 7*num_feet (a2c/model_split.py:205) */
    int Hc;       /* hidden size of the CRITIC trunk when it differs from H (0 = H): Policy.reset_critic rebuilds a 64-unit
                   * critic beside an actor of any width (a2c/model.py:80-87, called by a2c/main.py:85 on every warm start) */
} orc_policy_dims;

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
} orc_ppo_cfg;

/* ------------------------------------------------------------------ helpers */

/* y[n] = W[n,k] x[k] + b[n]   (nn.Linear) */
static void linear(const float *W, const float *b, const float *x, int n, int k, float *y) {
    for (int i = 0; i < n; ++i) {
        float acc = b ? b[i] : 0.0f;
        const float *w = W + (size_t)i * k;
        for (int j = 0; j < k; ++j) acc += w[j] * x[j];
        y[i] = acc;
    }
}

/* dx[k] (+)= W[n,k]^T dy[n] */
static void linear_bwd_x(const float *W, const float *dy, int n, int k, float *dx, int accumulate) {
    if (!accumulate) memset(dx, 0, sizeof(float) * k);
    for (int i = 0; i < n; ++i) {
        const float *w = W + (size_t)i * k;
        float d = dy[i];
        for (int j = 0; j < k; ++j) dx[j] += w[j] * d;
    }
}

/* dW[n,k] += dy[n] x[k]^T ; db[n] += dy[n] */
static void linear_bwd_w(float *dW, float *db, const float *dy, const float *x, int n, int k) {
    for (int i = 0; i < n; ++i) {
        float d = dy[i];
        float *w = dW + (size_t)i * k;
        for (int j = 0; j < k; ++j) w[j] += d * x[j];
        if (db) db[i] += d;
    }
}

static float log_sigmoid(float x) { /* min(x,0) - log1p(exp(-|x|)) */
    return fminf(x, 0.0f) - log1pf(expf(-fabsf(x)));
}

static float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

/* ------------------------------------------------------------ policy layout */

typedef struct {
    int n_trunks;        /* 2 (MLP: actor, critic) or 3 (split: contact, actuator, critic) */
    int Ht[3], Hmax;     /* hidden size per trunk (the critic's may differ: orc_policy_dims.Hc) */
    size_t w1[3], b1[3], w2[3], b2[3];
    size_t vw, vb;       /* value head */
    /* MLP */
    size_t mw, mb, logstd;
    /* split */
    size_t cmw, cmb, amw, amb, clw, clb, alw, alb;
    int nc, na;          /* 4f, 3f */
    size_t total;
} pol_layout;

static pol_layout policy_layout(const orc_policy_dims *d) {
    pol_layout L;
    memset(&L, 0, sizeof L);
    size_t o = 0;
    int O = d->O, H = d->H, A = d->A;
    L.n_trunks = d->kind == ORC_KIND_MLP ? 2 : 3;
    L.Hmax = H;
    for (int t = 0; t < L.n_trunks; ++t) {
        const int Ht = (t == L.n_trunks - 1 && d->Hc > 0) ? d->Hc : H;
        L.Ht[t] = Ht;
        if (Ht > L.Hmax) L.Hmax = Ht;
        L.w1[t] = o; o += (size_t)Ht * O;
        L.b1[t] = o; o += Ht;
        L.w2[t] = o; o += (size_t)Ht * Ht;
        L.b2[t] = o; o += Ht;
    }
    L.vw = o; o += L.Ht[L.n_trunks - 1];
    L.vb = o; o += 1;
    if (d->kind == ORC_KIND_MLP) {
        L.mw = o; o += (size_t)A * H;
        L.mb = o; o += A;
        L.logstd = o; o += A;
    } else {
        L.nc = 4 * d->num_feet;
        L.na = 3 * d->num_feet;
        L.cmw = o; o += (size_t)L.nc * H;
        L.cmb = o; o += L.nc;
        L.amw = o; o += (size_t)L.na * H;
        L.amb = o; o += L.na;
        L.clw = o; o += (size_t)L.nc * H;
        L.clb = o; o += L.nc;
        L.alw = o; o += (size_t)L.na * H;
        L.alb = o; o += L.na;
    }
    L.total = o;
    return L;
}

int64_t orc_policy_num_params(const orc_policy_dims *d) { return (int64_t)policy_layout(d).total; }

/* index of the critic trunk: the state_dict puts it 2nd for Policy, 3rd for SplitPolicy */
static int critic_trunk(const orc_policy_dims *d) { return d->kind == ORC_KIND_MLP ? 1 : 2; }

/* Per-row forward.  h1/h2: [n_trunks][H] activations kept for backward.
 * MLPBase.forward a2c/model.py:255-264 ; SplitPolicyBaseNew.forward a2c/model_split.py:187-198 ;
 * DiagGaussian.forward a2c/distributions.py:109-118 ; StateDiagGaussianNew.forward a2c/model_split.py:222-237 */
static void policy_row_forward(const orc_policy_dims *d, const pol_layout *L, const float *P,
                               const float *x, float *h1, float *h2, float *value, float *mean,
                               float *logstd) {
    /* h1 / h2 are [n_trunks][Hmax]; trunk t uses its first Ht[t] entries */
    int O = d->O, H = L->Hmax, A = d->A;
    float *z = (float *)malloc(sizeof(float) * H);
    for (int t = 0; t < L->n_trunks; ++t) {
        const int Ht = L->Ht[t];
        linear(P + L->w1[t], P + L->b1[t], x, Ht, O, z);
        for (int i = 0; i < Ht; ++i) h1[t * H + i] = tanhf(z[i]);
        linear(P + L->w2[t], P + L->b2[t], h1 + t * H, Ht, Ht, z);
        for (int i = 0; i < Ht; ++i) h2[t * H + i] = tanhf(z[i]);
    }
    free(z);
    int ct = critic_trunk(d);
    linear(P + L->vw, P + L->vb, h2 + ct * H, 1, L->Ht[ct], value);
    H = d->H;   /* the actor heads below read actor trunks, which are all H wide */
    if (d->kind == ORC_KIND_MLP) {
        linear(P + L->mw, P + L->mb, h2, A, H, mean);
        for (int k = 0; k < A; ++k) logstd[k] = P[L->logstd + k];
    } else {
        const int S = L->Hmax;   /* stride between the trunks' activation rows */
        linear(P + L->cmw, P + L->cmb, h2, L->nc, H, mean);
        linear(P + L->amw, P + L->amb, h2 + S, L->na, H, mean + L->nc);
        linear(P + L->clw, P + L->clb, h2, L->nc, H, logstd);
        linear(P + L->alw, P + L->alb, h2 + S, L->na, H, logstd + L->nc);
    }
}

#define HALF_LOG_2PI 0.91893853320467274178f /* math.log(math.sqrt(2*math.pi)) */

/* FixedNormal.log_probs a2c/distributions.py:52-53 (torch Normal.log_prob, summed over dims) */
static float normal_logp_sum(const float *a, const float *mean, const float *logstd, int A) {
    float s = 0.0f;
    for (int k = 0; k < A; ++k) {
        float sigma = expf(logstd[k]);
        float var = sigma * sigma;
        float diff = a[k] - mean[k];
        s += -(diff * diff) / (2.0f * var) - logf(sigma) - HALF_LOG_2PI;
    }
    return s;
}

/* FixedNormal.entropy a2c/distributions.py:55-56 */
static float normal_entropy_sum(const float *logstd, int A) {
    float s = 0.0f;
    for (int k = 0; k < A; ++k) s += 0.5f + HALF_LOG_2PI + logf(expf(logstd[k]));
    return s;
}

void orc_policy_forward(const orc_policy_dims *d, const float *P, const float *obs, int n,
                        float *value, float *mean, float *logstd) {
    pol_layout L = policy_layout(d);
    float *h1 = (float *)malloc(sizeof(float) * 3 * L.Hmax);
    float *h2 = (float *)malloc(sizeof(float) * 3 * L.Hmax);
    for (int r = 0; r < n; ++r)
        policy_row_forward(d, &L, P, obs + (size_t)r * d->O, h1, h2, value + r,
                           mean + (size_t)r * d->A, logstd + (size_t)r * d->A);
    free(h1);
    free(h2);
}

/* Policy.act a2c/model.py:89-101 / SplitPolicy.act a2c/model_split.py:70-82.
 * noise == NULL -> deterministic (dist.mode()); else action = mean + std*noise
 * (torch.normal(mean, std) == randn*std + mean). */
void orc_policy_act(const orc_policy_dims *d, const float *P, const float *obs, int n,
                    const float *noise, float *value, float *action, float *logp) {
    int A = d->A;
    float *mean = (float *)malloc(sizeof(float) * (size_t)n * A);
    float *ls = (float *)malloc(sizeof(float) * (size_t)n * A);
    orc_policy_forward(d, P, obs, n, value, mean, ls);
    for (int r = 0; r < n; ++r) {
        for (int k = 0; k < A; ++k) {
            size_t i = (size_t)r * A + k;
            action[i] = noise ? noise[i] * expf(ls[i]) + mean[i] : mean[i];
        }
        logp[r] = normal_logp_sum(action + (size_t)r * A, mean + (size_t)r * A, ls + (size_t)r * A, A);
    }
    free(mean);
    free(ls);
}

/* Policy.evaluate_actions a2c/model.py:107-114 ; entropy = dist.entropy().mean() */
void orc_policy_evaluate(const orc_policy_dims *d, const float *P, const float *obs,
                         const float *action, int n, float *value, float *logp, float *entropy) {
    int A = d->A;
    float *mean = (float *)malloc(sizeof(float) * (size_t)n * A);
    float *ls = (float *)malloc(sizeof(float) * (size_t)n * A);
    orc_policy_forward(d, P, obs, n, value, mean, ls);
    double ent = 0.0;
    for (int r = 0; r < n; ++r) {
        logp[r] = normal_logp_sum(action + (size_t)r * A, mean + (size_t)r * A, ls + (size_t)r * A, A);
        ent += normal_entropy_sum(ls + (size_t)r * A, A);
    }
    *entropy = (float)(ent / n);
    free(mean);
    free(ls);
}

/* --------------------------------------------------------------------- GAE */

/* RolloutStorage.compute_returns a2c/storage.py:103-142 (all four branches).
 * rewards[T,N]; value_preds, returns, masks, bad_masks [T+1,N]; next_value[N]. */
void orc_compute_returns(int T, int N, const float *rewards, float *value_preds, float *returns,
                         const float *masks, const float *bad_masks, const float *next_value,
                         int use_gae, float gamma, float lam, int proper_time_limits) {
    if (use_gae) {
        for (int n = 0; n < N; ++n) value_preds[(size_t)T * N + n] = next_value[n];
        for (int n = 0; n < N; ++n) {
            float gae = 0.0f;
            for (int t = T - 1; t >= 0; --t) {
                size_t i = (size_t)t * N + n, j = (size_t)(t + 1) * N + n;
                float delta = rewards[i] + gamma * value_preds[j] * masks[j] - value_preds[i];
                gae = delta + gamma * lam * masks[j] * gae;
                if (proper_time_limits) gae = gae * bad_masks[j];
                returns[i] = gae + value_preds[i];
            }
        }
    } else {
        for (int n = 0; n < N; ++n) returns[(size_t)T * N + n] = next_value[n];
        for (int n = 0; n < N; ++n)
            for (int t = T - 1; t >= 0; --t) {
                size_t i = (size_t)t * N + n, j = (size_t)(t + 1) * N + n;
                if (proper_time_limits)
                    returns[i] = (returns[j] * gamma * masks[j] + rewards[i]) * bad_masks[j] +
                                 (1.0f - bad_masks[j]) * value_preds[i];
                else
                    returns[i] = returns[j] * gamma * masks[j] + rewards[i];
            }
    }
}

/* ------------------------------------------------------------ Adam / clip */

/* torch.optim.Adam (single-tensor math): m.lerp_(g, 1-b1); v = b2*v + (1-b2) g*g;
 * denom = sqrt(v)/sqrt(1-b2^t) + eps; p -= (lr/(1-b1^t)) * m/denom.  Bias corrections are
 * Python doubles in torch.  a2c/algo/ppo.py:57,145 (eps 1e-5) ; a2c/algo/gail.py:48,188 (defaults). */
void orc_adam_step(float *P, const float *G, float *M, float *V, int64_t *t, int64_t n, float lr,
                   float eps) {
    const double b1 = 0.9, b2 = 0.999;
    *t += 1;
    double bc1 = 1.0 - pow(b1, (double)*t);
    double bc2 = 1.0 - pow(b2, (double)*t);
    float step_size = (float)((double)lr / bc1);
    float bc2_sqrt = (float)sqrt(bc2);
    for (int64_t i = 0; i < n; ++i) {
        float g = G[i];
        M[i] = M[i] + (g - M[i]) * (float)(1.0 - b1);
        V[i] = V[i] * (float)b2 + (float)(1.0 - b2) * g * g;
        float denom = sqrtf(V[i]) / bc2_sqrt + eps;
        P[i] = P[i] - step_size * (M[i] / denom);
    }
}

/* nn.utils.clip_grad_norm_ a2c/algo/ppo.py:143: coef = max_norm/(||g||+1e-6) clamped to 1. */
float orc_clip_grad_norm(float *G, int64_t n, float max_norm) {
    double ss = 0.0;
    for (int64_t i = 0; i < n; ++i) ss += (double)G[i] * (double)G[i];
    float total = (float)sqrt(ss);
    float coef = max_norm / (total + 1e-6f);
    if (coef > 1.0f) coef = 1.0f;
    for (int64_t i = 0; i < n; ++i) G[i] *= coef;
    return total;
}

/* --------------------------------------------------------------------- PPO */

/* Gradient SUM over the given rows of  value_loss*vcoef + action_loss - entropy*ecoef, with the
 * per-row mean factor inv_B applied (inv_B = 1/global minibatch rows).  Adds into G (flat,
 * state_dict order) and into sums[3] = { sum_r max(u,w)*0.5, sum_r -min(surr1,surr2), sum_r entropy_r }
 * (un-normalised; caller multiplies by inv_B).   a2c/algo/ppo.py:88-106,138-142.
 * rows[] are flattened indices t*N+n into obs[:-1], actions, value_preds[:-1], returns[:-1],
 * old_logp, adv  (a2c/storage.py:168-185). */
void orc_ppo_grad_rows(const orc_policy_dims *d, const float *P, const orc_ppo_cfg *cfg,
                       const float *obs, const float *actions, const float *value_preds,
                       const float *returns, const float *old_logp, const float *adv,
                       const int64_t *rows, int n_rows, float inv_B, float *G, double *sums) {
    pol_layout L = policy_layout(d);
    /* activation rows are [n_trunks][S = Hmax]; H stays the ACTOR width (every head but the value head reads actors) */
    int O = d->O, H = d->H, S = L.Hmax, A = d->A, nt = L.n_trunks, ct = critic_trunk(d);
    float *h1 = (float *)malloc(sizeof(float) * nt * S), *h2 = (float *)malloc(sizeof(float) * nt * S);
    float *dh2 = (float *)calloc((size_t)nt * S, sizeof(float));
    float *dz = (float *)malloc(sizeof(float) * S), *dh1 = (float *)malloc(sizeof(float) * S);
    float *mean = (float *)malloc(sizeof(float) * A), *ls = (float *)malloc(sizeof(float) * A);
    float *dmean = (float *)malloc(sizeof(float) * A), *dls = (float *)malloc(sizeof(float) * A);
    float eps = cfg->clip_param;
    for (int r = 0; r < n_rows; ++r) {
        int64_t idx = rows[r];
        const float *x = obs + (size_t)idx * O, *a = actions + (size_t)idx * A;
        float v;
        policy_row_forward(d, &L, P, x, h1, h2, &v, mean, ls);
        float logp = normal_logp_sum(a, mean, ls, A);
        float ent = normal_entropy_sum(ls, A);
        float adv_r = adv[idx], R = returns[idx], v_old = value_preds[idx];
        /* clipped surrogate a2c/algo/ppo.py:92-97 */
        float ratio = expf(logp - old_logp[idx]);
        float surr1 = ratio * adv_r;
        float rc = fminf(fmaxf(ratio, 1.0f - eps), 1.0f + eps);
        float surr2 = rc * adv_r;
        float w1 = surr1 < surr2 ? 1.0f : (surr1 > surr2 ? 0.0f : 0.5f); /* torch.min tie -> 1/2,1/2 */
        float in_range = (ratio >= 1.0f - eps && ratio <= 1.0f + eps) ? 1.0f : 0.0f;
        float dratio = -inv_B * (w1 * adv_r + (1.0f - w1) * adv_r * in_range);
        float dlogp = dratio * ratio;
        sums[1] += (double)(-fminf(surr1, surr2));
        /* value loss a2c/algo/ppo.py:99-108 */
        float dv;
        if (cfg->use_clipped_value_loss) {
            float dvv = v - v_old;
            float vc = v_old + fminf(fmaxf(dvv, -eps), eps);
            float u = (v - R) * (v - R), w = (vc - R) * (vc - R);
            float m1 = u > w ? 1.0f : (u < w ? 0.0f : 0.5f);
            float pass = (dvv >= -eps && dvv <= eps) ? 1.0f : 0.0f;
            dv = 0.5f * inv_B * (m1 * 2.0f * (v - R) + (1.0f - m1) * 2.0f * (vc - R) * pass);
            sums[0] += (double)(0.5f * fmaxf(u, w));
        } else {
            dv = 0.5f * inv_B * (-2.0f) * (R - v);
            sums[0] += (double)(0.5f * (R - v) * (R - v));
        }
        dv *= cfg->value_loss_coef;
        sums[2] += (double)ent;
        /* d logp / d mean, d logp / d logstd ; entropy: -ecoef*inv_B per dim */
        for (int k = 0; k < A; ++k) {
            float sigma = expf(ls[k]);
            float var = sigma * sigma;
            float diff = a[k] - mean[k];
            dmean[k] = dlogp * diff / var;
            dls[k] = dlogp * (diff * diff / var - 1.0f) - cfg->entropy_coef * inv_B;
        }
        /* heads backward */
        memset(dh2, 0, sizeof(float) * nt * S);
        linear_bwd_w(G + L.vw, G + L.vb, &dv, h2 + ct * S, 1, L.Ht[ct]);
        linear_bwd_x(P + L.vw, &dv, 1, L.Ht[ct], dh2 + ct * S, 1);
        if (d->kind == ORC_KIND_MLP) {
            linear_bwd_w(G + L.mw, G + L.mb, dmean, h2, A, H);
            linear_bwd_x(P + L.mw, dmean, A, H, dh2, 1);
            for (int k = 0; k < A; ++k) G[L.logstd + k] += dls[k];
        } else {
            linear_bwd_w(G + L.cmw, G + L.cmb, dmean, h2, L.nc, H);
            linear_bwd_x(P + L.cmw, dmean, L.nc, H, dh2, 1);
            linear_bwd_w(G + L.clw, G + L.clb, dls, h2, L.nc, H);
            linear_bwd_x(P + L.clw, dls, L.nc, H, dh2, 1);
            linear_bwd_w(G + L.amw, G + L.amb, dmean + L.nc, h2 + S, L.na, H);
            linear_bwd_x(P + L.amw, dmean + L.nc, L.na, H, dh2 + S, 1);
            linear_bwd_w(G + L.alw, G + L.alb, dls + L.nc, h2 + S, L.na, H);
            linear_bwd_x(P + L.alw, dls + L.nc, L.na, H, dh2 + S, 1);
        }
        /* trunks backward */
        for (int t = 0; t < nt; ++t) {
            const int Ht = L.Ht[t];
            for (int i = 0; i < Ht; ++i) dz[i] = dh2[t * S + i] * (1.0f - h2[t * S + i] * h2[t * S + i]);
            linear_bwd_w(G + L.w2[t], G + L.b2[t], dz, h1 + t * S, Ht, Ht);
            linear_bwd_x(P + L.w2[t], dz, Ht, Ht, dh1, 0);
            for (int i = 0; i < Ht; ++i) dz[i] = dh1[i] * (1.0f - h1[t * S + i] * h1[t * S + i]);
            linear_bwd_w(G + L.w1[t], G + L.b1[t], dz, x, Ht, O);
        }
    }
    free(h1); free(h2); free(dh2); free(dz); free(dh1); free(mean); free(ls); free(dmean); free(dls);
}

/* advantages = returns[:-1]-value_preds[:-1]; (adv-mean)/(std_unbiased+1e-5)  a2c/algo/ppo.py:66-68 */
void orc_advantages(const float *returns, const float *value_preds, int64_t n, float *adv) {
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        adv[i] = returns[i] - value_preds[i];
        s += adv[i];
    }
    float mean = (float)(s / (double)n);
    double ss = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        double dd = (double)adv[i] - (double)mean;
        ss += dd * dd;
    }
    float std = (float)sqrt(ss / (double)(n - 1));
    for (int64_t i = 0; i < n; ++i) adv[i] = (adv[i] - mean) / (std + 1e-5f);
}

/* All-cores form of orc_ppo_grad_rows for bench.py's cpu_baseline leg: rows are split into n_threads contiguous
 * chunks, each chunk accumulates into a private gradient, and the chunks are summed in chunk order (deterministic for a
 * given n_threads; differs from the single-thread sum only by fp32 summation order). */
void orc_ppo_grad_rows_mt(const orc_policy_dims *d, const float *P, const orc_ppo_cfg *cfg,
                          const float *obs, const float *actions, const float *value_preds,
                          const float *returns, const float *old_logp, const float *adv,
                          const int64_t *rows, int n_rows, float inv_B, float *G, double *sums, int n_threads) {
    int64_t n = orc_policy_num_params(d);
    if (n_threads < 1) n_threads = 1;
    float *Gt = (float *)calloc((size_t)n_threads * n, sizeof(float));
    double *St = (double *)calloc((size_t)n_threads * 3, sizeof(double));
#pragma omp parallel for num_threads(n_threads) schedule(static, 1)
    for (int t = 0; t < n_threads; ++t) {
        int lo = (int)((int64_t)n_rows * t / n_threads), hi = (int)((int64_t)n_rows * (t + 1) / n_threads);
        orc_ppo_grad_rows(d, P, cfg, obs, actions, value_preds, returns, old_logp, adv, rows + lo, hi - lo, inv_B,
                          Gt + (size_t)t * n, St + 3 * t);
    }
    for (int t = 0; t < n_threads; ++t) {
        for (int64_t i = 0; i < n; ++i) G[i] += Gt[(size_t)t * n + i];
        for (int i = 0; i < 3; ++i) sums[i] += St[3 * t + i];
    }
    free(Gt);
    free(St);
}

/* One optimizer step from an already-summed gradient: clip + Adam.  a2c/algo/ppo.py:143-145 */
void orc_ppo_apply(float *P, float *G, float *M, float *V, int64_t *t, int64_t n,
                   const orc_ppo_cfg *cfg) {
    orc_clip_grad_norm(G, n, cfg->max_grad_norm);
    orc_adam_step(P, G, M, V, t, n, cfg->lr, cfg->eps);
}

/* PPO.update a2c/algo/ppo.py:65-157.  perms: [ppo_epoch][T*N] int64 (the randperm each epoch's
 * SubsetRandomSampler draws, a2c/storage.py:159-162); minibatch k = perm[k*mb:(k+1)*mb], drop_last. */
void orc_ppo_update(const orc_policy_dims *d, float *P, float *M, float *V, int64_t *adam_t,
                    const orc_ppo_cfg *cfg, int T, int N, const float *obs, const float *actions,
                    const float *value_preds, const float *returns, const float *old_logp,
                    const int64_t *perms, float *out3) {
    int64_t n = orc_policy_num_params(d), TN = (int64_t)T * N;
    float *adv = (float *)malloc(sizeof(float) * TN);
    float *G = (float *)malloc(sizeof(float) * n);
    orc_advantages(returns, value_preds, TN, adv);
    int mb = (int)(TN / cfg->num_mini_batch);
    double tot[3] = {0, 0, 0};
    for (int e = 0; e < cfg->ppo_epoch; ++e)
        for (int k = 0; k < cfg->num_mini_batch; ++k) {
            double sums[3] = {0, 0, 0};
            memset(G, 0, sizeof(float) * n);
            orc_ppo_grad_rows(d, P, cfg, obs, actions, value_preds, returns, old_logp, adv,
                              perms + (size_t)e * TN + (size_t)k * mb, mb, 1.0f / (float)mb, G, sums);
            orc_ppo_apply(P, G, M, V, adam_t, n, cfg);
            for (int i = 0; i < 3; ++i) tot[i] += (double)(float)(sums[i] / mb);
        }
    int nu = cfg->ppo_epoch * cfg->num_mini_batch;
    for (int i = 0; i < 3; ++i) out3[i] = (float)(tot[i] / nu);
    free(adv);
    free(G);
}

/* ----------------------------------------------------------- discriminator */

int64_t orc_disc_num_params(int F, int Hd) { return (int64_t)Hd * F + Hd + (int64_t)Hd * Hd + Hd + Hd + 1; }

typedef struct { size_t w1, b1, w2, b2, w3, b3; } disc_layout;
static disc_layout disc_lay(int F, int Hd) {
    disc_layout L;
    size_t o = 0;
    L.w1 = o; o += (size_t)Hd * F;
    L.b1 = o; o += Hd;
    L.w2 = o; o += (size_t)Hd * Hd;
    L.b2 = o; o += Hd;
    L.w3 = o; o += Hd;
    L.b3 = o;
    return L;
}

static float disc_row_forward(int F, int Hd, const disc_layout *L, const float *P, const float *x,
                              float *h1, float *h2) {
    float d;
    linear(P + L->w1, P + L->b1, x, Hd, F, h1);
    for (int i = 0; i < Hd; ++i) h1[i] = tanhf(h1[i]);
    linear(P + L->w2, P + L->b2, h1, Hd, Hd, h2);
    for (int i = 0; i < Hd; ++i) h2[i] = tanhf(h2[i]);
    linear(P + L->w3, P + L->b3, h2, 1, Hd, &d);
    return d;
}

/* backward of the plain forward graph given dd = dL/dd for this row */
static void disc_row_backward(int F, int Hd, const disc_layout *L, const float *P, float *G,
                              const float *x, const float *h1, const float *h2, float dd,
                              float *tmp /* 2*Hd */) {
    float *dz2 = tmp, *dz1 = tmp + Hd;
    linear_bwd_w(G + L->w3, G + L->b3, &dd, h2, 1, Hd);
    for (int i = 0; i < Hd; ++i) dz2[i] = dd * P[L->w3 + i] * (1.0f - h2[i] * h2[i]);
    linear_bwd_w(G + L->w2, G + L->b2, dz2, h1, Hd, Hd);
    linear_bwd_x(P + L->w2, dz2, Hd, Hd, dz1, 0);
    for (int i = 0; i < Hd; ++i) dz1[i] *= (1.0f - h1[i] * h1[i]);
    linear_bwd_w(G + L->w1, G + L->b1, dz1, x, Hd, F);
}

/* Gradient SUM over nb (expert,policy,alpha) row triples of
 *   BCE(D(e),1)/Bg + BCE(D(p),0)/Bg + lambda*((||dD/dx(mix)||-1)^2)/Bg      (Bg = global batch)
 * a2c/algo/gail.py:165-188 with compute_grad_pen_combined :67-89 (lambda = 10); double-backward
 * derived by hand.  sums[3] += { expert_bce, policy_bce, (||g||-1)^2 }. */
void orc_disc_grad_rows(int F, int Hd, const float *P, const float *expert_rows /*[nb,F]*/,
                        const float *policy_rows /*[nb,F]*/, const float *alpha /*[nb]*/, int nb,
                        float inv_B, float lambda_, float *G, double *sums) {
    disc_layout L = disc_lay(F, Hd);
    const float *W1 = P + L.w1, *W2 = P + L.w2, *w3 = P + L.w3;
    float *buf = (float *)malloc(sizeof(float) * (size_t)(16 * Hd + 3 * F));
    float *h1 = buf, *h2 = h1 + Hd, *tmp = h2 + Hd; /* tmp: 2*Hd */
    float *s1 = tmp + 2 * Hd, *s2 = s1 + Hd, *d2 = s2 + Hd, *u1 = d2 + Hd, *d1 = u1 + Hd;
    float *bd1 = d1 + Hd, *bu1 = bd1 + Hd, *bd2 = bu1 + Hd, *z2b = bd2 + Hd, *h1b = z2b + Hd;
    float *z1b = h1b + Hd;
    float *xm = z1b + Hd, *gx = xm + F, *gb = gx + F;
    for (int r = 0; r < nb; ++r) {
        const float *e = expert_rows + (size_t)r * F, *p = policy_rows + (size_t)r * F;
        /* policy_d, expert_d and the two BCE terms  a2c/algo/gail.py:168-176 */
        float dp = disc_row_forward(F, Hd, &L, P, p, h1, h2);
        sums[1] += (double)(dp - log_sigmoid(dp));
        disc_row_backward(F, Hd, &L, P, G, p, h1, h2, inv_B * sigmoidf(dp), tmp);
        float de = disc_row_forward(F, Hd, &L, P, e, h1, h2);
        sums[0] += (double)(-log_sigmoid(de));
        disc_row_backward(F, Hd, &L, P, G, e, h1, h2, inv_B * (sigmoidf(de) - 1.0f), tmp);
        /* gradient penalty a2c/algo/gail.py:72-88 */
        float al = alpha[r];
        for (int j = 0; j < F; ++j) xm[j] = al * e[j] + (1.0f - al) * p[j];
        (void)disc_row_forward(F, Hd, &L, P, xm, h1, h2);
        for (int i = 0; i < Hd; ++i) {
            s1[i] = 1.0f - h1[i] * h1[i];
            s2[i] = 1.0f - h2[i] * h2[i];
            d2[i] = w3[i] * s2[i];
        }
        linear_bwd_x(W2, d2, Hd, Hd, u1, 0); /* u1 = W2^T d2 */
        for (int i = 0; i < Hd; ++i) d1[i] = u1[i] * s1[i];
        linear_bwd_x(W1, d1, Hd, F, gx, 0); /* g_x = W1^T d1 */
        float nn = 0.0f;
        for (int j = 0; j < F; ++j) nn += gx[j] * gx[j];
        nn = sqrtf(nn);
        sums[2] += (double)((nn - 1.0f) * (nn - 1.0f));
        float c = nn > 0.0f ? lambda_ * 2.0f * inv_B * (nn - 1.0f) / nn : 0.0f;
        for (int j = 0; j < F; ++j) gb[j] = c * gx[j];
        /* double-backward */
        linear_bwd_w(G + L.w1, NULL, d1, gb, Hd, F);       /* dW1 += d1 gb^T */
        linear(W1, NULL, gb, Hd, F, bd1);                   /* bd1 = W1 gb    */
        for (int i = 0; i < Hd; ++i) bu1[i] = bd1[i] * s1[i];
        linear_bwd_w(G + L.w2, NULL, d2, bu1, Hd, Hd);     /* dW2 += d2 bu1^T */
        linear(W2, NULL, bu1, Hd, Hd, bd2);                 /* bd2 = W2 bu1   */
        for (int i = 0; i < Hd; ++i) {
            G[L.w3 + i] += bd2[i] * s2[i];                   /* dw3 += bd2 . s2 */
            float s2b = bd2[i] * w3[i];
            float h2b = -2.0f * h2[i] * s2b;
            z2b[i] = h2b * s2[i];
        }
        linear_bwd_w(G + L.w2, G + L.b2, z2b, h1, Hd, Hd); /* dW2 += z2b h1^T ; db2 += z2b */
        linear_bwd_x(W2, z2b, Hd, Hd, h1b, 0);             /* h1b = W2^T z2b */
        for (int i = 0; i < Hd; ++i) {
            float s1b = bd1[i] * u1[i];
            z1b[i] = (h1b[i] - 2.0f * h1[i] * s1b) * s1[i];
        }
        linear_bwd_w(G + L.w1, G + L.b1, z1b, xm, Hd, F);  /* dW1 += z1b xm^T ; db1 += z1b */
    }
    free(buf);
}

/* All-cores form of orc_disc_grad_rows (bench.py cpu_baseline): row triples split over n_threads, private gradients
 * summed in chunk order. */
void orc_disc_grad_rows_mt(int F, int Hd, const float *P, const float *expert_rows, const float *policy_rows,
                           const float *alpha, int nb, float inv_B, float lambda_, float *G, double *sums, int n_threads) {
    int64_t n = orc_disc_num_params(F, Hd);
    if (n_threads < 1) n_threads = 1;
    if (n_threads > nb) n_threads = nb;
    float *Gt = (float *)calloc((size_t)n_threads * n, sizeof(float));
    double *St = (double *)calloc((size_t)n_threads * 3, sizeof(double));
#pragma omp parallel for num_threads(n_threads) schedule(static, 1)
    for (int t = 0; t < n_threads; ++t) {
        int lo = (int)((int64_t)nb * t / n_threads), hi = (int)((int64_t)nb * (t + 1) / n_threads);
        orc_disc_grad_rows(F, Hd, P, expert_rows + (size_t)lo * F, policy_rows + (size_t)lo * F, alpha + lo, hi - lo, inv_B,
                           lambda_, Gt + (size_t)t * n, St + 3 * t);
    }
    for (int t = 0; t < n_threads; ++t) {
        for (int64_t i = 0; i < n; ++i) G[i] += Gt[(size_t)t * n + i];
        for (int i = 0; i < 3; ++i) sums[i] += St[3 * t + i];
    }
    free(Gt);
    free(St);
}

/* Discriminator.update_gail_dyn a2c/algo/gail.py:154-193.
 * expert [Ne,F]; obs_feat [(T+1),N,F] (policy rows = next_obs_feat = obs_feat[1:], a2c/storage.py:172);
 * expert_perm [Ne] (DataLoader shuffle), policy_perm [T*N] (feed_forward_generator), alpha [n_d*B].
 * n_d = min(Ne/B (drop_last iff Ne > B), (T*N)/B)  -- zip stops at the shorter iterator.
 * Returns out3 = {mean(gail_loss+grad_pen), mean expert_loss, mean policy_loss} and n_d. */
int orc_disc_update(int F, int Hd, float *P, float *M, float *V, int64_t *adam_t, float lr,
                    float adam_eps, const float *expert, int64_t Ne, const float *obs_feat, int T,
                    int N, int B, const int64_t *expert_perm, const int64_t *policy_perm,
                    const float *alpha, float *out3) {
    int64_t np_ = orc_disc_num_params(F, Hd), TN = (int64_t)T * N;
    int64_t n_e = Ne > B ? Ne / B : 1; /* drop_last = len > batch (a2c/main_gail_dyn_ppo.py:170) */
    int64_t n_p = TN / B;
    int n_d = (int)(n_e < n_p ? n_e : n_p);
    float *G = (float *)malloc(sizeof(float) * np_);
    float *eb = (float *)malloc(sizeof(float) * (size_t)B * F), *pb = (float *)malloc(sizeof(float) * (size_t)B * F);
    double tot[3] = {0, 0, 0};
    const float *next_feat = obs_feat + (size_t)N * F;
    for (int k = 0; k < n_d; ++k) {
        int nb = (Ne > B) ? B : (int)Ne; /* a short single expert batch only when Ne <= B */
        for (int r = 0; r < nb; ++r) {
            memcpy(eb + (size_t)r * F, expert + (size_t)expert_perm[(size_t)k * B + r] * F, sizeof(float) * F);
            memcpy(pb + (size_t)r * F, next_feat + (size_t)policy_perm[(size_t)k * B + r] * F, sizeof(float) * F);
        }
        double sums[3] = {0, 0, 0};
        memset(G, 0, sizeof(float) * np_);
        orc_disc_grad_rows(F, Hd, P, eb, pb, alpha + (size_t)k * B, nb, 1.0f / (float)nb, 10.0f, G, sums);
        orc_adam_step(P, G, M, V, adam_t, np_, lr, adam_eps);
        float el = (float)(sums[0] / nb), pl = (float)(sums[1] / nb), gp = 10.0f * (float)(sums[2] / nb);
        tot[0] += (double)(el + pl + gp);
        tot[1] += (double)el;
        tot[2] += (double)pl;
    }
    for (int i = 0; i < 3; ++i) out3[i] = (float)(tot[i] / n_d);
    free(G); free(eb); free(pb);
    return n_d;
}

/* Discriminator.predict_reward_combined a2c/algo/gail.py:201-210.
 * first != 0 <=> self.returns is None (returns := reward.clone()). */
void orc_disc_predict_reward(int F, int Hd, const float *P, const float *x, int n, float gamma,
                             const float *masks, float offset, float *returns, int first,
                             float *reward) {
    disc_layout L = disc_lay(F, Hd);
    float *h1 = (float *)malloc(sizeof(float) * 2 * Hd), *h2 = h1 + Hd;
    for (int r = 0; r < n; ++r) {
        float d = disc_row_forward(F, Hd, &L, P, x + (size_t)r * F, h1, h2);
        float s = sigmoidf(d);
        float rw = logf(s + 1e-7f) - logf(1.0f - s + 1e-7f) + offset;
        reward[r] = rw;
        returns[r] = first ? rw : returns[r] * gamma * masks[r] + rw;
    }
    free(h1);
}

/* RunningMeanStd.update a2c/baselines/common/running_mean_std.py:34-58 on a float32 batch:
 * numpy computes batch mean/var in float32, the running state is float64. state = {mean,var,count}. */
void orc_rms_update(double *state, const float *x, int n) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += x[i];
    float bmean = (float)(s / n);
    double ss = 0.0;
    for (int i = 0; i < n; ++i) {
        float dd = x[i] - bmean;
        ss += (double)(dd * dd);
    }
    float bvar = (float)(ss / n);
    double mean = state[0], var = state[1], count = state[2];
    double delta = (double)bmean - mean, tot = count + n;
    double new_mean = mean + delta * n / tot;
    double M2 = var * count + (double)bvar * n + delta * delta * count * n / tot;
    state[0] = new_mean;
    state[1] = M2 / tot;
    state[2] = tot;
}

/* Reward relabel loop a2c/main_gail_dyn_ppo.py:275-292: for each step t
 *   rewards[t], ret = D.predict_reward_combined(obs_feat[t+1], gamma, masks[t], offset)
 *   ret_rms.update(ret) ; rewards[t] = clip(rewards[t]/sqrt(ret_rms.var+1e-7), -10, 10)
 * d_returns [N] is Discriminator.returns (persists across calls); *d_first = (returns is None). */
void orc_relabel(int F, int Hd, const float *P, int T, int N, const float *obs_feat,
                 const float *masks, float gamma, float offset, float *d_returns, int *d_first,
                 double *rms_state, float *rewards) {
    for (int t = 0; t < T; ++t) {
        float *rw = rewards + (size_t)t * N;
        orc_disc_predict_reward(F, Hd, P, obs_feat + (size_t)(t + 1) * N * F, N, gamma,
                                masks + (size_t)t * N, offset, d_returns, *d_first, rw);
        *d_first = 0;
        orc_rms_update(rms_state, d_returns, N);
        float scale = (float)sqrt(rms_state[1] + 1e-7);
        for (int n = 0; n < N; ++n) {
            /* numpy: float32 array / float64 scalar -> float32 result */
            float v = rw[n] / scale;
            rw[n] = fminf(fmaxf(v, -10.0f), 10.0f);
        }
    }
}

/* Alive-bonus offset a2c/main_gail_dyn_ppo.py:258-271:  r_sa = log d - log(1-d),
 * d = 1 - dones/(dones + T*N/expert_len), dones = sum(1-masks[0..T]) + N/2. */
double orc_alive_bonus(const float *masks, int T, int N, double gail_tar_length) {
    double dones = 0.0;
    for (int64_t i = 0; i < (int64_t)(T + 1) * N; ++i) dones += 1.0 - (double)masks[i];
    dones += N / 2.0;
    double expert_dones = ((double)T * N) / gail_tar_length;
    double d = 1.0 - dones / (dones + expert_dones);
    return log(d) - log(1.0 - d);
}
