/*
 * sg_cpu_fast.c -- a FAST CPU implementation of the two optimizer-step gradients of the GAIL+PPO update, for
 * bench.py's `cpu_baseline` leg only.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE.  oracle/sg_oracle.c is the parity oracle: scalar, row by row, left-to-right
 * float32 -- faithful, and therefore about half as fast as the reference's own PyTorch-CPU path (which runs batched
 * GEMMs through a vectorised BLAS).  Quoting the oracle's speed as "the CPU" would flatter the GPU.  This file restates
 * the same two functions
 *     fast_disc_grad_rows  == orc_disc_grad_rows   (a2c/algo/gail.py:67-89,165-188, double backward by hand)
 *     fast_ppo_grad_rows   == orc_ppo_grad_rows    (a2c/algo/ppo.py:88-106,138-142; a2c/model.py:255-264;
 *                                                   a2c/model_split.py:187-238; a2c/distributions.py:51-59,109-118)
 * the way a CPU wants them: the whole minibatch at once, every layer a blocked row-panel GEMM whose inner loop is a
 * unit-stride FMA stream the compiler vectorises (AVX2 / AVX-512), weights transposed once per step, tanh / exp
 * through the vector math library.  Summation order differs from the oracle's, nothing else: tests/test_oracle_golden.py
 * checks both functions against the oracle at the suite's 1e-4.  The product path never links, imports or calls it.
 * (a2c/ = /root/reference/third_party/a2c_ppo_acktr/)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_KIND_MLP 0
typedef struct { int kind, O, A, H, num_feet, Hc; } orc_policy_dims;   /* Hc: critic trunk width when it differs from H (0 = H) */
typedef struct {
    float clip_param; int ppo_epoch; int num_mini_batch; float value_loss_coef; float entropy_coef; float lr; float eps;
    float max_grad_norm; int use_clipped_value_loss;
} orc_ppo_cfg;

#define RESTRICT __restrict__
#define HALF_LOG_2PI 0.91893853320467274178f

/* ----------------------------------------------------------------------------------------- GEMM building blocks */
/* C[M,N] (+)= A[M,K] B[K,N]; rows of C in panels of 4, inner loop over N is a unit-stride FMA stream */
static void gemm_nn(int M, int N, int K, const float *RESTRICT A, int lda, const float *RESTRICT B, int ldb,
                    float *RESTRICT C, int ldc, int acc) {
    int i = 0;
    for (; i + 4 <= M; i += 4) {
        float *RESTRICT c0 = C + (size_t)i * ldc, *RESTRICT c1 = c0 + ldc, *RESTRICT c2 = c1 + ldc, *RESTRICT c3 = c2 + ldc;
        if (!acc) { memset(c0, 0, sizeof(float) * N); memset(c1, 0, sizeof(float) * N); memset(c2, 0, sizeof(float) * N); memset(c3, 0, sizeof(float) * N); }
        const float *a0 = A + (size_t)i * lda, *a1 = a0 + lda, *a2 = a1 + lda, *a3 = a2 + lda;
        for (int k = 0; k < K; ++k) {
            const float *RESTRICT b = B + (size_t)k * ldb;
            const float x0 = a0[k], x1 = a1[k], x2 = a2[k], x3 = a3[k];
            for (int j = 0; j < N; ++j) {
                const float bv = b[j];
                c0[j] += x0 * bv; c1[j] += x1 * bv; c2[j] += x2 * bv; c3[j] += x3 * bv;
            }
        }
    }
    for (; i < M; ++i) {
        float *RESTRICT c0 = C + (size_t)i * ldc;
        if (!acc) memset(c0, 0, sizeof(float) * N);
        const float *a0 = A + (size_t)i * lda;
        for (int k = 0; k < K; ++k) {
            const float *RESTRICT b = B + (size_t)k * ldb;
            const float x0 = a0[k];
            for (int j = 0; j < N; ++j) c0[j] += x0 * b[j];
        }
    }
}

/* C[M,N] += sum_r A[r,M]^T B[r,N]   (weight gradients: dW = dY^T X), panels of 4 rows of C */
static void gemm_tn_acc(int M, int N, int R, const float *RESTRICT A, int lda, const float *RESTRICT B, int ldb,
                        float *RESTRICT C, int ldc) {
    int m = 0;
    for (; m + 4 <= M; m += 4) {
        float *RESTRICT c0 = C + (size_t)m * ldc, *RESTRICT c1 = c0 + ldc, *RESTRICT c2 = c1 + ldc, *RESTRICT c3 = c2 + ldc;
        for (int r = 0; r < R; ++r) {
            const float *a = A + (size_t)r * lda + m;
            const float *RESTRICT b = B + (size_t)r * ldb;
            const float x0 = a[0], x1 = a[1], x2 = a[2], x3 = a[3];
            for (int j = 0; j < N; ++j) {
                const float bv = b[j];
                c0[j] += x0 * bv; c1[j] += x1 * bv; c2[j] += x2 * bv; c3[j] += x3 * bv;
            }
        }
    }
    for (; m < M; ++m) {
        float *RESTRICT c0 = C + (size_t)m * ldc;
        for (int r = 0; r < R; ++r) {
            const float x0 = A[(size_t)r * lda + m];
            const float *RESTRICT b = B + (size_t)r * ldb;
            for (int j = 0; j < N; ++j) c0[j] += x0 * b[j];
        }
    }
}

static void transpose(int rows, int cols, const float *RESTRICT A, float *RESTRICT At) {   /* At[cols,rows] */
    for (int i = 0; i < rows; ++i)
        for (int j = 0; j < cols; ++j) At[(size_t)j * rows + i] = A[(size_t)i * cols + j];
}

/* H = tanh(Z + b) over [R,N]; 1 - 2/(exp(2x)+1): the exp goes through the vector math library */
static void bias_tanh(int R, int N, float *RESTRICT Z, const float *RESTRICT b) {
    for (int r = 0; r < R; ++r) {
        float *RESTRICT z = Z + (size_t)r * N;
        for (int j = 0; j < N; ++j) {
            const float x = z[j] + b[j];
            z[j] = 1.0f - 2.0f / (expf(2.0f * x) + 1.0f);
        }
    }
}

static void colsum_acc(int R, int N, const float *RESTRICT A, float *RESTRICT g) {
    for (int r = 0; r < R; ++r) {
        const float *RESTRICT a = A + (size_t)r * N;
        for (int j = 0; j < N; ++j) g[j] += a[j];
    }
}

/* -------------------------------------------------------------------------------------------- discriminator */
/* Same contract as orc_disc_grad_rows (oracle/sg_oracle.c): gradient SUM over nb (expert, policy, alpha) triples,
 * flat state_dict order, sums[3] += { expert BCE, policy BCE, (|g|-1)^2 }.  SURVEY.md section 7.1 for the math. */
void fast_disc_grad_rows(int F, int Hd, const float *P, const float *expert_rows, const float *policy_rows,
                         const float *alpha, int nb, float inv_B, float lambda_, float *G, double *sums) {
    const size_t o_w1 = 0, o_b1 = o_w1 + (size_t)Hd * F, o_w2 = o_b1 + Hd, o_b2 = o_w2 + (size_t)Hd * Hd, o_w3 = o_b2 + Hd,
                 o_b3 = o_w3 + Hd;
    const float *W1 = P + o_w1, *b1 = P + o_b1, *W2 = P + o_w2, *b2 = P + o_b2, *w3 = P + o_w3;
    const int R3 = 3 * nb, R2 = 2 * nb;
    float *buf = (float *)malloc(sizeof(float) * ((size_t)F * Hd + (size_t)Hd * Hd + (size_t)R3 * F + (size_t)R3 * Hd * 2 +
                                                  (size_t)R2 * Hd * 2 + (size_t)nb * Hd * 7 + (size_t)nb * F * 2 + R3 + 64));
    float *W1t = buf, *W2t = W1t + (size_t)F * Hd;
    float *X = W2t + (size_t)Hd * Hd;                 /* [3nb,F]: expert | policy | mixup */
    float *H1 = X + (size_t)R3 * F, *H2 = H1 + (size_t)R3 * Hd;
    float *DZ2 = H2 + (size_t)R3 * Hd, *DZ1 = DZ2 + (size_t)R2 * Hd;      /* [2nb,Hd] BCE rows */
    float *D2 = DZ1 + (size_t)R2 * Hd, *U1 = D2 + (size_t)nb * Hd, *D1 = U1 + (size_t)nb * Hd, *BD1 = D1 + (size_t)nb * Hd,
          *BU1 = BD1 + (size_t)nb * Hd, *BD2 = BU1 + (size_t)nb * Hd, *Z2B = BD2 + (size_t)nb * Hd;
    float *GX = Z2B + (size_t)nb * Hd, *GB = GX + (size_t)nb * F;
    float *dd = GB + (size_t)nb * F;                  /* [3nb] logits, then dL/dd */
    transpose(Hd, F, W1, W1t);
    transpose(Hd, Hd, W2, W2t);
    memcpy(X, expert_rows, sizeof(float) * (size_t)nb * F);
    memcpy(X + (size_t)nb * F, policy_rows, sizeof(float) * (size_t)nb * F);
    float *XM = X + (size_t)R2 * F;
    for (int r = 0; r < nb; ++r) {
        const float al = alpha[r];
        const float *e = expert_rows + (size_t)r * F, *p = policy_rows + (size_t)r * F;
        for (int j = 0; j < F; ++j) XM[(size_t)r * F + j] = al * e[j] + (1.0f - al) * p[j];
    }
    /* forward on all 3nb rows */
    gemm_nn(R3, Hd, F, X, F, W1t, Hd, H1, Hd, 0);
    bias_tanh(R3, Hd, H1, b1);
    gemm_nn(R3, Hd, Hd, H1, Hd, W2t, Hd, H2, Hd, 0);
    bias_tanh(R3, Hd, H2, b2);
    double s_e = 0.0, s_p = 0.0, s_g = 0.0;
    float db3 = 0.0f;
    for (int r = 0; r < R2; ++r) {
        const float *h = H2 + (size_t)r * Hd;
        float d = P[o_b3];
        for (int j = 0; j < Hd; ++j) d += h[j] * w3[j];
        const float ls = fminf(d, 0.0f) - log1pf(expf(-fabsf(d))), sg = 1.0f / (1.0f + expf(-d));
        float g;
        if (r < nb) { s_e += (double)(-ls); g = inv_B * (sg - 1.0f); }       /* BCE(D(expert), 1) */
        else { s_p += (double)(d - ls); g = inv_B * sg; }                     /* BCE(D(policy), 0) */
        dd[r] = g;
        db3 += g;
        float *dz = DZ2 + (size_t)r * Hd;
        for (int j = 0; j < Hd; ++j) { G[o_w3 + j] += g * h[j]; dz[j] = g * w3[j] * (1.0f - h[j] * h[j]); }
    }
    G[o_b3] += db3;
    gemm_nn(R2, Hd, Hd, DZ2, Hd, W2, Hd, DZ1, Hd, 0);                         /* dZ2 W2 */
    for (size_t i = 0; i < (size_t)R2 * Hd; ++i) DZ1[i] *= 1.0f - H1[i] * H1[i];
    /* gradient penalty on the mixup rows */
    const float *H1m = H1 + (size_t)R2 * Hd, *H2m = H2 + (size_t)R2 * Hd;
    for (int r = 0; r < nb; ++r)
        for (int j = 0; j < Hd; ++j) { const float h = H2m[(size_t)r * Hd + j]; D2[(size_t)r * Hd + j] = w3[j] * (1.0f - h * h); }
    gemm_nn(nb, Hd, Hd, D2, Hd, W2, Hd, U1, Hd, 0);                           /* u1 = W2^T d2 */
    for (size_t i = 0; i < (size_t)nb * Hd; ++i) D1[i] = U1[i] * (1.0f - H1m[i] * H1m[i]);
    gemm_nn(nb, F, Hd, D1, Hd, W1, F, GX, F, 0);                              /* g = W1^T d1 */
    for (int r = 0; r < nb; ++r) {
        const float *g = GX + (size_t)r * F;
        float nn = 0.0f;
        for (int j = 0; j < F; ++j) nn += g[j] * g[j];
        nn = sqrtf(nn);
        s_g += (double)((nn - 1.0f) * (nn - 1.0f));
        const float c = nn > 0.0f ? lambda_ * 2.0f * inv_B * (nn - 1.0f) / nn : 0.0f;
        for (int j = 0; j < F; ++j) GB[(size_t)r * F + j] = c * g[j];
    }
    gemm_nn(nb, Hd, F, GB, F, W1t, Hd, BD1, Hd, 0);                           /* bd1 = W1 gb */
    for (size_t i = 0; i < (size_t)nb * Hd; ++i) BU1[i] = BD1[i] * (1.0f - H1m[i] * H1m[i]);
    gemm_nn(nb, Hd, Hd, BU1, Hd, W2t, Hd, BD2, Hd, 0);                        /* bd2 = W2 bu1 */
    for (int r = 0; r < nb; ++r)
        for (int j = 0; j < Hd; ++j) {
            const size_t i = (size_t)r * Hd + j;
            const float h = H2m[i], s2 = 1.0f - h * h, v = BD2[i];
            G[o_w3 + j] += v * s2;
            Z2B[i] = (-2.0f * h * (v * w3[j])) * s2;
        }
    float *H1B = BD2;                                                         /* bd2 is dead: reuse */
    gemm_nn(nb, Hd, Hd, Z2B, Hd, W2, Hd, H1B, Hd, 0);                         /* h1b = W2^T z2b */
    float *Z1B = BD1;                                                         /* in place over bd1 */
    for (size_t i = 0; i < (size_t)nb * Hd; ++i) {
        const float h = H1m[i];
        Z1B[i] = (H1B[i] - 2.0f * h * (BD1[i] * U1[i])) * (1.0f - h * h);
    }
    /* weight gradients */
    gemm_tn_acc(Hd, F, R2, DZ1, Hd, X, F, G + o_w1, F);
    gemm_tn_acc(Hd, F, nb, D1, Hd, GB, F, G + o_w1, F);
    gemm_tn_acc(Hd, F, nb, Z1B, Hd, XM, F, G + o_w1, F);
    colsum_acc(R2, Hd, DZ1, G + o_b1);
    colsum_acc(nb, Hd, Z1B, G + o_b1);
    gemm_tn_acc(Hd, Hd, R2, DZ2, Hd, H1, Hd, G + o_w2, Hd);
    gemm_tn_acc(Hd, Hd, nb, D2, Hd, BU1, Hd, G + o_w2, Hd);
    gemm_tn_acc(Hd, Hd, nb, Z2B, Hd, H1m, Hd, G + o_w2, Hd);
    colsum_acc(R2, Hd, DZ2, G + o_b2);
    colsum_acc(nb, Hd, Z2B, G + o_b2);
    sums[0] += s_e; sums[1] += s_p; sums[2] += s_g;
    free(buf);
}

/* --------------------------------------------------------------------------------------------------- PPO */
/* Same contract as orc_ppo_grad_rows: gradient SUM over `rows` (+ loss sums), flat state_dict order. */
void fast_ppo_grad_rows(const orc_policy_dims *d, const float *P, const orc_ppo_cfg *cfg, const float *obs,
                        const float *actions, const float *value_preds, const float *returns, const float *old_logp,
                        const float *adv, const int64_t *rows, int n_rows, float inv_B, float *G, double *sums) {
    const int O = d->O, H = d->H, A = d->A, mlp = d->kind == ORC_KIND_MLP, nt = mlp ? 2 : 3, ct = nt - 1, R = n_rows;
    const int nc = 4 * d->num_feet, na = 3 * d->num_feet;
    const int Hc = d->Hc > 0 ? d->Hc : H, HM = Hc > H ? Hc : H;       /* critic width, widest trunk */
    int Ht[3];
    size_t w1[3], b1[3], w2[3], b2[3], o = 0;
    for (int t = 0; t < nt; ++t) {
        const int h = t == ct ? Hc : H;
        Ht[t] = h;
        w1[t] = o; o += (size_t)h * O; b1[t] = o; o += h; w2[t] = o; o += (size_t)h * h; b2[t] = o; o += h;
    }
    const size_t vw = o, vb = vw + Hc;
    o = vb + 1;
    size_t mw = 0, mb = 0, lsd = 0, cmw = 0, cmb = 0, amw = 0, amb = 0, clw = 0, clb = 0, alw = 0, alb = 0;
    if (mlp) { mw = o; mb = mw + (size_t)A * H; lsd = mb + A; }
    else { cmw = o; cmb = cmw + (size_t)nc * H; amw = cmb + nc; amb = amw + (size_t)na * H; clw = amb + na; clb = clw + (size_t)nc * H;
           alw = clb + nc; alb = alw + (size_t)na * H; }
    const size_t wmax = (size_t)HM * (O > HM ? O : HM);
    float *buf = (float *)malloc(sizeof(float) * (wmax + (size_t)R * O + (size_t)nt * R * HM * 2 + (size_t)R * HM * 2 + (size_t)R * A * 4 + 4 * (size_t)R + 64));
    float *Wt = buf, *X = Wt + wmax, *H1 = X + (size_t)R * O, *H2 = H1 + (size_t)nt * R * HM;   /* trunk t: [R, Ht[t]] at t*R*HM */
    float *DH = H2 + (size_t)nt * R * HM, *DZ = DH + (size_t)R * HM;          /* [R, Ht] each */
    float *MEAN = DZ + (size_t)R * HM, *LS = MEAN + (size_t)R * A, *DM = LS + (size_t)R * A, *DL = DM + (size_t)R * A;
    float *V = DL + (size_t)R * A, *DV = V + R;
    for (int r = 0; r < R; ++r) memcpy(X + (size_t)r * O, obs + (size_t)rows[r] * O, sizeof(float) * O);
    /* forward, trunk by trunk (a2c/model.py:255-264, a2c/model_split.py:187-198) */
    for (int t = 0; t < nt; ++t) {
        const int h = Ht[t];
        float *h1 = H1 + (size_t)t * R * HM, *h2 = H2 + (size_t)t * R * HM;
        transpose(h, O, P + w1[t], Wt);
        gemm_nn(R, h, O, X, O, Wt, h, h1, h, 0);
        bias_tanh(R, h, h1, P + b1[t]);
        transpose(h, h, P + w2[t], Wt);
        gemm_nn(R, h, h, h1, h, Wt, h, h2, h, 0);
        bias_tanh(R, h, h2, P + b2[t]);
    }
    /* heads (small: A <= 28 outputs): dot products per row */
    const float *h2c = H2 + (size_t)ct * R * HM, *h2a = H2, *h2b = H2 + (size_t)R * HM;
    for (int r = 0; r < R; ++r) {
        const float *hc = h2c + (size_t)r * Hc;
        float v = P[vb];
        for (int j = 0; j < Hc; ++j) v += hc[j] * P[vw + j];
        V[r] = v;
        float *mean = MEAN + (size_t)r * A, *ls = LS + (size_t)r * A;
        for (int k = 0; k < A; ++k) {
            const float *h;
            const float *wm, *wl = NULL;
            float bm, bl = 0.0f;
            if (mlp) { h = h2a + (size_t)r * H; wm = P + mw + (size_t)k * H; bm = P[mb + k]; }
            else if (k < nc) { h = h2a + (size_t)r * H; wm = P + cmw + (size_t)k * H; bm = P[cmb + k]; wl = P + clw + (size_t)k * H; bl = P[clb + k]; }
            else { h = h2b + (size_t)r * H; wm = P + amw + (size_t)(k - nc) * H; bm = P[amb + k - nc]; wl = P + alw + (size_t)(k - nc) * H; bl = P[alb + k - nc]; }
            float sm = bm, sl = bl;
            for (int j = 0; j < H; ++j) sm += h[j] * wm[j];
            if (wl) for (int j = 0; j < H; ++j) sl += h[j] * wl[j];
            mean[k] = sm;
            ls[k] = mlp ? P[lsd + k] : sl;
        }
    }
    /* loss and d loss / d heads (a2c/algo/ppo.py:92-106) */
    const float eps = cfg->clip_param;
    double s_v = 0.0, s_a = 0.0, s_e = 0.0;
    for (int r = 0; r < R; ++r) {
        const int64_t idx = rows[r];
        const float *a = actions + (size_t)idx * A, *mean = MEAN + (size_t)r * A, *ls = LS + (size_t)r * A;
        float logp = 0.0f, ent = 0.0f;
        for (int k = 0; k < A; ++k) {
            const float sigma = expf(ls[k]), diff = a[k] - mean[k];
            logp += -(diff * diff) / (2.0f * sigma * sigma) - ls[k] - HALF_LOG_2PI;
            ent += 0.5f + HALF_LOG_2PI + ls[k];
        }
        const float adv_r = adv[idx], Rt = returns[idx], v_old = value_preds[idx], v = V[r];
        const float ratio = expf(logp - old_logp[idx]);
        const float surr1 = ratio * adv_r, rc = fminf(fmaxf(ratio, 1.0f - eps), 1.0f + eps), surr2 = rc * adv_r;
        const float wq = surr1 < surr2 ? 1.0f : (surr1 > surr2 ? 0.0f : 0.5f);
        const float in_range = (ratio >= 1.0f - eps && ratio <= 1.0f + eps) ? 1.0f : 0.0f;
        const float dlogp = -inv_B * (wq * adv_r + (1.0f - wq) * adv_r * in_range) * ratio;
        s_a += (double)(-fminf(surr1, surr2));
        float dv;
        if (cfg->use_clipped_value_loss) {
            const float dvv = v - v_old, vc = v_old + fminf(fmaxf(dvv, -eps), eps);
            const float u = (v - Rt) * (v - Rt), w = (vc - Rt) * (vc - Rt);
            const float m1 = u > w ? 1.0f : (u < w ? 0.0f : 0.5f), pass = (dvv >= -eps && dvv <= eps) ? 1.0f : 0.0f;
            dv = 0.5f * inv_B * (m1 * 2.0f * (v - Rt) + (1.0f - m1) * 2.0f * (vc - Rt) * pass);
            s_v += (double)(0.5f * fmaxf(u, w));
        } else {
            dv = 0.5f * inv_B * (-2.0f) * (Rt - v);
            s_v += (double)(0.5f * (Rt - v) * (Rt - v));
        }
        DV[r] = dv * cfg->value_loss_coef;
        s_e += (double)ent;
        for (int k = 0; k < A; ++k) {
            const float sigma = expf(ls[k]), var = sigma * sigma, diff = a[k] - mean[k];
            DM[(size_t)r * A + k] = dlogp * diff / var;
            DL[(size_t)r * A + k] = dlogp * (diff * diff / var - 1.0f) - cfg->entropy_coef * inv_B;
        }
    }
    sums[0] += s_v; sums[1] += s_a; sums[2] += s_e;
    /* backward, trunk by trunk */
    for (int t = 0; t < nt; ++t) {
        const float *h1 = H1 + (size_t)t * R * HM, *h2 = H2 + (size_t)t * R * HM;
        memset(DH, 0, sizeof(float) * (size_t)R * HM);
        if (t == ct) {
            for (int r = 0; r < R; ++r) {
                const float g = DV[r];
                const float *h = h2 + (size_t)r * Hc;
                float *dh = DH + (size_t)r * Hc;
                G[vb] += g;
                for (int j = 0; j < Hc; ++j) { G[vw + j] += g * h[j]; dh[j] += g * P[vw + j]; }
            }
        } else {
            /* head groups this trunk feeds: (grad rows source, first action dim, count, weight offset, bias offset) */
            const float *src[2] = {DM, DL};
            size_t woff[2], boff[2];
            int k0, cnt, ngr;
            if (mlp) { k0 = 0; cnt = A; ngr = 1; woff[0] = mw; boff[0] = mb; }
            else if (t == 0) { k0 = 0; cnt = nc; ngr = 2; woff[0] = cmw; boff[0] = cmb; woff[1] = clw; boff[1] = clb; }
            else { k0 = nc; cnt = na; ngr = 2; woff[0] = amw; boff[0] = amb; woff[1] = alw; boff[1] = alb; }
            for (int gI = 0; gI < ngr; ++gI)
                for (int r = 0; r < R; ++r) {
                    const float *h = h2 + (size_t)r * H;
                    float *dh = DH + (size_t)r * H;
                    for (int k = 0; k < cnt; ++k) {
                        const float g = src[gI][(size_t)r * A + k0 + k];
                        float *gw = G + woff[gI] + (size_t)k * H;
                        const float *pw = P + woff[gI] + (size_t)k * H;
                        G[boff[gI] + k] += g;
                        for (int j = 0; j < H; ++j) { gw[j] += g * h[j]; dh[j] += g * pw[j]; }
                    }
                }
            if (mlp) for (int r = 0; r < R; ++r) for (int k = 0; k < A; ++k) G[lsd + k] += DL[(size_t)r * A + k];
        }
        const int h = Ht[t];
        for (size_t i = 0; i < (size_t)R * h; ++i) DZ[i] = DH[i] * (1.0f - h2[i] * h2[i]);
        gemm_tn_acc(h, h, R, DZ, h, h1, h, G + w2[t], h);
        colsum_acc(R, h, DZ, G + b2[t]);
        gemm_nn(R, h, h, DZ, h, P + w2[t], h, DH, h, 0);                      /* dh1 = dz2 W2 */
        for (size_t i = 0; i < (size_t)R * h; ++i) DZ[i] = DH[i] * (1.0f - h1[i] * h1[i]);
        gemm_tn_acc(h, O, R, DZ, h, X, O, G + w1[t], O);
        colsum_acc(R, h, DZ, G + b1[t]);
    }
    free(buf);
}

/* ------------------------------------------------------------------------------------------------ all cores */
/* Rows split into n_threads contiguous chunks, each chunk through the batched routine into a private gradient, chunks
 * summed in chunk order (the all-cores leg of bench.py's cpu_baseline). */
void fast_ppo_grad_rows_mt(const orc_policy_dims *d, const float *P, const orc_ppo_cfg *cfg, const float *obs,
                           const float *actions, const float *value_preds, const float *returns, const float *old_logp,
                           const float *adv, const int64_t *rows, int n_rows, float inv_B, float *G, double *sums,
                           int n_threads, int64_t n_params) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads > n_rows) n_threads = n_rows;
    float *Gt = (float *)calloc((size_t)n_threads * n_params, sizeof(float));
    double *St = (double *)calloc((size_t)n_threads * 3, sizeof(double));
#pragma omp parallel for num_threads(n_threads) schedule(static, 1)
    for (int t = 0; t < n_threads; ++t) {
        const int lo = (int)((int64_t)n_rows * t / n_threads), hi = (int)((int64_t)n_rows * (t + 1) / n_threads);
        fast_ppo_grad_rows(d, P, cfg, obs, actions, value_preds, returns, old_logp, adv, rows + lo, hi - lo, inv_B,
                           Gt + (size_t)t * n_params, St + 3 * t);
    }
    for (int t = 0; t < n_threads; ++t) {
        for (int64_t i = 0; i < n_params; ++i) G[i] += Gt[(size_t)t * n_params + i];
        for (int i = 0; i < 3; ++i) sums[i] += St[3 * t + i];
    }
    free(Gt);
    free(St);
}

void fast_disc_grad_rows_mt(int F, int Hd, const float *P, const float *expert_rows, const float *policy_rows,
                            const float *alpha, int nb, float inv_B, float lambda_, float *G, double *sums, int n_threads) {
    const int64_t n = (int64_t)Hd * F + Hd + (int64_t)Hd * Hd + Hd + Hd + 1;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > nb) n_threads = nb;
    float *Gt = (float *)calloc((size_t)n_threads * n, sizeof(float));
    double *St = (double *)calloc((size_t)n_threads * 3, sizeof(double));
#pragma omp parallel for num_threads(n_threads) schedule(static, 1)
    for (int t = 0; t < n_threads; ++t) {
        const int lo = (int)((int64_t)nb * t / n_threads), hi = (int)((int64_t)nb * (t + 1) / n_threads);
        fast_disc_grad_rows(F, Hd, P, expert_rows + (size_t)lo * F, policy_rows + (size_t)lo * F, alpha + lo, hi - lo, inv_B,
                            lambda_, Gt + (size_t)t * n, St + 3 * t);
    }
    for (int t = 0; t < n_threads; ++t) {
        for (int64_t i = 0; i < n; ++i) G[i] += Gt[(size_t)t * n + i];
        for (int i = 0; i < 3; ++i) sums[i] += St[3 * t + i];
    }
    free(Gt);
    free(St);
}
