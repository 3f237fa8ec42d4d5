// sg_disc.hip -- GAIL discriminator: BCE-with-logits + gradient-penalty step (with the
// hand-derived double backward), Adam, reward prediction and the fused reward relabel.
//
// Replaces (reference, a2c/ = third_party/a2c_ppo_acktr/):
//   Discriminator.__init__                    a2c/algo/gail.py:35-51
//   Discriminator.compute_grad_pen_combined   a2c/algo/gail.py:67-89
//   Discriminator.update_gail_dyn             a2c/algo/gail.py:154-193
//   Discriminator.predict_reward_combined     a2c/algo/gail.py:201-210
//   reward relabel + RunningMeanStd           a2c/main_gail_dyn_ppo.py:275-292,
//                                             a2c/baselines/common/running_mean_std.py:27-58
//
// One optimizer step = two launches queued back to back (n_d * gail_epoch of them per update):
//   k_disc_chain   2*G workgroups, G = ceil(batch/16).  Workgroups [0,G) take 16 expert + 16 policy
//                 rows through forward/BCE/backward; workgroups [G,2G) take the 16 matching mixup
//                 rows through forward, input-gradient, penalty and the double backward.  The whole
//                 parameter vector (its HBM image is the LDS image) is staged once per workgroup;
//                 every contraction is an LDS-tile fp32 MFMA GEMM.  It writes the left/right factors
//                 of every weight-gradient outer product ("operand stacks", 512 stacked rows per
//                 layer) plus per-workgroup bias / loss partials.
//   k_disc_wgrad  one workgroup per 16x16 weight tile: TN GEMM over the stacked rows + Adam in place
//                 (no clipping for D); one block for biases, loss sums and the optimizer scalars;
//                 2*G blocks gather the NEXT step's rows into the other parity's operand stack.
//
// Gradient-penalty math (x = mixup row, s_i = 1 - h_i^2, lambda = 10, B = batch):
//   d2 = w3*s2; u1 = W2^T d2; d1 = u1*s1; g = W1^T d1; n = |g|; gb = lambda*(2/B)*(n-1)/n * g
//   dW1 += d1 gb^T; bd1 = W1 gb; bu1 = bd1*s1; sb1 = bd1*u1; dW2 += d2 bu1^T; bd2 = W2 bu1
//   dw3 += bd2*s2; sb2 = bd2*w3; z2b = (-2 h2 sb2)*s2; dW2 += z2b h1^T; db2 += z2b
//   h1b = W2^T z2b - 2 h1 sb1; z1b = h1b*s1; dW1 += z1b x^T; db1 += z1b
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <vector>

#include "sg_common.h"
#include "sg_rng.hpp"

int sg_fill_perm(sg_ctx* ctx, int64_t* d_perm, int64_t n, uint64_t seed, uint64_t stream_id);

#include "sg_disc_kernels.hpp"
#include "sg_disc_step4.hpp"

// shape-specialised instances: north-star / Laikago (F 86, Hd 100), Hopper (F 25, Hd 100), the
// tiny test shape, and the run-time-shape fallback
// The LDS-resident chain / forward kernels need the whole parameter image beside their activation tiles; a larger
// discriminator takes the global-weight instances (SG_DISC_GW=1 forces them for any shape: tests).
static bool disc_needs_gw(const sg_ctx* ctx, const SgDiscDesc& dd) {
    const char* e = getenv("SG_DISC_GW");
    if (e && e[0] == '1') return true;
    return disc_chain_lds_bytes(dd) > (size_t)ctx->lds_bytes;
}
static void launch_disc_chain(sg_ctx* ctx, const SgDiscDesc& dd, dim3 grid, size_t lds, const DiscArgs& a, bool gw) {
    const int kf = dd.Fp / 16, kh = dd.Hp / 16;
    const dim3 block(SG_DISC_THREADS);
    if (gw) { SG_LAUNCH(ctx, SG_PROF_DISC_CHAIN, (k_disc_chain<0, 0, true>), grid, block, lds, a); return; }
    if (kf == 6 && kh == 7) SG_LAUNCH(ctx, SG_PROF_DISC_CHAIN, (k_disc_chain<6, 7>), grid, block, lds, a);
    else if (kf == 2 && kh == 7) SG_LAUNCH(ctx, SG_PROF_DISC_CHAIN, (k_disc_chain<2, 7>), grid, block, lds, a);
    else if (kf == 1 && kh == 1) SG_LAUNCH(ctx, SG_PROF_DISC_CHAIN, (k_disc_chain<1, 1>), grid, block, lds, a);
    else SG_LAUNCH(ctx, SG_PROF_DISC_CHAIN, (k_disc_chain<0, 0>), grid, block, lds, a);
}

// The 4-row kernel exists for the shipped shapes (everything in it is a compile-time extent);
// other shapes, or SG_DISC_CHAIN=wide, take the 16-row kernel above.
static bool disc_chain_thin(const sg_ctx* ctx, const SgDiscDesc& dd) {
    const char* e = getenv("SG_DISC_CHAIN");
    if (e && !strcmp(e, "wide")) return false;
    if (disc_needs_gw(ctx, dd)) return false;
    const int kf = dd.Fp / 16, kh = dd.Hp / 16;
    return (kf == 6 && kh == 7) || (kf == 2 && kh == 7) || (kf == 1 && kh == 1);
}
static void launch_disc_chain4(sg_ctx* ctx, const SgDiscDesc& dd, dim3 grid, const DiscArgs& a, const PregatherArgs& next) {
    const int kf = dd.Fp / 16, kh = dd.Hp / 16;
    const dim3 block(64 * (kf > kh ? kf : kh));   // one wave per 16 output columns: no idle wave to launch and drain
#define SG_CHAIN4(KF_, KH_)                                                                                         \
    SG_LAUNCH(ctx, SG_PROF_DISC_CHAIN, (k_disc_chain4<KF_, KH_>), grid, block, 0, a.params, a.wT, a.ops, a.part, a.dbg, \
              a.B, a.G, a.inv_B, a.lambda_, next)
    if (kf == 6 && kh == 7) SG_CHAIN4(6, 7);
    else if (kf == 2 && kh == 7) SG_CHAIN4(2, 7);
    else SG_CHAIN4(1, 1);
#undef SG_CHAIN4
}
// One launch per step (sg_disc_step4.hpp) for the shapes the 4-row kernel exists for.
static void launch_disc_step4(sg_ctx* ctx, const SgDiscDesc& dd, sg_disc* d, float* ops, int B, int G, int k1, const Step4Args& sa) {
    const int kf = dd.Fp / 16, kh = dd.Hp / 16;
    const dim3 grid(((12 * G + 7) & ~7) + 8 * (kh + kf) + 2 * G), block(512);
    SgOptState* st = reinterpret_cast<SgOptState*>(d->d_state);
#define SG_STEP4(KF_, KH_)                                                                                                   \
    SG_LAUNCH(ctx, SG_PROF_DISC_STEP, (k_disc_step4<KF_, KH_>), grid, block, 0, d->d_params, d->d_m, d->d_v, d->d_wT, ops, st, \
              sg_wgrad_pack(G, 0, k1), B, sa)
    if (kf == 6 && kh == 7) SG_STEP4(6, 7);
    else if (kf == 2 && kh == 7) SG_STEP4(2, 7);
    else SG_STEP4(1, 1);
#undef SG_STEP4
}
unsigned* sg_disc_err_word(sg_disc* d) { return reinterpret_cast<unsigned*>(d->d_state) + SG_STEP4_ERR_WORD; }
static void disc_refresh_images(sg_disc* d) {
    hipLaunchKernelGGL(k_disc_images, dim3(32), dim3(256), 0, d->ctx->stream, d->desc, d->d_params, d->d_wT);
}

__global__ void k_fill_alpha(float* alpha, int64_t n, uint64_t seed, uint64_t stream) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) alpha[i] = sg_uniform(seed, stream, (uint64_t)i);
}

// Discriminator.compute_grad_pen_combined (a2c/algo/gail.py:67-89), the VALUE of the penalty only: per row
//   x = alpha e + (1 - alpha) p;  g = dD/dx(x) = W1^T[(1 - h1^2) . W2^T[(1 - h2^2) . w3]];  pen = (||g||_2 - 1)^2.
// Not on the update path (the update kernels form the penalty and its double backward themselves): one workgroup per row,
// plain FMAs from the canonical parameter layout.
__global__ void __launch_bounds__(128) k_disc_grad_pen(SgDiscDesc d, const float* __restrict__ W, const float* __restrict__ e,
                                                       const float* __restrict__ p, const float* __restrict__ alpha, int n,
                                                       float* __restrict__ pen) {
    extern __shared__ float sm[];
    float* xm = sm;                 // [Fp]
    float* h1 = xm + d.Fp;          // [Hp]
    float* h2 = h1 + d.Hp;
    float* d2 = h2 + d.Hp;
    float* d1 = d2 + d.Hp;
    float* red = d1 + d.Hp;         // [2]
    const int r = blockIdx.x, t = threadIdx.x, F = d.F, Hd = d.Hd;
    if (r >= n) return;
    const float al = alpha[r];
    for (int j = t; j < F; j += blockDim.x) xm[j] = al * e[(size_t)r * F + j] + (1.0f - al) * p[(size_t)r * F + j];
    __syncthreads();
    for (int i = t; i < Hd; i += blockDim.x) {
        float a = W[d.b1 + i];
        const float* w = W + d.w1 + (size_t)i * d.ldF;
        for (int j = 0; j < F; ++j) a += w[j] * xm[j];
        h1[i] = sg_tanh(a);
    }
    __syncthreads();
    for (int i = t; i < Hd; i += blockDim.x) {
        float a = W[d.b2 + i];
        const float* w = W + d.w2 + (size_t)i * d.ldH;
        for (int k = 0; k < Hd; ++k) a += w[k] * h1[k];
        const float h = sg_tanh(a);
        h2[i] = h;
        d2[i] = W[d.w3 + i] * (1.0f - h * h);
    }
    __syncthreads();
    for (int k = t; k < Hd; k += blockDim.x) {
        float a = 0.f;
        for (int i = 0; i < Hd; ++i) a += W[d.w2 + (size_t)i * d.ldH + k] * d2[i];
        d1[k] = a * (1.0f - h1[k] * h1[k]);
    }
    __syncthreads();
    float part = 0.f;
    for (int j = t; j < F; j += blockDim.x) {
        float a = 0.f;
        for (int i = 0; i < Hd; ++i) a += W[d.w1 + (size_t)i * d.ldF + j] * d1[i];
        part += a * a;
    }
    for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o, 64);
    if ((t & 63) == 0) red[t >> 6] = part;
    __syncthreads();
    if (t == 0) {
        const float nn = sqrtf(red[0] + red[1]);
        pen[r] = (nn - 1.0f) * (nn - 1.0f);
    }
}

// ------------------------------------------------------------------------ forward / rewards

struct DiscFwdArgs {
    SgDiscDesc d;
    const float* params;
    const float* x;   // [n, F]
    int n;
    float offset;
    const double* neg_offset_dev;   // != NULL: offset = -(float)*neg_offset_dev (the alive-bonus r_sa computed on the device)
    float* reward;    // [n]  log(s+1e-7) - log(1-s+1e-7) + offset   (prob != 0: sigmoid(D(x)), a2c/algo/gail.py:212-217)
    int prob;
};

static size_t disc_fwd_lds_bytes(const SgDiscDesc& d, bool gw = false) {
    return sizeof(float) * ((gw ? 0 : (size_t)d.total) + 32 * d.ldF + 2 * 32 * d.ldH);
}

template <bool GW>
__global__ __launch_bounds__(256) void k_disc_forward(DiscFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int R = 32;
    const SgDiscDesc& d = a.d;
    const int tid = threadIdx.x, ldF = d.ldF, ldH = d.ldH, Fp = d.Fp, Hp = d.Hp;
    const float* W = GW ? a.params : smem;
    float* X = smem + (GW ? 0 : d.total);
    float* H1 = X + R * ldF;
    float* H2 = H1 + R * ldH;
    const float* b1 = W + d.b1;
    const float* b2 = W + d.b2;
    const float* w3 = W + d.w3;
    if (!GW) sg_stage(smem, a.params, d.total / 4);
    const float offset = a.neg_offset_dev ? -(float)(*a.neg_offset_dev) : a.offset;
    for (int base = blockIdx.x * R; base < a.n; base += gridDim.x * R) {
        __syncthreads();
        {   // the row tile through range-checked buffer loads, all requests of a thread in flight together: rows past
            // n and the padding columns [F, Fp) read as zero without a guard the compiler would turn into a branch
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(a.x) + (size_t)base * d.F, 0, (a.n - base < R ? a.n - base : R) * d.F * 4, 0x00020000);
            for (int i0 = tid; i0 < R * Fp; i0 += 12 * blockDim.x) {
                float v[12];
#pragma unroll
                for (int u = 0; u < 12; ++u) {
                    const int i = i0 + u * blockDim.x, r = i / Fp, c = i - r * Fp;
                    const int off = (i < R * Fp && c < d.F) ? (r * d.F + c) * 4 : 0x7ffffff0;
                    v[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, off, 0, 0));
                }
#pragma unroll
                for (int u = 0; u < 12; ++u) {
                    const int i = i0 + u * blockDim.x, r = i / Fp, c = i - r * Fp;
                    if (i < R * Fp) X[r * ldF + c] = v[u];
                }
            }
        }
        __syncthreads();
        sg_layer_nt<2, GW>(X, ldF, W + d.w1, ldF, Fp, Hp, [&](int r, int c, float v) { H1[r * ldH + c] = sg_tanh(v + b1[c]); });
        __syncthreads();
        sg_layer_nt<2, GW>(H1, ldH, W + d.w2, ldH, Hp, Hp, [&](int r, int c, float v) { H2[r * ldH + c] = sg_tanh(v + b2[c]); });
        __syncthreads();
        const int r = tid >> 3, sub = tid & 7;
        float s = 0.f;
        for (int c = sub; c < Hp; c += 8) s += H2[r * ldH + c] * w3[c];
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
        if (sub == 0 && base + r < a.n) {
            const float sg = sg_sigmoid(s + W[d.b3]);
            a.reward[base + r] = a.prob ? sg : logf(sg + 1e-7f) - logf(1.f - sg + 1e-7f) + offset;  // a2c/algo/gail.py:204-205
        }
    }
}

// returns = returns*gamma*masks + reward (first call: returns = reward)   a2c/algo/gail.py:206-209
__global__ void k_returns_step(float* returns, const float* reward, const float* masks, float gamma, int first, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) returns[i] = first ? reward[i] : returns[i] * gamma * masks[i] + reward[i];
}

// Per-column scan over T of the same recurrence; keeps every step's returns for the statistics.
__global__ void k_returns_scan(float* d_returns, const float* raw /*[T,N]*/, const float* masks /*[T+1,N]*/,
                               float gamma, int first, int T, int N, float* rets /*[T,N]*/) {
    constexpr int C = 8;   // steps fetched with independent loads before their (serial) updates
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float ret = first ? 0.f : d_returns[n];
    for (int t0 = 0; t0 < T; t0 += C) {
        float rw[C], mk[C];
#pragma unroll
        for (int u = 0; u < C; ++u) {
            const int t = t0 + u < T ? t0 + u : T - 1;
            rw[u] = raw[(size_t)t * N + n];
            mk[u] = masks[(size_t)t * N + n];
        }
#pragma unroll
        for (int u = 0; u < C; ++u) {
            const int t = t0 + u;
            if (t >= T) break;
            ret = (first && t == 0) ? rw[u] : ret * gamma * mk[u] + rw[u];
            rets[(size_t)t * N + n] = ret;
        }
    }
    d_returns[n] = ret;
}

// per-step batch sums (pass 0) and sums of squares about the batch mean (pass 1); one block per t
__global__ __launch_bounds__(256) void k_batch_stats(const float* rets, int N, double n_global, double* stats /*[2][T]*/, int pass) {
    __shared__ double ws[4];
    const int t = blockIdx.x;
    const float* x = rets + (size_t)t * N;
    double s = 0.0;
    if (pass == 0) {
        for (int i = threadIdx.x; i < N; i += blockDim.x) s += (double)x[i];
    } else {
        const float mean = (float)(stats[t] / n_global);   // numpy: float32 batch mean
        for (int i = threadIdx.x; i < N; i += blockDim.x) { const float dd = x[i] - mean; s += (double)(dd * dd); }
    }
    s = sg_wave_sum(s);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) stats[(size_t)pass * gridDim.x + t] = ws[0] + ws[1] + ws[2] + ws[3];
}

// Sequential Chan merge over t (float64 state) -> per-step scale = sqrt(var_t + 1e-7)
// RunningMeanStd.update over the T batches in order (float64 Chan merge, running_mean_std.py:27-58): serial, so one
// lane runs it -- from LDS, where the whole block first put the 2T batch statistics with parallel loads.
__global__ __launch_bounds__(256) void k_rms_scan(const double* stats, int T, double n_global, double* rms /*[3] in/out*/,
                                                  float* scale /*[T]*/) {
    extern __shared__ __attribute__((aligned(16))) double sst[];   // [2T] statistics, then [T] scales
    for (int i = threadIdx.x; i < 2 * T; i += blockDim.x) sst[i] = stats[i];
    __syncthreads();
    if (threadIdx.x == 0) {
        double mean = rms[0], var = rms[1], count = rms[2];
        for (int t = 0; t < T; ++t) {
            const double bmean = (double)(float)(sst[t] / n_global);
            const double bvar = (double)(float)(sst[T + t] / n_global);
            const double delta = bmean - mean, tot = count + n_global;
            const double new_mean = mean + delta * n_global / tot;
            const double M2 = var * count + bvar * n_global + delta * delta * count * n_global / tot;
            mean = new_mean; var = M2 / tot; count = tot;
            sst[2 * T + t] = sqrt(var + 1e-7);
        }
        rms[0] = mean; rms[1] = var; rms[2] = count;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += blockDim.x) scale[t] = (float)sst[2 * T + t];
}

__global__ void k_normalize_rewards(float* rewards, const float* scale, int T, int N) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)T * N) return;
    const float v = rewards[i] / scale[i / N];
    rewards[i] = fminf(fmaxf(v, -10.f), 10.f);
}

// --------------------------------------------------------------------------------------- API

extern "C" int sg_disc_create(sg_ctx* ctx, int input_dim, int hidden_dim, sg_disc** out) {
    SG_DEVICE_WIDE();
    SG_REQUIRE(ctx && out, "sg_disc_create: NULL argument");
    SG_REQUIRE(input_dim > 0 && hidden_dim > 0, "sg_disc_create: bad dims");
    SG_CHECK(hipSetDevice(ctx->device));
    sg_disc* d = new sg_disc();
    d->ctx = ctx;
    d->desc = sg_make_disc_desc(input_dim, hidden_dim);
    // any (input_dim, hidden_dim) the reference's constructor accepts (a2c/algo/gail.py:40-43): a parameter image that fits a
    // CU's LDS runs on the LDS-resident kernels, a larger one on the global-weight instances; only the activation tiles
    // of a 16-row block have to fit
    SG_REQUIRE(disc_chain_lds_bytes(d->desc, true) <= (size_t)ctx->lds_bytes,
               "sg_disc_create: the activation tiles of a (%d x %d) discriminator need %zu bytes of LDS, the CU has %d", input_dim,
               hidden_dim, disc_chain_lds_bytes(d->desc, true), ctx->lds_bytes);
    const size_t tot = d->desc.total;
    SG_CHECK(sg_dev_malloc((void**)&d->d_params, sizeof(float) * tot));
    SG_CHECK(sg_dev_malloc((void**)&d->d_m, sizeof(float) * tot));
    SG_CHECK(sg_dev_malloc((void**)&d->d_v, sizeof(float) * tot));
    static_assert(sizeof(SgOptState) <= 4 * SG_STEP4_FLAG_WORD0, "the hand-off flags of k_disc_step4 sit behind the optimizer state");
    SG_CHECK(sg_dev_malloc((void**)&d->d_state, SG_STEP4_STATE_BYTES));   // SgOptState | hand-off flags | error word (k_disc_step4)
    SG_CHECK(hipMemsetAsync(d->d_state, 0, SG_STEP4_STATE_BYTES, ctx->stream));
    SG_CHECK(sg_dev_malloc((void**)&d->d_loss_acc, sizeof(double) * 8));
    SG_CHECK(sg_dev_malloc((void**)&d->d_scal, sizeof(double) * 8));
    {
        const double scal0[8] = {0.0, 1.0, 1e-4, 0.0, 0.0, 0.0, 0.0, 0.0};   // RunningMeanStd(): mean 0, var 1, count 1e-4
        SG_CHECK(hipMemcpyAsync(d->d_scal, scal0, sizeof scal0, hipMemcpyHostToDevice, ctx->stream));
        SG_CHECK(hipStreamSynchronize(ctx->stream));
    }
    const size_t wT_f = (size_t)2 * d->desc.Hp * (d->desc.Fp + d->desc.Hp);   // images of W1, W2, W2^T, W1^T
    SG_CHECK(sg_dev_malloc((void**)&d->d_wT, sizeof(float) * wT_f));
    SG_CHECK(hipMemsetAsync(d->d_wT, 0, sizeof(float) * wT_f, ctx->stream));
    SG_CHECK(hipMemsetAsync(d->d_params, 0, sizeof(float) * tot, ctx->stream));
    SG_CHECK(hipMemsetAsync(d->d_m, 0, sizeof(float) * tot, ctx->stream));
    SG_CHECK(hipMemsetAsync(d->d_v, 0, sizeof(float) * tot, ctx->stream));
    SgOptState st;
    memset(&st, 0, sizeof st);
    st.lr = 1e-3f;
    SG_CHECK(hipMemcpyAsync(d->d_state, &st, sizeof st, hipMemcpyHostToDevice, ctx->stream));
    SG_CHECK(hipStreamSynchronize(ctx->stream));
    sg_ctx_learner_born(ctx);
#if SG_STEP4_VERIFY
    SG_CHECK(sg_dev_malloc((void**)&d->d_dbg_step4, sizeof(long long) * 8 * 512 + sizeof(float) * 128 * 8 * 32 * 64 + 4 * 96 * 8 * 8 * 64 + 8 * 4 * 512));
    SG_CHECK(hipMemset(d->d_dbg_step4, 0, sizeof(long long) * 8 * 512 + sizeof(float) * 128 * 8 * 32 * 64 + 4 * 96 * 8 * 8 * 64 + 8 * 4 * 512));
#endif
    *out = d;
    return 0;
}

#if SG_STEP4_VERIFY
// debug builds only (not in include/simgan_hip.h): the operand words the tile workgroups of the last step consumed
extern "C" SG_API int sg_debug_step4_log(sg_disc* d, float* out, long long n_floats) {
    SG_CHECK(hipStreamSynchronize(d->ctx->stream));
    SG_CHECK(hipMemcpy(out, reinterpret_cast<float*>(d->d_dbg_step4 + 8 * 512), sizeof(float) * n_floats, hipMemcpyDeviceToHost));
    return 0;
}
extern "C" SG_API int sg_debug_step4_stamps(sg_disc* d, long long* out) {   // [512][4]
    SG_CHECK(hipStreamSynchronize(d->ctx->stream));
    SG_CHECK(hipMemcpy(out, reinterpret_cast<unsigned*>(reinterpret_cast<float*>(d->d_dbg_step4 + 8 * 512) + (size_t)128 * 8 * 32 * 64) + 96 * 8 * 8 * 64, 8 * 4 * 512, hipMemcpyDeviceToHost));
    return 0;
}
extern "C" SG_API int sg_debug_step4_chainlog(sg_disc* d, unsigned* out) {   // [96][8][8][64]
    SG_CHECK(hipStreamSynchronize(d->ctx->stream));
    SG_CHECK(hipMemcpy(out, reinterpret_cast<float*>(d->d_dbg_step4 + 8 * 512) + (size_t)128 * 8 * 32 * 64, 4 * 96 * 8 * 8 * 64, hipMemcpyDeviceToHost));
    return 0;
}
#endif
extern "C" int sg_disc_destroy(sg_disc* d) {
    SG_DEVICE_WIDE();
    if (!d) return 0;
    (void)hipStreamSynchronize(d->ctx->stream);
#if SG_STEP4_VERIFY
    if (d->d_dbg_step4) {
        std::vector<long long> h(8 * 512);
        (void)hipMemcpy(h.data(), d->d_dbg_step4, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
        fprintf(stderr, "[step4 verify] %lld operand words differed on re-read (steps so far %lld)\n", h[0], (long long)d->opt_t);
        for (long long i = 0; i < h[0] && i < 100; ++i) {
            const long long* o = h.data() + 8 + 4 * i;
            fprintf(stderr, "  block %lld wave %lld lane %lld half %lld cc %lld s %lld side %lld: consumed %08llx re-read %08llx step %lld off %lld\n",
                    o[0] >> 32, (o[0] >> 16) & 0xffff, (o[0] >> 8) & 0xff, (o[0] >> 4) & 1, (o[0] >> 3) & 1, (o[0] >> 1) & 3, o[0] & 1,
                    (unsigned long long)o[1] >> 32, (unsigned long long)o[1] & 0xffffffffull, o[2], o[3]);
        }
        (void)sg_dev_free(d->d_dbg_step4);
        d->d_dbg_step4 = nullptr;
    }
#endif
    sg_ctx_learner_gone(d->ctx);
    for (auto& q : d->ctx->res_d) if (q == d) q = nullptr;
    float* ptrs[] = {d->d_params, d->d_m, d->d_v, d->d_slabs, d->d_state, d->d_expert, d->d_alpha, d->d_returns, d->d_feat_all, d->d_rows, d->d_wT, d->d_erows, d->d_prows};
    for (float* q : ptrs) if (q) (void)sg_dev_free(q);
    if (d->d_eperm) (void)sg_dev_free(d->d_eperm);
    if (d->d_pperm) (void)sg_dev_free(d->d_pperm);
    if (d->d_loss_acc) (void)sg_dev_free(d->d_loss_acc);
    if (d->d_scal) (void)sg_dev_free(d->d_scal);
    if (d->epoch_graph) (void)hipGraphExecDestroy(d->epoch_graph);
    delete d;
    return 0;
}

extern "C" int sg_disc_num_params(const sg_disc* d, int64_t* n) {
    SG_REQUIRE(d && n, "sg_disc_num_params: NULL argument");
    *n = sg_disc_flat_count(d->desc);
    return 0;
}

static int disc_put(sg_disc* d, float* dev, const float* flat, int64_t n, const char* who) {
    SG_REQUIRE(n == sg_disc_flat_count(d->desc), "%s: expected %lld floats, got %lld", who,
               (long long)sg_disc_flat_count(d->desc), (long long)n);
    std::vector<float> padded(d->desc.total, 0.f);
    sg_disc_pad(d->desc, flat, padded.data());
    SG_CHECK(hipStreamSynchronize(d->ctx->stream));
    SG_COPY_SYNC(d->ctx, dev, padded.data(), sizeof(float) * padded.size(), hipMemcpyHostToDevice);
    return 0;
}
static int disc_get(sg_disc* d, const float* dev, float* flat, int64_t n, const char* who) {
    SG_REQUIRE(n == sg_disc_flat_count(d->desc), "%s: expected %lld floats, got %lld", who,
               (long long)sg_disc_flat_count(d->desc), (long long)n);
    std::vector<float> padded(d->desc.total);
    SG_CHECK(hipStreamSynchronize(d->ctx->stream));
    SG_COPY_SYNC(d->ctx, padded.data(), dev, sizeof(float) * padded.size(), hipMemcpyDeviceToHost);
    sg_disc_unpad(d->desc, padded.data(), flat);
    return 0;
}

extern "C" int sg_disc_set_params(sg_disc* d, const float* flat, int64_t n) {
    SG_REQUIRE(d && flat, "sg_disc_set_params: NULL argument");
    SG_TRY(disc_put(d, d->d_params, flat, n, "sg_disc_set_params"));
    disc_refresh_images(d);
    SG_CHECK(hipGetLastError());
    return 0;
}
extern "C" int sg_disc_get_params(sg_disc* d, float* flat, int64_t n) {
    SG_REQUIRE(d && flat, "sg_disc_get_params: NULL argument");
    return disc_get(d, d->d_params, flat, n, "sg_disc_get_params");
}
extern "C" int sg_disc_get_adam(sg_disc* d, float* m, float* v, int64_t n, int64_t* step) {
    SG_REQUIRE(d && m && v && step, "sg_disc_get_adam: NULL argument");
    SG_TRY(disc_get(d, d->d_m, m, n, "sg_disc_get_adam"));
    SG_TRY(disc_get(d, d->d_v, v, n, "sg_disc_get_adam"));
    *step = d->opt_t;
    return 0;
}
extern "C" int sg_disc_set_adam(sg_disc* d, const float* m, const float* v, int64_t n, int64_t step) {
    SG_REQUIRE(d && m && v, "sg_disc_set_adam: NULL argument");
    SG_TRY(disc_put(d, d->d_m, m, n, "sg_disc_set_adam"));
    SG_TRY(disc_put(d, d->d_v, v, n, "sg_disc_set_adam"));
    SG_REQUIRE(step >= 0 && step < (1ll << 30), "sg_disc_set_adam: step out of range");
    d->opt_t = step;
    const int t0 = (int)step;
    SG_COPY_SYNC(d->ctx, &reinterpret_cast<SgOptState*>(d->d_state)->t0, &t0, sizeof t0, hipMemcpyHostToDevice);
    // the hand-off flags of k_disc_step4 hold Adam step numbers: a step count set from outside may repeat old ones
    SG_CHECK(hipMemsetAsync(reinterpret_cast<unsigned*>(d->d_state) + SG_STEP4_FLAG_WORD0, 0, 4 * SG_STEP4_MAX_FLAGS * SG_STEP4_FLAG_STRIDE, d->ctx->stream));
    SG_CHECK(hipStreamSynchronize(d->ctx->stream));
    return 0;
}

extern "C" int sg_disc_set_expert(sg_disc* d, const float* expert, int64_t n_rows) {
    SG_REQUIRE(d && expert && n_rows > 0, "sg_disc_set_expert: bad argument");
    SG_CHECK(hipStreamSynchronize(d->ctx->stream));
    if (d->d_expert) SG_CHECK(sg_dev_free(d->d_expert));
    const size_t bytes = sizeof(float) * (size_t)n_rows * d->desc.F;
    SG_CHECK(sg_dev_malloc((void**)&d->d_expert, bytes));
    SG_COPY_SYNC(d->ctx, d->d_expert, expert, bytes, hipMemcpyHostToDevice);
    d->n_expert = n_rows;
    return 0;
}

template <typename T>
static int ensure_cap(T** ptr, int64_t* cap, int64_t need, hipStream_t stream) {
    if (*cap >= need) return 0;
    SG_CHECK(hipStreamSynchronize(stream));
    if (*ptr) SG_CHECK(sg_dev_free(*ptr));
    SG_CHECK(sg_dev_malloc((void**)ptr, sizeof(T) * (size_t)need));
    *cap = need;
    return 0;
}

// Injected draws arrive as host arrays with explicit element counts: lengths are checked against what the epoch
// consumes and every index against the range it addresses BEFORE anything is copied to the device (a short array would be
// a host out-of-bounds read, a bad index an unchecked device gather).
static int check_index_array(const char* who, const char* name, const int64_t* a, int64_t n, int64_t need, int64_t limit) {
    SG_REQUIRE(n == need, "%s: %s holds %lld indices, the epoch needs exactly %lld", who, name, (long long)n, (long long)need);
    for (int64_t i = 0; i < n; ++i)
        SG_REQUIRE(a[i] >= 0 && a[i] < limit, "%s: %s[%lld] = %lld is outside [0, %lld)", who, name, (long long)i, (long long)a[i], (long long)limit);
    return 0;
}

// One epoch of discriminator steps on `rows_local` (device, [TN_loc, F]: this rank's policy rows; n_cols = environment
// columns per time slot when the rows are a rollout's (t, n) grid, 0 when they are unstructured).
//
// Row numbering for world > 1.  The reference is single-process: at num_processes = world * N its flattened row id is
// t * (world*N) + n_global (a2c/storage.py:168-185), with rank r owning columns [r*N, (r+1)*N).  Every INJECTED policy
// permutation is in that global numbering, in both data-parallel modes, so the reference's own draws at the global size
// reproduce the reference's result; the library's own generator draws in the same numbering in replicated mode and, in
// sharded mode, one local permutation per rank (each rank contributes batch/world rows to every step).
static int disc_update_core(sg_disc* d, const float* rows_local, int64_t TN_loc, int n_cols, int batch_size,
                            const int64_t* expert_perm, int64_t n_expert_perm, const int64_t* policy_perm, int64_t n_policy_perm,
                            const float* alpha, int64_t n_alpha, uint64_t seed, float out3[3], int* n_steps, uint64_t src_version = 0) {
    sg_ctx* ctx = d->ctx;
    const SgDiscDesc& dd = d->desc;
    const int rowF = dd.F;
    const char* who = "sg_disc_update_gail_dyn";
    SG_REQUIRE(d->d_expert, "sg_disc_update_gail_dyn: no expert data (call sg_disc_set_expert first)");
    SG_REQUIRE(batch_size > 0, "sg_disc_update_gail_dyn: batch_size must be positive");
    SG_REQUIRE(n_cols >= 0 && (n_cols == 0 || TN_loc % n_cols == 0), "sg_disc_update_gail_dyn: %lld rows are not a grid of %d columns",
               (long long)TN_loc, n_cols);
    // Data-parallel modes (world > 1).  "replicated" (default): the discriminator is replicated; once
    // per call the ranks all-gather their next_obs_feat rows, then every rank runs the SAME sequence of
    // full-batch steps on the global row set (same seeds, deterministic kernels => identical replicas),
    // with no per-step collective.  "sharded" (SG_DISC_DP=sharded): each rank takes batch/world rows of
    // every step and the gradient is all-reduced per step (tests/test_dp_design.py).  Both reproduce the
    // reference at num_processes = world * N; replicated avoids 2,560+ latency-bound 100 KB all-reduces.
    const int world = ctx->world;
    const bool replicated = ctx->use_comm && !ctx->disc_sharded;
    const bool sharded = ctx->use_comm && ctx->disc_sharded;
    // sharded mode with injected draws: rank r takes the batch positions whose policy row it owns (all three row kinds
    // of that position: expert, policy, mixup), so the counts per step are uneven -- the parity form of SURVEY.md 8(e)
    const bool owned = sharded && world > 1 && policy_perm != nullptr;
    SG_REQUIRE(!owned || (expert_perm && alpha), "sg_disc_update_gail_dyn: sharded data-parallel mode takes injected draws all "
               "together (expert_perm, policy_perm and alpha), or none of them");
    SG_REQUIRE(!(sharded && world > 1 && !policy_perm && (expert_perm || alpha)), "sg_disc_update_gail_dyn: sharded data-parallel "
               "mode takes injected draws all together (expert_perm, policy_perm and alpha), or none of them");
    const int split = (sharded && !owned) ? world : 1;
    SG_REQUIRE(batch_size % split == 0, "sg_disc_update_gail_dyn: batch_size %d must divide by world size %d", batch_size, world);
    // the reference's alpha*expert + (1-alpha)*policy raises on a size mismatch when the loader
    // yields a short batch (a2c/algo/gail.py:75)
    SG_REQUIRE(d->n_expert >= batch_size, "The size of tensor a (%lld) must match the size of tensor b (%d) at "
               "non-singleton dimension 0 (expert rows < gail batch size)", (long long)d->n_expert, batch_size);
    SG_CHECK(hipSetDevice(ctx->device));
    const int64_t TN_glob = TN_loc * (ctx->use_comm ? world : 1);
    const int64_t TN = (replicated || owned) ? TN_glob : TN_loc;   // rows the policy permutation ranges over
    const int64_t n_e = d->n_expert / batch_size;   // drop_last (or exactly one full batch)
    const int64_t n_p = (sharded && !owned) ? TN_loc / (batch_size / split) : TN / batch_size;
    const int n_d = (int)(n_e < n_p ? n_e : n_p);
    SG_REQUIRE(n_d > 0, "sg_disc_update_gail_dyn: rollout (%lld rows) smaller than one batch (%d)", (long long)TN, batch_size / split);
    if (n_steps) *n_steps = n_d;
    if (expert_perm) SG_TRY(check_index_array(who, "expert_perm", expert_perm, n_expert_perm, d->n_expert, d->n_expert));
    if (policy_perm) SG_TRY(check_index_array(who, "policy_perm", policy_perm, n_policy_perm, TN, TN));
    if (alpha) SG_REQUIRE(n_alpha >= (int64_t)n_d * batch_size, "sg_disc_update_gail_dyn: alpha holds %lld draws, the epoch consumes %lld",
                          (long long)n_alpha, (long long)n_d * batch_size);
    const float* next_feat = rows_local;
    if (replicated) {
        // The gail_epoch calls of one learner update read the SAME rollout (a2c/main_gail_dyn_ppo.py:253-256): the union gathered
        // for the first of them is reused by the others -- one all-gather (22 MB per rank at the north-star shape) per update
        // instead of five.  "Same" = the same device rows at the same obs_feat version of the rollout (every entry point that
        // writes obs_feat bumps it); every rank runs the same call sequence, so every rank decides alike.  Rows the caller
        // assembled (sg_disc_update_rows) carry no version and are gathered every time.  SG_DISC_GATHER_CACHE=0: always gather.
        const char* cenv = getenv("SG_DISC_GATHER_CACHE");
        const bool reuse = src_version && d->gather_version == src_version && d->gather_src == rows_local && d->gather_rows == TN_loc &&
                           d->feat_all_cap >= TN * rowF && !(cenv && !strcmp(cenv, "0"));
        if (!reuse) {
            d->gather_rows = 0;
            SG_TRY(ensure_cap(&d->d_feat_all, &d->feat_all_cap, TN * rowF, ctx->stream));
            SG_TRY(sg_comm_allgather_f32(ctx, next_feat, d->d_feat_all, TN_loc * rowF));
            d->gather_src = rows_local; d->gather_rows = TN_loc; d->gather_version = src_version;
        }
        d->n_gathers += reuse ? 0 : 1;
        next_feat = d->d_feat_all;
    }

    // per-step geometry: rows [off[k], off[k] + cnt[k]) of the epoch's row copies; B_loc = the largest count (it sizes the
    // launch and the scratch; rows past a step's count are masked inside the kernels)
    std::vector<int> step_cnt(n_d), step_off(n_d + 1);
    std::vector<int64_t> ep_own, pp_own;
    std::vector<float> al_own;
    int B_loc = batch_size / split;
    if (owned) {
        const int64_t Ng = n_cols ? (int64_t)n_cols * world : 0;
        B_loc = 1;
        for (int k = 0; k < n_d; ++k) {
            step_off[k] = (int)pp_own.size();
            for (int i = 0; i < batch_size; ++i) {
                const int64_t g = policy_perm[(size_t)k * batch_size + i];
                int64_t owner, local;
                if (Ng) { const int64_t t = g / Ng, c = g - t * Ng; owner = c / n_cols; local = t * n_cols + (c - owner * n_cols); }
                else { owner = g / TN_loc; local = g - owner * TN_loc; }
                if (owner != ctx->rank) continue;
                pp_own.push_back(local);
                ep_own.push_back(expert_perm[(size_t)k * batch_size + i]);
                al_own.push_back(alpha[(size_t)k * batch_size + i]);
            }
            step_cnt[k] = (int)pp_own.size() - step_off[k];
            if (step_cnt[k] > B_loc) B_loc = step_cnt[k];
        }
        step_off[n_d] = (int)pp_own.size();
    } else {
        for (int k = 0; k <= n_d; ++k) { step_off[k] = k * B_loc; if (k < n_d) step_cnt[k] = B_loc; }
    }
    const int64_t rows_total = step_off[n_d];

    SG_TRY(ensure_cap(&d->d_eperm, &d->eperm_cap, d->n_expert, ctx->stream));
    SG_TRY(ensure_cap(&d->d_pperm, &d->pperm_cap, TN, ctx->stream));
    SG_TRY(ensure_cap(&d->d_alpha, &d->alpha_cap, (int64_t)n_d * batch_size, ctx->stream));
    SG_TRY(ensure_cap(&d->d_erows, &d->erows_cap, (rows_total + B_loc) * rowF, ctx->stream));   // + one step: the last step's
    SG_TRY(ensure_cap(&d->d_prows, &d->prows_cap, (rows_total + B_loc) * rowF, ctx->stream));   // "next" pointers stay in range
    d->rng_calls += 1;
    d->last_draws[0] = d->n_expert; d->last_draws[1] = TN; d->last_draws[2] = (int64_t)n_d * batch_size;
    if (owned) {
        // the device arrays hold this rank's share of the draws, in consumption order (sg_disc_last_draws: not available)
        d->last_draws[0] = d->last_draws[1] = d->last_draws[2] = 0;
        if (rows_total) {
            SG_CHECK(hipMemcpyAsync(d->d_eperm, ep_own.data(), sizeof(int64_t) * rows_total, hipMemcpyHostToDevice, ctx->stream));
            SG_CHECK(hipMemcpyAsync(d->d_pperm, pp_own.data(), sizeof(int64_t) * rows_total, hipMemcpyHostToDevice, ctx->stream));
            SG_CHECK(hipMemcpyAsync(d->d_alpha, al_own.data(), sizeof(float) * rows_total, hipMemcpyHostToDevice, ctx->stream));
            SG_CHECK(hipStreamSynchronize(ctx->stream));   // the host vectors go out of scope with this call
        }
    } else {
        // expert permutation / alpha are GLOBAL (identical on every rank); the library-drawn policy permutation is per rank
        // in sharded mode
        if (expert_perm) SG_CHECK(hipMemcpyAsync(d->d_eperm, expert_perm, sizeof(int64_t) * d->n_expert, hipMemcpyHostToDevice, ctx->stream));
        else SG_TRY(sg_fill_perm(ctx, d->d_eperm, d->n_expert, seed, 0xE0000000ull + d->rng_calls));
        if (policy_perm) SG_CHECK(hipMemcpyAsync(d->d_pperm, policy_perm, sizeof(int64_t) * TN, hipMemcpyHostToDevice, ctx->stream));
        else SG_TRY(sg_fill_perm(ctx, d->d_pperm, TN, seed, 0xF0000000ull + d->rng_calls * 1024 + (uint64_t)(sharded ? ctx->rank : 0)));
        if (alpha) SG_CHECK(hipMemcpyAsync(d->d_alpha, alpha, sizeof(float) * (size_t)n_d * batch_size, hipMemcpyHostToDevice, ctx->stream));
        else {
            const int64_t na = (int64_t)n_d * batch_size;
            hipLaunchKernelGGL(k_fill_alpha, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, ctx->stream, d->d_alpha, na,
                               seed, 0xA1000000ull + d->rng_calls);
        }
        // a queued epoch (out3 == NULL) returns without a host wait: the caller's arrays must have been read by then
        if (!out3 && (expert_perm || policy_perm || alpha)) SG_CHECK(hipStreamSynchronize(ctx->stream));
    }

    const int G = (B_loc + 15) / 16;
    // k_disc_wgrad takes {G, flags, step index} packed into one preloaded scalar (sg_wgrad_pack)
    SG_REQUIRE(G < 1024 && n_d < (1 << SG_WGRAD_PACK_K1_BITS) && dd.Hp < 65536 && dd.Fp < 32768,
               "sg_disc_update_gail_dyn: %d steps per epoch / batch %d exceed the weight-gradient kernel's packed arguments "
               "(< %d steps, batch < 16384 per rank)", n_d, B_loc, 1 << SG_WGRAD_PACK_K1_BITS);
    // scratch of one step: operand stacks | per-workgroup vector partials | (data-parallel) flat gradient
    const bool thin = disc_chain_thin(ctx, dd);
    const bool gw = disc_needs_gw(ctx, dd);
    const int n_chain_wg = thin ? 12 * G : 2 * G;
    // (two sets of partials: k_disc_step4 double-buffers them by step parity like the stacks; the two-launch forms use the first)
    const size_t ops_f = disc_ops_floats(dd, G), part_f = (size_t)12 * G * 4 * dd.Hp, grad_f = (size_t)dd.total + 8;
    if (d->n_slabs < G) {
        SG_CHECK(hipStreamSynchronize(ctx->stream));
        if (d->d_slabs) SG_CHECK(sg_dev_free(d->d_slabs));
        SG_CHECK(sg_dev_malloc((void**)&d->d_slabs, sizeof(float) * (2 * ops_f + part_f + grad_f)));
        d->n_slabs = G;
        // ld-padding entries of the flat gradient are never written by k_disc_wgrad: they must read as zero
        SG_CHECK(hipMemsetAsync(d->d_slabs, 0, sizeof(float) * (2 * ops_f + part_f + grad_f), ctx->stream));
    }

    DiscArgs a;
    a.d = dd; a.params = d->d_params; a.expert = d->d_expert;
    a.next_feat = next_feat;
    a.B = B_loc; a.G = G; a.inv_B = 1.0f / (float)batch_size; a.lambda_ = 10.0f;
    // the operand stacks are double-buffered by step parity: while k_disc_wgrad of step k reads
    // stacks[k&1], its spare workgroups gather step k+1's input rows into stacks[(k+1)&1]
    float* stacks[2] = {d->d_slabs, d->d_slabs + ops_f};
    a.part = d->d_slabs + 2 * ops_f; a.st = reinterpret_cast<SgOptState*>(d->d_state);
    a.dbg = d->d_dbg;
    a.wT = d->d_wT;
    float* grad = d->d_slabs + 2 * ops_f + part_f;
    WgradArgs wa;
    wa.d = dd; wa.part = a.part; wa.G = G; wa.params = d->d_params; wa.m = d->d_m; wa.v = d->d_v;
    wa.grad_out = sharded ? grad : nullptr; wa.st = a.st; wa.eps = 1e-8f; wa.inv_B = a.inv_B; wa.lambda_ = a.lambda_;
    wa.loss_acc = d->d_loss_acc;
    wa.nparts = n_chain_wg; wa.wT = d->d_wT; wa.dbg = d->d_dbg;
    const size_t lds = disc_chain_lds_bytes(dd, gw);
    const int n_tiles = (dd.Hp / 16) * (dd.Hp / 16) + (dd.Hp / 16) * (dd.Fp / 16);
    const int nblk = (dd.total + 255) / 256;
    const int n_vec = (3 * dd.Hp + 4 + 63) / 64;
    const int th_ = dd.Hp / 16, tf_ = dd.Fp / 16;
    const char* xenv = getenv("SG_WGRAD_XCD");
    wa.xcd_map = (th_ <= 7 && n_vec <= (8 - th_) * (th_ + tf_) && !(xenv && !strcmp(xenv, "0"))) ? 1 : 0;
    // blocks that copy the next step's rows: 2G beside the 4-row chain blocks (they have the time), else 2G in k_disc_wgrad,
    // which then only keeps one spare block when no tile slot is free for the lane that evaluates the next Adam scalars
    const int n_gather_wgrad = !thin ? 2 * G : (wa.xcd_map && (8 - th_) * (th_ + tf_) > n_vec) ? 0 : 1;
    const int n_wgrad_blocks = (wa.xcd_map ? 8 * (th_ + tf_) : n_tiles + n_vec) + n_gather_wgrad;
    // one launch per step (k_disc_step4) whenever the 4-row kernel runs a full, unsharded batch; SG_DISC_FUSED=0: two launches
    const char* fenv = getenv("SG_DISC_FUSED");
    // ... and the device is this context's alone (sg_ctx_exclusive: a launch that waits inside itself needs all its workgroups
    // resident); SG_DISC_FUSED=1 forces it (tests), =0 forbids it
    const bool fused = thin && !sharded && !owned && !d->d_dbg && wa.xcd_map && 12 * G <= SG_STEP4_MAX_FLAGS && !d->self_wait_failed &&
                       (fenv ? strcmp(fenv, "0") != 0 : sg_ctx_exclusive(ctx));
    const bool dbg_timing = getenv("SG_DEBUG_TIMING") != nullptr;
    const auto t_enq0 = std::chrono::steady_clock::now();
    // One epoch = zero the loss sums, gather step 0's rows, then (chain, weight gradient) per step, then commit
    // the step count.  No argument of these ~2 n_d launches depends on anything but buffer addresses and the
    // batch geometry (the Adam step number lives on the device), so the sequence is captured into a hipGraph
    // once and replayed: the host then issues one call per epoch instead of a thousand, and a descheduled
    // host thread can no longer starve the GPU in the middle of an epoch.
    auto enqueue_epoch = [&]() -> int {
        hipLaunchKernelGGL(k_zero_f64, dim3(1), dim3(64), 0, ctx->stream, d->d_loss_acc, 3);
        {
            EpochRowsArgs er;
            er.expert = d->d_expert; er.feat = a.next_feat; er.eperm = d->d_eperm; er.pperm = d->d_pperm;
            er.erows = d->d_erows; er.prows = d->d_prows; er.F = dd.F;
            if (owned) {   // the draws were compacted to this rank's share on the host: one run of rows_total rows
                er.n_d = 1; er.B_loc = (int)rows_total; er.batch_size = (int)rows_total; er.roff = 0;
            } else {       // rank r takes rows [r*B_loc, (r+1)*B_loc) of the global expert batch (sharded mode)
                er.n_d = n_d; er.B_loc = B_loc; er.batch_size = batch_size; er.roff = sharded ? ctx->rank * B_loc : 0;
            }
            // replicated mode: the all-gathered union is rank-major, the permutation is in the reference's (t, n_global) order
            er.remap_N = (replicated && world > 1) ? n_cols : 0; er.remap_W = world; er.TN_loc = TN_loc;
            if (rows_total) hipLaunchKernelGGL(k_disc_epoch_rows, dim3(2048), dim3(256), 0, ctx->stream, er);
        }
        for (int k = 0; k < n_d; ++k) {
            // uniform: rank r takes rows [r*B_loc, (r+1)*B_loc) of the global expert batch and of alpha
            const size_t al_off = owned ? (size_t)step_off[k] : (size_t)k * batch_size + (sharded ? (size_t)ctx->rank * B_loc : 0);
            const size_t al_next = owned ? (size_t)step_off[k + 1] : al_off + batch_size;
            a.eperm = d->d_eperm + al_off;
            a.alpha = d->d_alpha + al_off;
            a.pperm = d->d_pperm + (size_t)step_off[k];
            a.B = step_cnt[k];
            a.ops = stacks[k & 1];
            wa.ops = a.ops;
            PregatherArgs pg;
            pg.B = step_cnt[k]; pg.G = G; pg.F = dd.F; pg.Fp = dd.Fp; pg.ldF = dd.ldF; pg.Hp = dd.Hp; pg.st = a.st;
            wa.k1 = k + 1;
            if (k == 0) {   // the first step of the epoch has no predecessor to gather for it
                pg.erows = d->d_erows; pg.prows = d->d_prows; pg.alpha = a.alpha; pg.ops = a.ops;
                hipLaunchKernelGGL(k_disc_pregather, dim3(2 * G), dim3(512), 0, ctx->stream, pg);
            }
            pg.B = k + 1 < n_d ? step_cnt[k + 1] : 0;
            pg.erows = d->d_erows + (size_t)step_off[k + 1] * dd.F; pg.prows = d->d_prows + (size_t)step_off[k + 1] * dd.F;
            pg.alpha = d->d_alpha + al_next;
            pg.ops = (k + 1 < n_d) ? stacks[(k + 1) & 1] : nullptr;
            wa.next = pg;
            if (fused) {
                Step4Args sa;
                sa.next = pg; sa.loss_acc = d->d_loss_acc;
                sa.dbg = (SG_STEP4_VERIFY || k == n_d - 2 || n_d < 2) ? d->d_dbg_step4 : nullptr;   // stamps: one representative step
                launch_disc_step4(ctx, dd, d, a.ops, a.B, G, wa.k1, sa);
                continue;
            }
            if (thin) launch_disc_chain4(ctx, dd, dim3(n_chain_wg + 2 * G), a, pg);
            else launch_disc_chain(ctx, dd, dim3(n_chain_wg), lds, a, gw);
            SG_LAUNCH(ctx, SG_PROF_DISC_WGRAD, k_disc_wgrad, dim3(n_wgrad_blocks), dim3(SG_WGRAD_THREADS), 0, wa.ops, wa.params, wa.m,
                      wa.v, wa.st, wa.wT, dd.Hp | (dd.Fp << 16),
                      sg_wgrad_pack(wa.G, (wa.xcd_map ? 1 : 0) | (wa.grad_out ? 2 : 0) | (wa.dbg ? 4 : 0) | (thin ? 8 : 0), wa.k1), wa);
            if (sharded) {
                SG_TRY(sg_comm_allreduce_f32(ctx, grad, (int64_t)grad_f));
                hipLaunchKernelGGL(k_disc_adam_flat, dim3(nblk), dim3(256), 0, ctx->stream, d->d_params, d->d_m, d->d_v, grad,
                                   dd.total, a.st, 1e-8f, a.inv_B, a.lambda_, d->d_loss_acc, dd, d->d_wT, wa.k1);
            }
        }
        hipLaunchKernelGGL(k_opt_commit, dim3(1), dim3(1), 0, ctx->stream, a.st, n_d);
        return 0;
    };
    // Sharded mode has one RCCL all-reduce per step inside the sequence: captured with it (see sg_ppo_update), unless
    // SG_DISC_GRAPH_COMM=0 or the RCCL build refuses the capture.
    const char* genv = getenv("SG_DISC_GRAPH");
    const char* gcenv = getenv("SG_DISC_GRAPH_COMM");
    const bool comm_ok = !d->graph_refused && (!sharded || (sg_comm_graph_ok(ctx) && !(gcenv && !strcmp(gcenv, "0"))));
    bool use_graph = comm_ok && !owned && !ctx->profile && !d->d_dbg && !(genv && !strcmp(genv, "0"));
    if (use_graph) {
        const uint64_t key[12] = {(uint64_t)(uintptr_t)d->d_slabs, (uint64_t)(uintptr_t)d->d_eperm, (uint64_t)(uintptr_t)d->d_pperm,
                                  (uint64_t)(uintptr_t)d->d_alpha, (uint64_t)(uintptr_t)next_feat, (uint64_t)(uintptr_t)d->d_expert,
                                  (uint64_t)n_d, (uint64_t)B_loc, (uint64_t)batch_size,
                                  (uint64_t)thin | (sharded ? 2u : 0u) | (gw ? 4u : 0u) | (fused ? 8u : 0u) | (sg_comm_peer_on(ctx) ? 16u : 0u) | ((uint64_t)((replicated && world > 1) ? n_cols : 0) << 8), (uint64_t)ops_f | ((uint64_t)sg_comm_peer_generation(ctx) << 40),
                                  (uint64_t)(uintptr_t)d->d_erows ^ ((uint64_t)(uintptr_t)d->d_prows << 1)};
        if (!d->epoch_graph || memcmp(key, d->epoch_graph_key, sizeof key) != 0) {
            if (d->epoch_graph) { SG_CHECK(hipGraphExecDestroy(d->epoch_graph)); d->epoch_graph = nullptr; }
            if (sg_try_capture(ctx, &d->epoch_graph, enqueue_epoch) != 0) {
                d->graph_refused = true;   // reported once on stderr; this object launches kernel by kernel from now on
                use_graph = false;
            } else {
                memcpy(d->epoch_graph_key, key, sizeof key);
            }
        }
        if (use_graph) SG_CHECK(hipGraphLaunch(d->epoch_graph, ctx->stream));
    }
    if (!use_graph) SG_TRY(enqueue_epoch());
    SG_CHECK(hipGetLastError());
    const auto t_enq1 = std::chrono::steady_clock::now();
    d->opt_t += n_d;
    d->last_n_d = n_d;
    // out3 == NULL: the caller does not want this epoch's losses (the reference's main keeps only the last epoch's,
    // a2c/main_gail_dyn_ppo.py:255-256) -- the epoch stays queued, the host does not wait for it, and the next epoch's launch
    // overlaps its execution.  Everything the next call touches is ordered behind it on the library's stream.
    if (!out3) return 0;
    double acc[3];
    SG_TRY(sg_ctx_fetch_f64(ctx, d->d_loss_acc, acc, 3));
    if (dbg_timing) {
        const auto t_done = std::chrono::steady_clock::now();
        fprintf(stderr, "[sg] disc epoch: %d steps, enqueue %.3f ms, enqueue->done %.3f ms\n", n_d,
                std::chrono::duration<double, std::milli>(t_enq1 - t_enq0).count(),
                std::chrono::duration<double, std::milli>(t_done - t_enq1).count());
    }
    if (fused && acc[0] != acc[0]) {   // NaN: either the data, or a weight-gradient workgroup of k_disc_step4 gave up waiting
        unsigned err = 0;
        SG_COPY_SYNC(ctx, &err, reinterpret_cast<unsigned*>(d->d_state) + SG_STEP4_ERR_WORD, sizeof err, hipMemcpyDeviceToHost);
        if (err) {
            SG_CHECK(hipMemsetAsync(reinterpret_cast<unsigned*>(d->d_state) + SG_STEP4_ERR_WORD, 0, sizeof err, ctx->stream));
            d->self_wait_failed = true;
            SG_REQUIRE(false, "sg_disc_update_gail_dyn: the weight-gradient workgroups of k_disc_step4 waited %d s for the chain "
                       "workgroups of their own launch and gave up (the discriminator's state is undefined; SG_DISC_FUSED=0 runs "
                       "the step as two launches)", (int)(SG_STEP4_TIMEOUT_TICKS / 100000000ll));
        }
    }
    for (int i = 0; i < 3; ++i) out3[i] = (float)(acc[i] / n_d);
    return 0;
}

extern "C" int sg_disc_update_gail_dyn(sg_disc* d, sg_rollout* r, int batch_size, const int64_t* expert_perm, int64_t n_expert_perm,
                                       const int64_t* policy_perm, int64_t n_policy_perm, const float* alpha, int64_t n_alpha,
                                       uint64_t seed, float out3[3], int* n_steps) {
    SG_REQUIRE(d && r, "sg_disc_update_gail_dyn: NULL argument");
    SG_REQUIRE(r->F == d->desc.F, "sg_disc_update_gail_dyn: rollout feat_len %d != discriminator input_dim %d", r->F, d->desc.F);
    // policy rows = next_obs_feat = obs_feat[1:]   (a2c/storage.py:172, a2c/algo/gail.py:165)
    // (the trailing argument: the rollout's obs_feat version, see the replicated mode's all-gather in disc_update_core)
    return disc_update_core(d, r->d_field[SG_F_OBS_FEAT] + (size_t)r->N * r->F, (int64_t)r->T * r->N, r->N, batch_size,
                            expert_perm, n_expert_perm, policy_perm, n_policy_perm, alpha, n_alpha, seed, out3, n_steps, r->feat_version);
}

extern "C" int sg_disc_update_rows(sg_disc* d, const float* policy_rows, int64_t n_rows, int n_cols, int batch_size,
                                   const int64_t* expert_perm, int64_t n_expert_perm, const int64_t* policy_perm, int64_t n_policy_perm,
                                   const float* alpha, int64_t n_alpha, uint64_t seed, float out3[3], int* n_steps) {
    SG_REQUIRE(d && policy_rows && n_rows > 0, "sg_disc_update_rows: bad argument");
    sg_ctx* ctx = d->ctx;
    SG_CHECK(hipSetDevice(ctx->device));
    SG_TRY(ensure_cap(&d->d_rows, &d->rows_cap, n_rows * d->desc.F, ctx->stream));
    SG_CHECK(hipMemcpyAsync(d->d_rows, policy_rows, sizeof(float) * (size_t)n_rows * d->desc.F, hipMemcpyHostToDevice, ctx->stream));
    return disc_update_core(d, d->d_rows, n_rows, n_cols, batch_size, expert_perm, n_expert_perm, policy_perm, n_policy_perm, alpha,
                            n_alpha, seed, out3, n_steps);
}

static int disc_forward_dev(sg_disc* d, const float* d_x, int n, float offset, float* d_reward, int prob = 0,
                            const double* neg_offset_dev = nullptr) {
    sg_ctx* ctx = d->ctx;
    DiscFwdArgs f;
    f.d = d->desc; f.params = d->d_params; f.x = d_x; f.n = n; f.offset = offset; f.reward = d_reward; f.prob = prob;
    f.neg_offset_dev = neg_offset_dev;
    int grid = (n + 31) / 32;
    if (grid > 2 * ctx->num_cu) grid = 2 * ctx->num_cu;
    const bool gw = disc_needs_gw(ctx, d->desc) || disc_fwd_lds_bytes(d->desc) > (size_t)ctx->lds_bytes;
    if (gw) SG_LAUNCH(ctx, SG_PROF_RELABEL, k_disc_forward<true>, dim3(grid), dim3(256), disc_fwd_lds_bytes(d->desc, true), f);
    else SG_LAUNCH(ctx, SG_PROF_RELABEL, k_disc_forward<false>, dim3(grid), dim3(256), disc_fwd_lds_bytes(d->desc), f);
    SG_CHECK(hipGetLastError());
    return 0;
}

static int ensure_returns(sg_disc* d, int n) {
    if (d->d_returns && d->returns_n == n) return 0;
    SG_REQUIRE(d->returns_none || d->returns_n == n,
               "Discriminator.returns holds %d rows but %d were passed (the reference would broadcast-fail)", d->returns_n, n);
    SG_CHECK(hipStreamSynchronize(d->ctx->stream));
    if (d->d_returns) SG_CHECK(sg_dev_free(d->d_returns));
    SG_CHECK(sg_dev_malloc((void**)&d->d_returns, sizeof(float) * n));
    SG_CHECK(hipMemsetAsync(d->d_returns, 0, sizeof(float) * n, d->ctx->stream));
    d->returns_n = n;
    return 0;
}

extern "C" int sg_disc_predict_reward(sg_disc* d, const float* x, int n, float gamma, const float* masks,
                                      float offset, float* reward, float* returns) {
    SG_REQUIRE(d && x && masks && reward && returns && n > 0, "sg_disc_predict_reward: bad argument");
    sg_ctx* ctx = d->ctx;
    SG_CHECK(hipSetDevice(ctx->device));
    SG_TRY(ensure_returns(d, n));
    float* scratch = nullptr;
    const size_t fx = (size_t)n * d->desc.F;
    SG_TRY(sg_ctx_scratch(ctx, sizeof(float) * (fx + 2 * (size_t)n), &scratch));
    float* d_x = scratch;
    float* d_masks = d_x + fx;
    float* d_rew = d_masks + n;
    SG_CHECK(hipMemcpyAsync(d_x, x, sizeof(float) * fx, hipMemcpyHostToDevice, ctx->stream));
    SG_CHECK(hipMemcpyAsync(d_masks, masks, sizeof(float) * n, hipMemcpyHostToDevice, ctx->stream));
    SG_TRY(disc_forward_dev(d, d_x, n, offset, d_rew));
    hipLaunchKernelGGL(k_returns_step, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d->d_returns, d_rew, d_masks,
                       gamma, d->returns_none ? 1 : 0, n);
    SG_CHECK(hipGetLastError());
    d->returns_none = false;
    SG_CHECK(hipMemcpyAsync(reward, d_rew, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream));
    SG_CHECK(hipMemcpyAsync(returns, d->d_returns, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream));
    SG_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}

// The unchanged main's relabel loop (a2c/main_gail_dyn_ppo.py:275-280) makes T calls of predict_reward_combined(obs_feat[t + 1],
// gamma, masks[t], offset), each an upload, a launch pair and two read-backs.  This is all T of them in one pass over the rollout's
// device copy: the same forward kernel on the same rows and the same recurrence in the same order (k_returns_scan's body is
// k_returns_step's), so reward[t] / returns[t] are bit for bit what the t-th call returns; Discriminator.returns ends where the
// T-th call leaves it.
extern "C" int sg_disc_predict_reward_steps(sg_disc* d, sg_rollout* r, float gamma, float offset, float* reward, float* returns) {
    SG_REQUIRE(d && r && reward && returns, "sg_disc_predict_reward_steps: NULL argument");
    SG_REQUIRE(r->F == d->desc.F, "sg_disc_predict_reward_steps: rollout feat_len %d != discriminator input_dim %d", r->F, d->desc.F);
    sg_ctx* ctx = d->ctx;
    SG_CHECK(hipSetDevice(ctx->device));
    const int T = r->T, N = r->N;
    const size_t TN = (size_t)T * N;
    SG_TRY(ensure_returns(d, N));
    float* scratch = nullptr;
    SG_TRY(sg_ctx_scratch(ctx, sizeof(float) * 2 * TN, &scratch));
    float *raw = scratch, *rets = scratch + TN;
    SG_TRY(disc_forward_dev(d, r->d_field[SG_F_OBS_FEAT] + (size_t)N * r->F, (int)TN, offset, raw));
    hipLaunchKernelGGL(k_returns_scan, dim3((N + 63) / 64), dim3(64), 0, ctx->stream, d->d_returns, raw, r->d_field[SG_F_MASKS], gamma,
                       d->returns_none ? 1 : 0, T, N, rets);
    SG_CHECK(hipGetLastError());
    d->returns_none = false;
    SG_CHECK(hipMemcpyAsync(reward, raw, sizeof(float) * TN, hipMemcpyDeviceToHost, ctx->stream));
    SG_CHECK(hipMemcpyAsync(returns, rets, sizeof(float) * TN, hipMemcpyDeviceToHost, ctx->stream));
    SG_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int sg_disc_grad_pen(sg_disc* d, const float* expert_rows, const float* policy_rows, const float* alpha, int n,
                                uint64_t seed, float* pen) {
    SG_REQUIRE(d && expert_rows && policy_rows && pen && n > 0, "sg_disc_grad_pen: bad argument");
    sg_ctx* ctx = d->ctx;
    SG_CHECK(hipSetDevice(ctx->device));
    const SgDiscDesc& dd = d->desc;
    float* scratch = nullptr;
    const size_t fx = (size_t)n * dd.F;
    SG_TRY(sg_ctx_scratch(ctx, sizeof(float) * (2 * fx + 2 * (size_t)n), &scratch));
    float *d_e = scratch, *d_p = d_e + fx, *d_al = d_p + fx, *d_pen = d_al + n;
    SG_CHECK(hipMemcpyAsync(d_e, expert_rows, sizeof(float) * fx, hipMemcpyHostToDevice, ctx->stream));
    SG_CHECK(hipMemcpyAsync(d_p, policy_rows, sizeof(float) * fx, hipMemcpyHostToDevice, ctx->stream));
    if (alpha) {
        for (int i = 0; i < n; ++i) SG_REQUIRE(alpha[i] >= 0.f && alpha[i] <= 1.f, "sg_disc_grad_pen: alpha[%d] = %g is not in [0, 1]", i, (double)alpha[i]);
        SG_CHECK(hipMemcpyAsync(d_al, alpha, sizeof(float) * n, hipMemcpyHostToDevice, ctx->stream));
    } else {
        hipLaunchKernelGGL(k_fill_alpha, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_al, (int64_t)n, seed, 0xA2000000ull);
    }
    const size_t lds = sizeof(float) * (size_t)(dd.Fp + 4 * dd.Hp + 2);
    hipLaunchKernelGGL(k_disc_grad_pen, dim3((unsigned)n), dim3(128), lds, ctx->stream, dd, d->d_params, d_e, d_p, d_al, n, d_pen);
    SG_CHECK(hipGetLastError());
    SG_CHECK(hipMemcpyAsync(pen, d_pen, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream));
    SG_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int sg_disc_predict_prob(sg_disc* d, const float* x, int n, float* prob) {
    SG_REQUIRE(d && x && prob && n > 0, "sg_disc_predict_prob: bad argument");
    sg_ctx* ctx = d->ctx;
    SG_CHECK(hipSetDevice(ctx->device));
    float* scratch = nullptr;
    const size_t fx = (size_t)n * d->desc.F;
    SG_TRY(sg_ctx_scratch(ctx, sizeof(float) * (fx + (size_t)n), &scratch));
    SG_CHECK(hipMemcpyAsync(scratch, x, sizeof(float) * fx, hipMemcpyHostToDevice, ctx->stream));
    SG_TRY(disc_forward_dev(d, scratch, n, 0.f, scratch + fx, 1));
    SG_CHECK(hipMemcpyAsync(prob, scratch + fx, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream));
    SG_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int sg_disc_last_draws(sg_disc* d, int64_t* expert_perm, int64_t* policy_perm, float* alpha, int64_t counts3[3]) {
    SG_REQUIRE(d, "sg_disc_last_draws: NULL argument");
    SG_REQUIRE(d->last_draws[0] > 0, "sg_disc_last_draws: no update epoch has run on this discriminator");
    if (counts3) for (int i = 0; i < 3; ++i) counts3[i] = d->last_draws[i];
    SG_CHECK(hipSetDevice(d->ctx->device));
    SG_CHECK(hipStreamSynchronize(d->ctx->stream));
    if (expert_perm) SG_COPY_SYNC(d->ctx, expert_perm, d->d_eperm, sizeof(int64_t) * d->last_draws[0], hipMemcpyDeviceToHost);
    if (policy_perm) SG_COPY_SYNC(d->ctx, policy_perm, d->d_pperm, sizeof(int64_t) * d->last_draws[1], hipMemcpyDeviceToHost);
    if (alpha) SG_COPY_SYNC(d->ctx, alpha, d->d_alpha, sizeof(float) * d->last_draws[2], hipMemcpyDeviceToHost);
    return 0;
}

extern "C" int sg_disc_reset_returns(sg_disc* d) {
    SG_REQUIRE(d, "sg_disc_reset_returns: NULL argument");
    d->returns_none = true;
    return 0;
}

extern "C" int sg_disc_get_returns(sg_disc* d, float* returns, int n, int* is_none) {
    SG_REQUIRE(d && is_none, "sg_disc_get_returns: NULL argument");
    *is_none = d->returns_none ? 1 : 0;
    if (d->returns_none || !returns) return 0;
    SG_REQUIRE(n == d->returns_n, "sg_disc_get_returns: holds %d rows, asked for %d", d->returns_n, n);
    SG_CHECK(hipMemcpyAsync(returns, d->d_returns, sizeof(float) * n, hipMemcpyDeviceToHost, d->ctx->stream));
    SG_CHECK(hipStreamSynchronize(d->ctx->stream));
    return 0;
}

extern "C" int sg_disc_set_returns(sg_disc* d, const float* returns, int n) {
    SG_REQUIRE(d && returns && n > 0, "sg_disc_set_returns: bad argument");
    d->returns_none = true;
    SG_TRY(ensure_returns(d, n));
    SG_CHECK(hipMemcpyAsync(d->d_returns, returns, sizeof(float) * n, hipMemcpyHostToDevice, d->ctx->stream));
    SG_CHECK(hipStreamSynchronize(d->ctx->stream));
    d->returns_none = false;
    return 0;
}

// rms_dev: device {mean, var, count}, updated in place; neg_offset_dev != NULL: the offset is -(*neg_offset_dev)
static int relabel_core(sg_disc* d, sg_rollout* r, float gamma, float offset, const double* neg_offset_dev, double* rms_host) {
    sg_ctx* ctx = d->ctx;
    SG_REQUIRE(r->F == d->desc.F, "sg_disc_relabel_rewards: rollout feat_len %d != discriminator input_dim %d", r->F, d->desc.F);
    SG_CHECK(hipSetDevice(ctx->device));
    const int T = r->T, N = r->N;
    const int64_t TN = (int64_t)T * N;
    SG_TRY(ensure_returns(d, N));
    // scratch: rets [T,N] | stats [2][T] doubles | rms [3] doubles | scale [T]
    float* scratch = nullptr;
    const size_t bytes = sizeof(float) * (size_t)TN + sizeof(double) * (2 * (size_t)T + 4) + sizeof(float) * T + 64;
    SG_TRY(sg_ctx_scratch(ctx, bytes, &scratch));
    float* rets = scratch;
    double* stats = reinterpret_cast<double*>(scratch + ((TN + 3) & ~(int64_t)3));
    double* rms = rms_host ? stats + 2 * (size_t)T : d->d_scal;   // caller's state staged in scratch, or the device-resident one
    float* scale = reinterpret_cast<float*>(stats + 2 * (size_t)T + 4);
    if (rms_host) SG_TRY(sg_ctx_put_f64(ctx, rms, rms_host, 3));
    float* rewards = r->d_field[SG_F_REWARDS];
    // rewards[t] <- D(obs_feat[t+1]) (+offset): rows t*N+n of obs_feat[1:]
    SG_TRY(disc_forward_dev(d, r->d_field[SG_F_OBS_FEAT] + (size_t)N * r->F, (int)TN, offset, rewards, 0, neg_offset_dev));
    hipLaunchKernelGGL(k_returns_scan, dim3((N + 63) / 64), dim3(64), 0, ctx->stream, d->d_returns, rewards,
                       r->d_field[SG_F_MASKS], gamma, d->returns_none ? 1 : 0, T, N, rets);
    d->returns_none = false;
    const double n_global = (double)N * ctx->world;
    hipLaunchKernelGGL(k_batch_stats, dim3(T), dim3(256), 0, ctx->stream, rets, N, n_global, stats, 0);
    if (ctx->use_comm) SG_TRY(sg_comm_allreduce_f64(ctx, stats, T));            // per-step sums over all ranks
    hipLaunchKernelGGL(k_batch_stats, dim3(T), dim3(256), 0, ctx->stream, rets, N, n_global, stats, 1);
    if (ctx->use_comm) SG_TRY(sg_comm_allreduce_f64(ctx, stats + T, T));        // squares about the global mean
    hipLaunchKernelGGL(k_rms_scan, dim3(1), dim3(256), sizeof(double) * 3 * T, ctx->stream, stats, T, n_global, rms, scale);
    hipLaunchKernelGGL(k_normalize_rewards, dim3((unsigned)((TN + 255) / 256)), dim3(256), 0, ctx->stream, rewards, scale, T, N);
    SG_CHECK(hipGetLastError());
    if (rms_host) SG_TRY(sg_ctx_fetch_f64(ctx, rms, rms_host, 3));
    return 0;
}

extern "C" int sg_disc_relabel_rewards(sg_disc* d, sg_rollout* r, float gamma, float offset, double rms_state[3]) {
    SG_REQUIRE(d && r && rms_state, "sg_disc_relabel_rewards: NULL argument");
    return relabel_core(d, r, gamma, offset, nullptr, rms_state);
}

// a2c/main_gail_dyn_ppo.py:258-271 on the device: dones = sum(1 - masks) + N/2; d = 1 - dones / (dones + T*N/len_e);
// r_sa = log d - log(1 - d)   (float64, as numpy computes it on the host)
__global__ void k_alive_bonus(double* scal, double n_glob, double t_steps, double tar_length, int no_alive_bonus) {
    const double dones = scal[3] + n_glob / 2.0;
    const double expert_dones = (t_steps * n_glob) / tar_length;
    const double d_sa = 1.0 - dones / (dones + expert_dones);
    scal[4] = no_alive_bonus ? 0.0 : log(d_sa) - log(1.0 - d_sa);
}

extern "C" int sg_disc_relabel_rewards_auto(sg_disc* d, sg_rollout* r, float gamma, double gail_tar_length, int no_alive_bonus) {
    SG_REQUIRE(d && r && gail_tar_length > 0, "sg_disc_relabel_rewards_auto: bad argument");
    sg_ctx* ctx = d->ctx;
    SG_CHECK(hipSetDevice(ctx->device));
    SG_TRY(sg_rollout_count_dones_dev(r, d->d_scal + 3));
    hipLaunchKernelGGL(k_alive_bonus, dim3(1), dim3(1), 0, ctx->stream, d->d_scal, (double)r->N * ctx->world, (double)r->T,
                       gail_tar_length, no_alive_bonus);
    return relabel_core(d, r, gamma, 0.f, d->d_scal + 4, nullptr);
}

extern "C" int sg_disc_set_rms(sg_disc* d, const double rms_state[3]) {
    SG_REQUIRE(d && rms_state, "sg_disc_set_rms: NULL argument");
    SG_COPY_SYNC(d->ctx, d->d_scal, rms_state, sizeof(double) * 3, hipMemcpyHostToDevice);
    return 0;
}

extern "C" int sg_disc_get_scalars(sg_disc* d, double out5[5]) {
    SG_REQUIRE(d && out5, "sg_disc_get_scalars: NULL argument");
    SG_COPY_SYNC(d->ctx, out5, d->d_scal, sizeof(double) * 5, hipMemcpyDeviceToHost);
    return 0;
}
