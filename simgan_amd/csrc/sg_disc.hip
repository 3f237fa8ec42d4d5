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
//   k_disc_grad   2*G workgroups, G = ceil(batch/16).  Workgroups [0,G) take 16 expert + 16 policy
//                 rows through forward/BCE/backward; workgroups [G,2G) take the 16 matching mixup
//                 rows through forward, input-gradient, penalty and the double backward.  The whole
//                 parameter vector (its HBM image is the LDS image) is staged once per workgroup;
//                 every contraction is an LDS-tile fp32 MFMA GEMM; partial gradients go to a
//                 per-workgroup slab.
//   k_disc_adam   sums the slabs per parameter and applies Adam (no clipping for D).
//
// Gradient-penalty math (x = mixup row, s_i = 1 - h_i^2, lambda = 10, B = batch):
//   d2 = w3*s2; u1 = W2^T d2; d1 = u1*s1; g = W1^T d1; n = |g|; gb = lambda*(2/B)*(n-1)/n * g
//   dW1 += d1 gb^T; bd1 = W1 gb; bu1 = bd1*s1; sb1 = bd1*u1; dW2 += d2 bu1^T; bd2 = W2 bu1
//   dw3 += bd2*s2; sb2 = bd2*w3; z2b = (-2 h2 sb2)*s2; dW2 += z2b h1^T; db2 += z2b
//   h1b = W2^T z2b - 2 h1 sb1; z1b = h1b*s1; dW1 += z1b x^T; db1 += z1b
#include <math.h>
#include <string.h>

#include <vector>

#include "sg_common.h"
#include "sg_rng.hpp"

int sg_fill_perm(sg_ctx* ctx, int64_t* d_perm, int64_t n, uint64_t seed, uint64_t stream_id);

struct DiscArgs {
    SgDiscDesc d;
    const float* params;
    const float* expert;     // [n_expert, F]
    const float* next_feat;  // obs_feat[1:] flattened [T*N, F]
    const int64_t* eperm;    // this step's expert row ids  [B]
    const int64_t* pperm;    // this step's policy row ids  [B]
    const float* alpha;      // [B]
    int B;                   // local rows of this step
    int G;                   // ceil(B/16)
    float inv_B;             // 1 / global batch rows
    float lambda_;
    float* slabs;
    int slab_stride;
    SgOptState* st;
    long long* dbg;          // optional phase timestamps [block][32] (test hook), NULL in production
};

// barrier + (test hook) shader-clock timestamp of the phase that just ended
#define SG_PHASE_SYNC(n)                                                         \
    do {                                                                         \
        __syncthreads();                                                         \
        if (a.dbg && threadIdx.x == 0) a.dbg[blockIdx.x * 32 + (n)] = clock64(); \
    } while (0)

__device__ __forceinline__ float sg_log_sigmoid(float x) { return fminf(x, 0.f) - log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sg_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

static size_t disc_grad_lds_bytes(const SgDiscDesc& d) {
    const size_t bce = (size_t)32 * d.ldF + 3 * 32 * d.ldH + 64;
    const size_t mix = (size_t)2 * 16 * d.ldF + 7 * 16 * d.ldH + 64;
    return sizeof(float) * ((size_t)d.total + (bce > mix ? bce : mix));
}

// sum over the L (power of two, <= 64) consecutive lanes that share a row
__device__ __forceinline__ float sg_rowlane_sum(float s, int L) {
    for (int o = 1; o < L; o <<= 1) s += __shfl_xor(s, o);
    return s;
}

#define SG_DISC_THREADS 512

// KF = pad16(F)/16 and KH = pad16(Hd)/16 as compile-time constants (0 = take them from the
// descriptor at run time): with the shape fixed, every GEMM extent, LDS offset and staging trip
// count folds to a constant, the K/N dispatch switches of the tile engine collapse to the one
// body needed, and the kernel's code shrinks ~6x (it has to stay resident in the 64 KB I-cache).
template <int KF, int KH>
__global__ __launch_bounds__(SG_DISC_THREADS) void k_disc_grad(DiscArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    SgDiscDesc d = a.d;
    if (KF > 0 && KH > 0) {   // same arithmetic as sg_make_disc_desc
        d.Fp = 16 * KF; d.ldF = d.Fp + 4; d.Hp = 16 * KH; d.ldH = d.Hp + 4;
        d.w1 = 0; d.b1 = d.Hp * d.ldF; d.w2 = d.b1 + d.Hp; d.b2 = d.w2 + d.Hp * d.ldH;
        d.w3 = d.b2 + d.Hp; d.b3 = d.w3 + d.Hp; d.total = d.b3 + 16;
    }
    const int tid = threadIdx.x;
    const int ldF = d.ldF, ldH = d.ldH, Fp = d.Fp, Hp = d.Hp, F = d.F;
    float* W = smem;
    const float* W1 = W + d.w1;
    const float* b1 = W + d.b1;
    const float* W2 = W + d.w2;
    const float* b2 = W + d.b2;
    const float* w3 = W + d.w3;
    float* buf = W + d.total;
    float* slab = a.slabs + (size_t)blockIdx.x * a.slab_stride;
    if (a.dbg && tid == 0) a.dbg[blockIdx.x * 32] = clock64();

    const int li = tid & 15, lq = (tid & 63) >> 4;
    if ((int)blockIdx.x < a.G) {
        // ------------------------------------------------ BCE group: rows 0-15 expert, 16-31 policy
        constexpr int R = 32;
        const int g = blockIdx.x;
        float* X = buf;
        float* H1 = X + R * ldF;
        float* H2 = H1 + R * ldH;   // h2, then dZ2 in place
        float* DZ1 = H2 + R * ldH;
        float* DD = DZ1 + R * ldH;
        float* LOSS = DD + R;
        // row gather, 8 elements per lane in flight (index load -> row load are dependent round trips)
        for (int base = tid; base < R * Fp; base += 8 * blockDim.x) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * blockDim.x;
                const int r = i / Fp, c = i - r * Fp;
                const int b = g * 16 + (r & 15);
                v[u] = 0.f;
                if (i < R * Fp && b < a.B && c < F)
                    v[u] = (r < 16) ? a.expert[(size_t)a.eperm[b] * F + c] : a.next_feat[(size_t)a.pperm[b] * F + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * blockDim.x;
                if (i < R * Fp) X[(i / Fp) * ldF + (i % Fp)] = v[u];
            }
        }
        sg_stage(W, a.params, d.total / 4);
        SG_PHASE_SYNC(1);
        // Adam t / bias corrections for the k_disc_adam that follows: one lane of the last wave, which
        // has no tile in the 7-tile GEMM phases, so the double-precision pow() hides behind them
        if (blockIdx.x == 0 && tid == (int)blockDim.x - 64) sg_opt_advance(a.st);
        sg_layer_nt<2>(X, ldF, W1, ldF, Fp, Hp, [&](int r, int c, float v) { H1[r * ldH + c] = sg_tanh(v + b1[c]); });
        SG_PHASE_SYNC(2);
        sg_layer_nt<2>(H1, ldH, W2, ldH, Hp, Hp, [&](int r, int c, float v) { H2[r * ldH + c] = sg_tanh(v + b2[c]); });
        SG_PHASE_SYNC(3);
        {   // logits, BCE losses and dL/dd: blockDim/32 lanes per row   (a2c/algo/gail.py:168-176)
            const int L = blockDim.x / R, r = tid / L, sub = tid % L;
            float s = 0.f;
            for (int c = sub; c < Hp; c += L) s += H2[r * ldH + c] * w3[c];
            s = sg_rowlane_sum(s, L);
            if (sub == 0) {
                const float dd = s + W[d.b3];
                const bool valid = g * 16 + (r & 15) < a.B;
                float loss = 0.f, grad = 0.f;
                if (valid) {
                    if (r < 16) { loss = -sg_log_sigmoid(dd); grad = a.inv_B * (sg_sigmoid(dd) - 1.f); }
                    else { loss = dd - sg_log_sigmoid(dd); grad = a.inv_B * sg_sigmoid(dd); }
                }
                DD[r] = grad;
                LOSS[r] = loss;
            }
        }
        SG_PHASE_SYNC(4);
        // dw3, db2, db3 and dZ2 (in place over H2), one thread per hidden column
        for (int c = tid; c < Hp; c += blockDim.x) {
            const float w = w3[c];
            float gw = 0.f, gb = 0.f;
            for (int r = 0; r < R; ++r) {
                const float h = H2[r * ldH + c], dd = DD[r];
                gw += dd * h;
                const float dz = dd * w * (1.f - h * h);
                gb += dz;
                H2[r * ldH + c] = dz;
            }
            slab[d.w3 + c] = gw;
            slab[d.b2 + c] = gb;
        }
        if (tid >= 256 && tid < 272) {
            float gb = 0.f;
            if (tid == 256) for (int r = 0; r < R; ++r) gb += DD[r];
            slab[d.b3 + tid - 256] = gb;
        }
        if (tid == 320) {
            float le = 0.f, lp = 0.f;
            for (int r = 0; r < 16; ++r) { le += LOSS[r]; lp += LOSS[16 + r]; }
            float* ls = slab + d.total;
            ls[0] = le; ls[1] = lp; ls[2] = 0.f;
        }
        SG_PHASE_SYNC(5);
        // dZ1 = (dZ2 W2) * (1 - h1^2) and db1 in the epilogue; dW2 = dZ2^T h1 alongside
        sg_layer_nn_t<2>(H2, ldH, W2, ldH, Hp, Hp, [&](int tn, f32x4 (&acc)[2][1]) {
            const int c = tn * 16 + li;
            float z[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = i * 16 + 4 * lq + r;
                    const float h = H1[row * ldH + c];
                    z[i][r] = acc[i][0][r] * (1.f - h * h);
                    DZ1[row * ldH + c] = z[i][r];
                }
            const float sb = sg_tile_colsum<2>(z);
            if (lq == 0) slab[d.b1 + c] = sb;
        });
        sg_grad_tn<2>(H2, ldH, H1, ldH, Hp, Hp, slab + d.w2, ldH, false);
        SG_PHASE_SYNC(6);
        sg_grad_tn<2>(DZ1, ldH, X, ldF, Hp, Fp, slab + d.w1, ldF, false);
    } else {
        // ------------------------------------------------ mixup group: gradient penalty on 16 rows
        constexpr int R = 16;
        const int g = blockIdx.x - a.G;
        float* XM = buf;
        float* GX = XM + R * ldF;   // g, then gb
        float* H1 = GX + R * ldF;
        float* H2 = H1 + R * ldH;
        float* D2 = H2 + R * ldH;   // d2
        float* U1 = D2 + R * ldH;   // u1, then sb1, then z1b (all in place, element-wise)
        float* D1 = U1 + R * ldH;   // d1
        float* BU1 = D1 + R * ldH;  // bu1
        float* Z2B = BU1 + R * ldH; // z2b
        float* ROWL = Z2B + R * ldH;
        for (int base = tid; base < R * Fp; base += 4 * blockDim.x) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = base + u * blockDim.x;
                const int r = i / Fp, c = i - r * Fp;
                const int b = g * 16 + r;
                v[u] = 0.f;
                if (i < R * Fp && b < a.B && c < F) {
                    const float al = a.alpha[b];
                    v[u] = al * a.expert[(size_t)a.eperm[b] * F + c] + (1.f - al) * a.next_feat[(size_t)a.pperm[b] * F + c];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = base + u * blockDim.x;
                if (i < R * Fp) XM[(i / Fp) * ldF + (i % Fp)] = v[u];
            }
        }
        sg_stage(W, a.params, d.total / 4);
        SG_PHASE_SYNC(8);
        sg_layer_nt<1>(XM, ldF, W1, ldF, Fp, Hp, [&](int r, int c, float v) { H1[r * ldH + c] = sg_tanh(v + b1[c]); });
        SG_PHASE_SYNC(9);
        sg_layer_nt<1>(H1, ldH, W2, ldH, Hp, Hp, [&](int r, int c, float v) {
            const float h = sg_tanh(v + b2[c]);
            H2[r * ldH + c] = h;
            D2[r * ldH + c] = w3[c] * (1.f - h * h);
        });
        SG_PHASE_SYNC(10);
        sg_layer_nn<1>(D2, ldH, W2, ldH, Hp, Hp, [&](int r, int c, float v) {      // u1 = d2 W2
            const float h = H1[r * ldH + c];
            U1[r * ldH + c] = v;
            D1[r * ldH + c] = v * (1.f - h * h);
        });
        SG_PHASE_SYNC(11);
        sg_layer_nn<1>(D1, ldH, W1, ldF, Hp, Fp, [&](int r, int c, float v) { GX[r * ldF + c] = v; });  // g = d1 W1
        SG_PHASE_SYNC(12);
        {   // per-row |g|, penalty and gb = c_r * g: blockDim/16 lanes per row   (a2c/algo/gail.py:88)
            const int L = blockDim.x / R, r = tid / L, sub = tid % L;
            float s = 0.f;
            for (int c = sub; c < Fp; c += L) { const float v = GX[r * ldF + c]; s += v * v; }
            s = sg_rowlane_sum(s, L);
            const float nn = sqrtf(s);
            const bool valid = g * 16 + r < a.B;
            const float cr = (valid && nn > 0.f) ? a.lambda_ * 2.f * a.inv_B * (nn - 1.f) / nn : 0.f;
            for (int c = sub; c < Fp; c += L) GX[r * ldF + c] *= cr;
            if (sub == 0) ROWL[r] = valid ? (nn - 1.f) * (nn - 1.f) : 0.f;
        }
        SG_PHASE_SYNC(13);
        sg_layer_nt<1>(GX, ldF, W1, ldF, Fp, Hp, [&](int r, int c, float v) {      // bd1 = gb W1^T
            const float h = H1[r * ldH + c];
            BU1[r * ldH + c] = v * (1.f - h * h);
            U1[r * ldH + c] = v * U1[r * ldH + c];                                  // sb1 = bd1*u1
        });
        SG_PHASE_SYNC(14);
        sg_layer_nt_t<1>(BU1, ldH, W2, ldH, Hp, Hp, [&](int tn, f32x4 (&acc)[1][1]) {  // bd2 = bu1 W2^T
            const int c = tn * 16 + li;
            const float w = w3[c];
            float t3[1][4], z[1][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * lq + r;
                const float h = H2[row * ldH + c], s2 = 1.f - h * h, v = acc[0][0][r];
                t3[0][r] = v * s2;                                                  // -> dw3
                z[0][r] = (-2.f * h * (v * w)) * s2;                                // z2b
                Z2B[row * ldH + c] = z[0][r];
            }
            const float sw = sg_tile_colsum<1>(t3), sb = sg_tile_colsum<1>(z);
            if (lq == 0) { slab[d.w3 + c] = sw; slab[d.b2 + c] = sb; }
        });
        SG_PHASE_SYNC(15);
        sg_layer_nn_t<1>(Z2B, ldH, W2, ldH, Hp, Hp, [&](int tn, f32x4 (&acc)[1][1]) {  // h1b = z2b W2
            const int c = tn * 16 + li;
            float z[1][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * lq + r;
                const float h = H1[row * ldH + c];
                z[0][r] = (acc[0][0][r] - 2.f * h * U1[row * ldH + c]) * (1.f - h * h);  // z1b
                U1[row * ldH + c] = z[0][r];
            }
            const float sb = sg_tile_colsum<1>(z);
            if (lq == 0) slab[d.b1 + c] = sb;
        });
        sg_grad_tn2<1>(D2, ldH, BU1, ldH, Z2B, H1, Hp, Hp, slab + d.w2, ldH, false);   // dW2 = d2^T bu1 + z2b^T h1
        SG_PHASE_SYNC(16);
        sg_grad_tn2<1>(D1, ldH, GX, ldF, U1, XM, Hp, Fp, slab + d.w1, ldF, false);     // dW1 = d1^T gb + z1b^T x
        if (tid < 16) slab[d.b3 + tid] = 0.f;
        if (tid == 64) {
            float lg = 0.f;
            for (int r = 0; r < R; ++r) lg += ROWL[r];
            float* ls = slab + d.total;
            ls[0] = 0.f; ls[1] = 0.f; ls[2] = lg;
        }
    }
    SG_PHASE_SYNC(31);
}

// shape-specialised instances: north-star / Laikago (F 86, Hd 100), Hopper (F 25, Hd 100), the
// tiny test shape, and the run-time-shape fallback
static void launch_disc_grad(sg_ctx* ctx, const SgDiscDesc& dd, dim3 grid, size_t lds, const DiscArgs& a) {
    const int kf = dd.Fp / 16, kh = dd.Hp / 16;
    const dim3 block(SG_DISC_THREADS);
    if (kf == 6 && kh == 7) SG_LAUNCH(ctx, SG_PROF_DISC_GRAD, (k_disc_grad<6, 7>), grid, block, lds, a);
    else if (kf == 2 && kh == 7) SG_LAUNCH(ctx, SG_PROF_DISC_GRAD, (k_disc_grad<2, 7>), grid, block, lds, a);
    else if (kf == 1 && kh == 1) SG_LAUNCH(ctx, SG_PROF_DISC_GRAD, (k_disc_grad<1, 1>), grid, block, lds, a);
    else SG_LAUNCH(ctx, SG_PROF_DISC_GRAD, (k_disc_grad<0, 0>), grid, block, lds, a);
}

// out[i] = sum over slabs (data-parallel mode: feeds the all-reduce)
__global__ void k_slab_sum(const float* slabs, int n_slabs, int slab_stride, int count, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float g = 0.f;
    for (int s = 0; s < n_slabs; ++s) g += slabs[(size_t)s * slab_stride + i];
    out[i] = g;
}

// Slab reduction + Adam (torch.optim.Adam defaults: lr 1e-3, betas (0.9, 0.999), eps 1e-8;
// a2c/algo/gail.py:48,186-188) + running loss sums (a2c/algo/gail.py:181-184).
__global__ __launch_bounds__(256) void k_disc_adam(float* params, float* m, float* v, const float* slabs,
                                                   int n_slabs, int slab_stride, int total, const SgOptState* st,
                                                   float eps, float inv_B, float lambda_, double* loss_acc) {
    const float s_step_size = st->step_size, s_bc2_sqrt = st->bc2_sqrt;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        // 16 slab loads in flight per lane (one memory round trip for the usual 2*128/16 slabs);
        // partial sums are combined in a fixed order, so the result is deterministic
        float mi = m[i], vi = v[i], pi = params[i];
        float g = 0.f;
        for (int s0 = 0; s0 < n_slabs; s0 += 16) {
            float p[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) p[u] = (s0 + u < n_slabs) ? slabs[(size_t)(s0 + u) * slab_stride + i] : 0.f;
            g += (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]))) +
                 (((p[8] + p[9]) + (p[10] + p[11])) + ((p[12] + p[13]) + (p[14] + p[15])));
        }
        mi = mi + (g - mi) * (float)(1.0 - 0.9);
        vi = vi * (float)0.999 + (float)(1.0 - 0.999) * g * g;
        const float denom = sqrtf(vi) / s_bc2_sqrt + eps;
        params[i] = pi - s_step_size * (mi / denom);
        m[i] = mi;
        v[i] = vi;
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x < 64) {  // last block has the fewest parameters
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        for (int s = threadIdx.x; s < n_slabs; s += 64) {
            const float* ls = slabs + (size_t)s * slab_stride + total;
            s0 += ls[0]; s1 += ls[1]; s2 += ls[2];
        }
        s0 = sg_wave_sum(s0); s1 = sg_wave_sum(s1); s2 = sg_wave_sum(s2);
        if (threadIdx.x == 0) {
            const float el = s0 * inv_B, pl = s1 * inv_B, gp = lambda_ * (s2 * inv_B);
            loss_acc[0] += (double)(el + pl + gp);
            loss_acc[1] += (double)el;
            loss_acc[2] += (double)pl;
        }
    }
}

__global__ void k_fill_alpha(float* alpha, int64_t n, uint64_t seed, uint64_t stream) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) alpha[i] = sg_uniform(seed, stream, (uint64_t)i);
}

// ------------------------------------------------------------------------ forward / rewards

struct DiscFwdArgs {
    SgDiscDesc d;
    const float* params;
    const float* x;   // [n, F]
    int n;
    float offset;
    float* reward;    // [n]  log(s+1e-7) - log(1-s+1e-7) + offset
};

static size_t disc_fwd_lds_bytes(const SgDiscDesc& d) {
    return sizeof(float) * ((size_t)d.total + 32 * d.ldF + 2 * 32 * d.ldH);
}

__global__ __launch_bounds__(256) void k_disc_forward(DiscFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int R = 32;
    const SgDiscDesc& d = a.d;
    const int tid = threadIdx.x, ldF = d.ldF, ldH = d.ldH, Fp = d.Fp, Hp = d.Hp;
    float* W = smem;
    float* X = W + d.total;
    float* H1 = X + R * ldF;
    float* H2 = H1 + R * ldH;
    const float* b1 = W + d.b1;
    const float* b2 = W + d.b2;
    const float* w3 = W + d.w3;
    sg_stage(W, a.params, d.total / 4);
    for (int base = blockIdx.x * R; base < a.n; base += gridDim.x * R) {
        __syncthreads();
        for (int i = tid; i < R * Fp; i += blockDim.x) {
            const int r = i / Fp, c = i - r * Fp;
            X[r * ldF + c] = (base + r < a.n && c < d.F) ? a.x[(size_t)(base + r) * d.F + c] : 0.f;
        }
        __syncthreads();
        sg_layer_nt<2>(X, ldF, W + d.w1, ldF, Fp, Hp, [&](int r, int c, float v) { H1[r * ldH + c] = sg_tanh(v + b1[c]); });
        __syncthreads();
        sg_layer_nt<2>(H1, ldH, W + d.w2, ldH, Hp, Hp, [&](int r, int c, float v) { H2[r * ldH + c] = sg_tanh(v + b2[c]); });
        __syncthreads();
        const int r = tid >> 3, sub = tid & 7;
        float s = 0.f;
        for (int c = sub; c < Hp; c += 8) s += H2[r * ldH + c] * w3[c];
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
        if (sub == 0 && base + r < a.n) {
            const float sg = sg_sigmoid(s + W[d.b3]);
            a.reward[base + r] = logf(sg + 1e-7f) - logf(1.f - sg + 1e-7f) + a.offset;  // a2c/algo/gail.py:204-205
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
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float ret = first ? 0.f : d_returns[n];
    for (int t = 0; t < T; ++t) {
        const float r = raw[(size_t)t * N + n];
        ret = (first && t == 0) ? r : ret * gamma * masks[(size_t)t * N + n] + r;
        rets[(size_t)t * N + n] = ret;
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
__global__ void k_rms_scan(const double* stats, int T, double n_global, double* rms /*[3] in/out*/, float* scale /*[T]*/) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    double mean = rms[0], var = rms[1], count = rms[2];
    for (int t = 0; t < T; ++t) {
        const double bmean = (double)(float)(stats[t] / n_global);
        const double bvar = (double)(float)(stats[T + t] / n_global);
        const double delta = bmean - mean, tot = count + n_global;
        const double new_mean = mean + delta * n_global / tot;
        const double M2 = var * count + bvar * n_global + delta * delta * count * n_global / tot;
        mean = new_mean; var = M2 / tot; count = tot;
        scale[t] = (float)sqrt(var + 1e-7);
    }
    rms[0] = mean; rms[1] = var; rms[2] = count;
}

__global__ void k_normalize_rewards(float* rewards, const float* scale, int T, int N) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)T * N) return;
    const float v = rewards[i] / scale[i / N];
    rewards[i] = fminf(fmaxf(v, -10.f), 10.f);
}

// --------------------------------------------------------------------------------------- API

extern "C" int sg_disc_create(sg_ctx* ctx, int input_dim, int hidden_dim, sg_disc** out) {
    SG_REQUIRE(ctx && out, "sg_disc_create: NULL argument");
    SG_REQUIRE(input_dim > 0 && hidden_dim > 0, "sg_disc_create: bad dims");
    SG_CHECK(hipSetDevice(ctx->device));
    sg_disc* d = new sg_disc();
    d->ctx = ctx;
    d->desc = sg_make_disc_desc(input_dim, hidden_dim);
    SG_REQUIRE(disc_grad_lds_bytes(d->desc) <= (size_t)ctx->lds_bytes,
               "sg_disc_create: discriminator (%d x %d) needs %zu bytes of LDS, the CU has %d", input_dim, hidden_dim,
               disc_grad_lds_bytes(d->desc), ctx->lds_bytes);
    const size_t tot = d->desc.total;
    SG_CHECK(hipMalloc((void**)&d->d_params, sizeof(float) * tot));
    SG_CHECK(hipMalloc((void**)&d->d_m, sizeof(float) * tot));
    SG_CHECK(hipMalloc((void**)&d->d_v, sizeof(float) * tot));
    SG_CHECK(hipMalloc((void**)&d->d_state, sizeof(SgOptState)));
    SG_CHECK(hipMalloc((void**)&d->d_loss_acc, sizeof(double) * 8));
    SG_CHECK(hipMemsetAsync(d->d_params, 0, sizeof(float) * tot, ctx->stream));
    SG_CHECK(hipMemsetAsync(d->d_m, 0, sizeof(float) * tot, ctx->stream));
    SG_CHECK(hipMemsetAsync(d->d_v, 0, sizeof(float) * tot, ctx->stream));
    SgOptState st;
    memset(&st, 0, sizeof st);
    st.lr = 1e-3f;
    SG_CHECK(hipMemcpyAsync(d->d_state, &st, sizeof st, hipMemcpyHostToDevice, ctx->stream));
    SG_CHECK(hipStreamSynchronize(ctx->stream));
    *out = d;
    return 0;
}

extern "C" int sg_disc_destroy(sg_disc* d) {
    if (!d) return 0;
    (void)hipStreamSynchronize(d->ctx->stream);
    float* ptrs[] = {d->d_params, d->d_m, d->d_v, d->d_slabs, d->d_state, d->d_expert, d->d_alpha, d->d_returns};
    for (float* q : ptrs) if (q) (void)hipFree(q);
    if (d->d_eperm) (void)hipFree(d->d_eperm);
    if (d->d_pperm) (void)hipFree(d->d_pperm);
    if (d->d_loss_acc) (void)hipFree(d->d_loss_acc);
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
    SG_CHECK(hipMemcpy(dev, padded.data(), sizeof(float) * padded.size(), hipMemcpyHostToDevice));
    return 0;
}
static int disc_get(sg_disc* d, const float* dev, float* flat, int64_t n, const char* who) {
    SG_REQUIRE(n == sg_disc_flat_count(d->desc), "%s: expected %lld floats, got %lld", who,
               (long long)sg_disc_flat_count(d->desc), (long long)n);
    std::vector<float> padded(d->desc.total);
    SG_CHECK(hipStreamSynchronize(d->ctx->stream));
    SG_CHECK(hipMemcpy(padded.data(), dev, sizeof(float) * padded.size(), hipMemcpyDeviceToHost));
    sg_disc_unpad(d->desc, padded.data(), flat);
    return 0;
}

extern "C" int sg_disc_set_params(sg_disc* d, const float* flat, int64_t n) {
    SG_REQUIRE(d && flat, "sg_disc_set_params: NULL argument");
    return disc_put(d, d->d_params, flat, n, "sg_disc_set_params");
}
extern "C" int sg_disc_get_params(sg_disc* d, float* flat, int64_t n) {
    SG_REQUIRE(d && flat, "sg_disc_get_params: NULL argument");
    return disc_get(d, d->d_params, flat, n, "sg_disc_get_params");
}
extern "C" int sg_disc_get_adam(sg_disc* d, float* m, float* v, int64_t n, int64_t* step) {
    SG_REQUIRE(d && m && v && step, "sg_disc_get_adam: NULL argument");
    SG_TRY(disc_get(d, d->d_m, m, n, "sg_disc_get_adam"));
    SG_TRY(disc_get(d, d->d_v, v, n, "sg_disc_get_adam"));
    SgOptState st;
    SG_CHECK(hipMemcpy(&st, d->d_state, sizeof st, hipMemcpyDeviceToHost));
    *step = (int64_t)st.step;
    return 0;
}
extern "C" int sg_disc_set_adam(sg_disc* d, const float* m, const float* v, int64_t n, int64_t step) {
    SG_REQUIRE(d && m && v, "sg_disc_set_adam: NULL argument");
    SG_TRY(disc_put(d, d->d_m, m, n, "sg_disc_set_adam"));
    SG_TRY(disc_put(d, d->d_v, v, n, "sg_disc_set_adam"));
    const float fs = (float)step;
    SG_CHECK(hipMemcpy(&reinterpret_cast<SgOptState*>(d->d_state)->step, &fs, sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

extern "C" int sg_disc_set_expert(sg_disc* d, const float* expert, int64_t n_rows) {
    SG_REQUIRE(d && expert && n_rows > 0, "sg_disc_set_expert: bad argument");
    SG_CHECK(hipStreamSynchronize(d->ctx->stream));
    if (d->d_expert) SG_CHECK(hipFree(d->d_expert));
    const size_t bytes = sizeof(float) * (size_t)n_rows * d->desc.F;
    SG_CHECK(hipMalloc((void**)&d->d_expert, bytes));
    SG_CHECK(hipMemcpy(d->d_expert, expert, bytes, hipMemcpyHostToDevice));
    d->n_expert = n_rows;
    return 0;
}

template <typename T>
static int ensure_cap(T** ptr, int64_t* cap, int64_t need, hipStream_t stream) {
    if (*cap >= need) return 0;
    SG_CHECK(hipStreamSynchronize(stream));
    if (*ptr) SG_CHECK(hipFree(*ptr));
    SG_CHECK(hipMalloc((void**)ptr, sizeof(T) * (size_t)need));
    *cap = need;
    return 0;
}

extern "C" int sg_disc_update_gail_dyn(sg_disc* d, sg_rollout* r, int batch_size, const int64_t* expert_perm,
                                       const int64_t* policy_perm, const float* alpha, uint64_t seed,
                                       float out3[3], int* n_steps) {
    SG_REQUIRE(d && r && out3, "sg_disc_update_gail_dyn: NULL argument");
    sg_ctx* ctx = d->ctx;
    const SgDiscDesc& dd = d->desc;
    SG_REQUIRE(d->d_expert, "sg_disc_update_gail_dyn: no expert data (call sg_disc_set_expert first)");
    SG_REQUIRE(r->F == dd.F, "sg_disc_update_gail_dyn: rollout feat_len %d != discriminator input_dim %d", r->F, dd.F);
    SG_REQUIRE(batch_size > 0, "sg_disc_update_gail_dyn: batch_size must be positive");
    const int world = ctx->world;
    SG_REQUIRE(batch_size % world == 0, "sg_disc_update_gail_dyn: batch_size %d must divide by world size %d", batch_size, world);
    const int B_loc = batch_size / world;
    // the reference's alpha*expert + (1-alpha)*policy raises on a size mismatch when the loader
    // yields a short batch (a2c/algo/gail.py:75)
    SG_REQUIRE(d->n_expert >= batch_size, "The size of tensor a (%lld) must match the size of tensor b (%d) at "
               "non-singleton dimension 0 (expert rows < gail batch size)", (long long)d->n_expert, batch_size);
    SG_CHECK(hipSetDevice(ctx->device));
    const int64_t TN = (int64_t)r->T * r->N;
    const int64_t n_e = d->n_expert / batch_size;   // drop_last (or exactly one full batch)
    const int64_t n_p = TN / B_loc;                 // local rows contribute batch/world per step
    const int n_d = (int)(n_e < n_p ? n_e : n_p);
    SG_REQUIRE(n_d > 0, "sg_disc_update_gail_dyn: rollout (%lld rows) smaller than one batch (%d)", (long long)TN, B_loc);
    if (n_steps) *n_steps = n_d;

    SG_TRY(ensure_cap(&d->d_eperm, &d->eperm_cap, d->n_expert, ctx->stream));
    SG_TRY(ensure_cap(&d->d_pperm, &d->pperm_cap, TN, ctx->stream));
    SG_TRY(ensure_cap(&d->d_alpha, &d->alpha_cap, (int64_t)n_d * batch_size, ctx->stream));
    d->rng_calls += 1;
    // expert permutation / alpha are GLOBAL (identical on every rank); the policy permutation is per rank
    if (expert_perm) SG_CHECK(hipMemcpyAsync(d->d_eperm, expert_perm, sizeof(int64_t) * d->n_expert, hipMemcpyHostToDevice, ctx->stream));
    else SG_TRY(sg_fill_perm(ctx, d->d_eperm, d->n_expert, seed, 0xE0000000ull + d->rng_calls));
    if (policy_perm) SG_CHECK(hipMemcpyAsync(d->d_pperm, policy_perm, sizeof(int64_t) * TN, hipMemcpyHostToDevice, ctx->stream));
    else SG_TRY(sg_fill_perm(ctx, d->d_pperm, TN, seed, 0xF0000000ull + d->rng_calls * 1024 + (uint64_t)ctx->rank));
    if (alpha) SG_CHECK(hipMemcpyAsync(d->d_alpha, alpha, sizeof(float) * (size_t)n_d * batch_size, hipMemcpyHostToDevice, ctx->stream));
    else {
        const int64_t na = (int64_t)n_d * batch_size;
        hipLaunchKernelGGL(k_fill_alpha, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, ctx->stream, d->d_alpha, na,
                           seed, 0xA1000000ull + d->rng_calls);
    }

    const int G = (B_loc + 15) / 16;
    const int slab_stride = dd.total + 8;
    if (d->n_slabs < 2 * G + 1) {
        SG_CHECK(hipStreamSynchronize(ctx->stream));
        if (d->d_slabs) SG_CHECK(hipFree(d->d_slabs));
        SG_CHECK(hipMalloc((void**)&d->d_slabs, sizeof(float) * (size_t)(2 * G + 1) * slab_stride));
        d->n_slabs = 2 * G + 1;
        SG_CHECK(hipMemsetAsync(d->d_slabs, 0, sizeof(float) * (size_t)(2 * G + 1) * slab_stride, ctx->stream));
    }
    SG_CHECK(hipMemsetAsync(d->d_loss_acc, 0, sizeof(double) * 3, ctx->stream));

    DiscArgs a;
    a.d = dd; a.params = d->d_params; a.expert = d->d_expert;
    a.next_feat = r->d_field[SG_F_OBS_FEAT] + (size_t)r->N * r->F;
    a.B = B_loc; a.G = G; a.inv_B = 1.0f / (float)batch_size; a.lambda_ = 10.0f;
    a.slabs = d->d_slabs; a.slab_stride = slab_stride; a.st = reinterpret_cast<SgOptState*>(d->d_state);
    a.dbg = d->d_dbg;
    const size_t lds = disc_grad_lds_bytes(dd);
    const int nblk = (dd.total + 255) / 256;
    float* grad = d->d_slabs + (size_t)(2 * G) * slab_stride;   // data-parallel: reduced gradient "slab"
    for (int k = 0; k < n_d; ++k) {
        // rank r takes rows [r*B_loc, (r+1)*B_loc) of the global expert batch and of alpha
        a.eperm = d->d_eperm + (size_t)k * batch_size + (size_t)ctx->rank * B_loc;
        a.alpha = d->d_alpha + (size_t)k * batch_size + (size_t)ctx->rank * B_loc;
        a.pperm = d->d_pperm + (size_t)k * B_loc;
        launch_disc_grad(ctx, dd, dim3(2 * G), lds, a);
        if (ctx->use_comm) {
            hipLaunchKernelGGL(k_slab_sum, dim3((slab_stride + 255) / 256), dim3(256), 0, ctx->stream, d->d_slabs, 2 * G,
                               slab_stride, slab_stride, grad);
            SG_TRY(sg_comm_allreduce_f32(ctx, grad, slab_stride));
        }
        SG_LAUNCH(ctx, SG_PROF_DISC_ADAM, k_disc_adam, dim3(nblk), dim3(256), 0, d->d_params, d->d_m, d->d_v,
                  ctx->use_comm ? grad : d->d_slabs, ctx->use_comm ? 1 : 2 * G, slab_stride, dd.total, a.st, 1e-8f,
                  a.inv_B, a.lambda_, d->d_loss_acc);
    }
    SG_CHECK(hipGetLastError());
    double acc[3];
    SG_CHECK(hipMemcpyAsync(acc, d->d_loss_acc, sizeof acc, hipMemcpyDeviceToHost, ctx->stream));
    SG_CHECK(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 3; ++i) out3[i] = (float)(acc[i] / n_d);
    return 0;
}

static int disc_forward_dev(sg_disc* d, const float* d_x, int n, float offset, float* d_reward) {
    sg_ctx* ctx = d->ctx;
    DiscFwdArgs f;
    f.d = d->desc; f.params = d->d_params; f.x = d_x; f.n = n; f.offset = offset; f.reward = d_reward;
    int grid = (n + 31) / 32;
    if (grid > 2 * ctx->num_cu) grid = 2 * ctx->num_cu;
    SG_LAUNCH(ctx, SG_PROF_RELABEL, k_disc_forward, dim3(grid), dim3(256), disc_fwd_lds_bytes(d->desc), f);
    SG_CHECK(hipGetLastError());
    return 0;
}

static int ensure_returns(sg_disc* d, int n) {
    if (d->d_returns && d->returns_n == n) return 0;
    SG_REQUIRE(d->returns_none || d->returns_n == n,
               "Discriminator.returns holds %d rows but %d were passed (the reference would broadcast-fail)", d->returns_n, n);
    SG_CHECK(hipStreamSynchronize(d->ctx->stream));
    if (d->d_returns) SG_CHECK(hipFree(d->d_returns));
    SG_CHECK(hipMalloc((void**)&d->d_returns, sizeof(float) * n));
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

extern "C" int sg_disc_relabel_rewards(sg_disc* d, sg_rollout* r, float gamma, float offset, double rms_state[3]) {
    SG_REQUIRE(d && r && rms_state, "sg_disc_relabel_rewards: NULL argument");
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
    double* rms = stats + 2 * (size_t)T;
    float* scale = reinterpret_cast<float*>(rms + 4);
    SG_CHECK(hipMemcpyAsync(rms, rms_state, sizeof(double) * 3, hipMemcpyHostToDevice, ctx->stream));
    float* rewards = r->d_field[SG_F_REWARDS];
    // rewards[t] <- D(obs_feat[t+1]) (+offset): rows t*N+n of obs_feat[1:]
    SG_TRY(disc_forward_dev(d, r->d_field[SG_F_OBS_FEAT] + (size_t)N * r->F, (int)TN, offset, rewards));
    hipLaunchKernelGGL(k_returns_scan, dim3((N + 63) / 64), dim3(64), 0, ctx->stream, d->d_returns, rewards,
                       r->d_field[SG_F_MASKS], gamma, d->returns_none ? 1 : 0, T, N, rets);
    d->returns_none = false;
    const double n_global = (double)N * ctx->world;
    hipLaunchKernelGGL(k_batch_stats, dim3(T), dim3(256), 0, ctx->stream, rets, N, n_global, stats, 0);
    if (ctx->use_comm) SG_TRY(sg_comm_allreduce_f64(ctx, stats, T));            // per-step sums over all ranks
    hipLaunchKernelGGL(k_batch_stats, dim3(T), dim3(256), 0, ctx->stream, rets, N, n_global, stats, 1);
    if (ctx->use_comm) SG_TRY(sg_comm_allreduce_f64(ctx, stats + T, T));        // squares about the global mean
    hipLaunchKernelGGL(k_rms_scan, dim3(1), dim3(1), 0, ctx->stream, stats, T, n_global, rms, scale);
    hipLaunchKernelGGL(k_normalize_rewards, dim3((unsigned)((TN + 255) / 256)), dim3(256), 0, ctx->stream, rewards, scale, T, N);
    SG_CHECK(hipGetLastError());
    SG_CHECK(hipMemcpyAsync(rms_state, rms, sizeof(double) * 3, hipMemcpyDeviceToHost, ctx->stream));
    SG_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}

// Test hook: enable/read per-phase shader-clock timestamps of k_disc_grad (tools/phase_times.py).
extern "C" int sg_test_disc_phase_times(sg_disc* d, int enable, long long* out, int n_blocks) {
    SG_REQUIRE(d, "sg_test_disc_phase_times: NULL argument");
    SG_CHECK(hipStreamSynchronize(d->ctx->stream));
    if (enable && !d->d_dbg) {
        SG_CHECK(hipMalloc((void**)&d->d_dbg, sizeof(long long) * 32 * 64));
        SG_CHECK(hipMemset(d->d_dbg, 0, sizeof(long long) * 32 * 64));
    }
    if (out && d->d_dbg) {
        SG_REQUIRE(n_blocks <= 64, "sg_test_disc_phase_times: at most 64 blocks");
        SG_CHECK(hipMemcpy(out, d->d_dbg, sizeof(long long) * 32 * n_blocks, hipMemcpyDeviceToHost));
    }
    if (!enable && d->d_dbg) { SG_CHECK(hipFree(d->d_dbg)); d->d_dbg = nullptr; }
    return 0;
}
