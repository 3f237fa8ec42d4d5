// sg_disc_kernels.hpp -- device kernels of one GAIL discriminator optimizer step.
//
// One step = two launches queued back to back (n_d * gail_epoch of them per update):
//
//   k_disc_chain<KF,KH>   2*G workgroups of 512 threads, G = ceil(batch/16).  Workgroups [0,G)
//       take 16 expert + 16 policy rows through forward / BCE-with-logits / backward-to-
//       activations; workgroups [G,2G) take the 16 matching mixup rows through forward, the input
//       gradient, the penalty and its double backward (a2c/algo/gail.py:67-89,165-188).  This is
//       the serial critical path of the step: a chain of 6 (BCE) or 9 (mixup) dependent LDS-tile
//       MFMA GEMM phases on the 98 KB parameter image staged in LDS.  It does NOT form weight
//       gradients: each phase's epilogue drops the operands of dW = dY^T X into two global
//       "operand stacks" (left = dY-like rows, right = X-like rows), and writes per-workgroup
//       partial bias / w3 / loss sums.
//   k_disc_wgrad          one workgroup per 16x16 tile of W1 and W2 (91 at F=86, Hd=100) + one for
//       the vectors.  dW2 = L2^T R2 and dW1 = L1^T R1 are plain TN GEMMs over the stacked rows
//       (K = 4*16*G = 512 for batch 128); the 4 waves split K, combine through LDS, and the 256
//       threads apply torch-Adam to the tile's 256 parameters in place.  The weight-gradient
//       third of the MFMA work thus runs on ~92 otherwise idle CUs instead of lengthening the
//       16-workgroup chain, and no gradient slab is written or re-read.
//
// Gradient-penalty math (x = mixup row, s_i = 1 - h_i^2, lambda = 10, B = batch):
//   d2 = w3*s2; u1 = W2^T d2; d1 = u1*s1; g = W1^T d1; n = |g|; gb = lambda*(2/B)*(n-1)/n * g
//   dW1 += d1 gb^T; bd1 = W1 gb; bu1 = bd1*s1; sb1 = bd1*u1; dW2 += d2 bu1^T; bd2 = W2 bu1
//   dw3 += bd2*s2; sb2 = bd2*w3; z2b = (-2 h2 sb2)*s2; dW2 += z2b h1^T; db2 += z2b
//   h1b = W2^T z2b - 2 h1 sb1; z1b = h1b*s1; dW1 += z1b x^T; db1 += z1b
//
// Operand stacks (rows Kt = 4*nb, nb = 16*G):            left (dY-like)        right (X-like)
//   rows [0, 2nb)        BCE workgroup g, local row r      L2: dZ2   L1: dZ1     R2: h1    R1: x
//   rows [2nb, 3nb)      mixup workgroup g, first term     L2: d2    L1: d1      R2: bu1   R1: gb
//   rows [3nb, 4nb)      mixup workgroup g, second term    L2: z2b   L1: z1b     R2: h1    R1: x~
#pragma once
#include "sg_common.h"
#include "sg_thin.hpp"

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
    float* ops;              // operand stacks: L2 | R2 | L1 [Kt][ldH] each, then R1 [Kt][ldF]
    float* part;             // [2G][4*Hp]: db1 | db2 | dw3 | {db3, loss_expert, loss_policy, loss_gp, 0...}
    SgOptState* st;
    long long* dbg;          // optional phase timestamps [block][32] (test hook), NULL in production
    const float* wT;         // k_disc_chain4: weight images W1 | W2 | W2^T | W1^T (sg4_img_index), kept in step by k_disc_wgrad
};

// barrier + (test hook) shader-clock timestamp of the phase that just ended
#define SG_PHASE_SYNC(n)                                                         \
    do {                                                                         \
        __syncthreads();                                                         \
        if (a.dbg && threadIdx.x == 0) a.dbg[blockIdx.x * 32 + (n)] = clock64(); \
    } while (0)

__device__ __forceinline__ float sg_log_sigmoid(float x) { return fminf(x, 0.f) - log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sg_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

// sum over the L (power of two, <= 64) consecutive lanes that share a row
__device__ __forceinline__ float sg_rowlane_sum(float s, int L) {
    for (int o = 1; o < L; o <<= 1) s += __shfl_xor(s, o);
    return s;
}

#define SG_DISC_THREADS 512

// Operand stacks of one step.  L2, R2, L1 (Hp columns) and R1t (Fp columns) are stored one 16-column tile
// after another, each tile as [Kt rows][16] (SG_STK): the [K x 16] slab a weight-gradient workgroup
// contracts over is then one contiguous run of whole cache lines.  R1 is ALSO kept row-major
// ([Kt][ldF]) for rows the chain kernels read back as GEMM input (the pre-gathered batch rows).
#define SG_STK(Kt, row, c) ((((size_t)((c) >> 4)) * (size_t)(Kt) + (size_t)(row)) * 16 + ((c) & 15))
struct SgStacks { float *L2, *R2, *L1, *R1, *R1t; };
__host__ __device__ __forceinline__ SgStacks sg_disc_stacks(float* ops, int Kt, int Hp, int Fp, int ldF) {
    SgStacks k;
    k.L2 = ops; k.R2 = k.L2 + (size_t)Kt * Hp; k.L1 = k.R2 + (size_t)Kt * Hp; k.R1 = k.L1 + (size_t)Kt * Hp;
    k.R1t = k.R1 + (size_t)Kt * ldF;
    return k;
}

// gw: the global-weight instance (k_disc_chain<0, 0, true>): LDS holds the activation tiles only
static size_t disc_chain_lds_bytes(const SgDiscDesc& d, bool gw = false) {
    const size_t bce = (size_t)32 * d.ldF + 2 * 32 * d.ldH + 64;
    const size_t mix = (size_t)2 * 16 * d.ldF + 7 * 16 * d.ldH + 64;
    return sizeof(float) * ((gw ? 0 : (size_t)d.total) + (bce > mix ? bce : mix));
}
static size_t disc_ops_floats(const SgDiscDesc& d, int G) {
    const size_t Kt = (size_t)64 * G;
    return Kt * (3 * (size_t)d.Hp + d.ldF + d.Fp);
}

// Input rows of ONE chain workgroup of the NEXT step, gathered by permutation index into the right
// operand stack it will read them from: rows [32g, 32g+32) = 16 expert + 16 policy rows for BCE
// workgroup g (j < G), rows [3nb+16g, +16) = the alpha-mixed rows for mixup workgroup g (j >= G).
// The rows come from the epoch's permuted copies (k_disc_epoch_rows: a2c/storage.py:168-185 gather and the
// DataLoader's shuffled batches, resolved once per epoch), so this is one contiguous read per row, no index
// chase; the mixup (a2c/algo/gail.py:72-75) is formed here.  It runs beside the previous step's weight-gradient
// tiles instead of at the head of the serial chain.  Columns [F, Fp) are written as zeros.
struct PregatherArgs {
    const float* erows;   // this step's expert rows [B][F] in batch order
    const float* prows;   // this step's policy rows [B][F]
    const float* alpha;   // [B]
    float* ops;     // operand stacks of the step being prepared
    int B, G, F, Fp, ldF, Hp;
    SgOptState* st; // the first step's Adam scalars are prepared alongside (one lane of k_disc_pregather)
};

// Once per epoch: expert and policy rows in the order the epoch's steps consume them.
struct EpochRowsArgs {
    const float *expert, *feat;
    const int64_t *eperm, *pperm;
    float *erows, *prows;
    int n_d, B_loc, batch_size, roff, F;
    // replicated data-parallel mode: `feat` is the all-gathered union, rank-major ([world][TN_loc] rows), while the
    // permutation is in the reference's numbering at num_processes = world * N: row t * (N*world) + rank * N + n
    // (a2c/storage.py:168-185).  remap_N = N (0: the permutation indexes `feat` directly), remap_W = world.
    int remap_N, remap_W;
    int64_t TN_loc;
};
__global__ __launch_bounds__(256) void k_disc_epoch_rows(EpochRowsArgs a) {
    const int64_t total = (int64_t)a.n_d * a.B_loc * a.F;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / a.F;
        const int c = (int)(i - row * a.F);
        const int k = (int)(row / a.B_loc), b = (int)(row - (int64_t)k * a.B_loc);
        a.erows[i] = a.expert[(size_t)a.eperm[(size_t)k * a.batch_size + a.roff + b] * a.F + c];
        int64_t g = a.pperm[(size_t)k * a.B_loc + b];
        if (a.remap_N) {
            const int64_t Ng = (int64_t)a.remap_N * a.remap_W, t = g / Ng, cg = g - t * Ng, rk = cg / a.remap_N;
            g = rk * a.TN_loc + t * a.remap_N + (cg - rk * a.remap_N);
        }
        a.prows[i] = a.feat[(size_t)g * a.F + c];
    }
}

__device__ __forceinline__ void sg_disc_pregather(const PregatherArgs& p, int j) {
    const int nb = 16 * p.G, Kt = 4 * nb;
    const SgStacks stk = sg_disc_stacks(p.ops, Kt, p.Hp, p.Fp, p.ldF);
    float* R1s = stk.R1;
    float* R1t = stk.R1t;
    const int Fp = p.Fp, F = p.F;
    const bool bce = j < p.G;
    const int g = bce ? j : j - p.G;
    const int rows = bce ? 32 : 16;
    const int row0 = bce ? 32 * g : 3 * nb + 16 * g;
    for (int base = threadIdx.x; base < rows * Fp; base += 8 * blockDim.x) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * blockDim.x;
            const int r = i / Fp, c = i - r * Fp;
            const int b = g * 16 + (r & 15);
            v[u] = 0.f;
            if (i < rows * Fp && b < p.B && c < F) {
                if (bce) v[u] = (r < 16) ? p.erows[(size_t)b * F + c] : p.prows[(size_t)b * F + c];
                else {
                    const float al = p.alpha[b];
                    v[u] = al * p.erows[(size_t)b * F + c] + (1.f - al) * p.prows[(size_t)b * F + c];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * blockDim.x;
            if (i < rows * Fp) {
                R1s[(size_t)(row0 + i / Fp) * p.ldF + (i % Fp)] = v[u];
                R1t[SG_STK(Kt, row0 + i / Fp, i % Fp)] = v[u];
            }
        }
    }
}

__global__ __launch_bounds__(512) void k_disc_pregather(PregatherArgs p) {
    if (blockIdx.x == 0 && threadIdx.x == 0) sg_opt_prepare(p.st, p.st->t0 + 1);
    sg_disc_pregather(p, blockIdx.x);
}

// KF = pad16(F)/16 and KH = pad16(Hd)/16 as compile-time constants (0 = take them from the
// descriptor at run time): with the shape fixed, every GEMM extent, LDS offset and staging trip
// count folds to a constant and the K/N dispatch switches of the tile engine collapse to the one
// body needed (the kernel has to stay resident in the 64 KB instruction cache).
// GW = true (run-time extents only): the parameter vector is NOT staged -- the layer GEMMs read W1 / W2 from global memory
// (sg_gemm.hpp, "GW"), so any (input_dim, hidden_dim) the reference's constructor accepts runs (a2c/algo/gail.py:40-43,
// --gail-dis-hdim a2c/arguments.py:212-215), as long as the 16- / 32-row activation tiles fit a CU's LDS.
template <int KF, int KH, bool GW = false>
__global__ __launch_bounds__(SG_DISC_THREADS) void k_disc_chain(DiscArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    SgDiscDesc d = a.d;
    if (KF > 0 && KH > 0) {   // same arithmetic as sg_make_disc_desc
        d.Fp = 16 * KF; d.ldF = d.Fp + 4; d.Hp = 16 * KH; d.ldH = d.Hp + 4;
        d.w1 = 0; d.b1 = d.Hp * d.ldF; d.w2 = d.b1 + d.Hp; d.b2 = d.w2 + d.Hp * d.ldH;
        d.w3 = d.b2 + d.Hp; d.b3 = d.w3 + d.Hp; d.total = d.b3 + 16;
    }
    const int tid = threadIdx.x;
    const int ldF = d.ldF, ldH = d.ldH, Fp = d.Fp, Hp = d.Hp;
    const float* W = GW ? a.params : smem;
    const float* W1 = W + d.w1;
    const float* b1 = W + d.b1;
    const float* W2 = W + d.w2;
    const float* b2 = W + d.b2;
    const float* w3 = W + d.w3;
    float* buf = smem + (GW ? 0 : d.total);
    const int nb = 16 * a.G, Kt = 4 * nb;
    const SgStacks stk = sg_disc_stacks(a.ops, Kt, Hp, Fp, ldF);
    float* L2s = stk.L2;
    float* R2s = stk.R2;
    float* L1s = stk.L1;
    float* R1s = stk.R1;     // row-major: pre-gathered input rows
    float* R1t = stk.R1t;    // tiled: gb rows for the weight gradient
    float* part = a.part + (size_t)blockIdx.x * (4 * Hp);   // db1 | db2 | dw3 | scalars
    if (a.dbg && tid == 0) a.dbg[blockIdx.x * 32] = clock64();

    const int li = tid & 15, lq = (tid & 63) >> 4;
    if ((int)blockIdx.x < a.G) {
        // ------------------------------------------------ BCE group: rows 0-15 expert, 16-31 policy
        constexpr int R = 32;
        const int g = blockIdx.x;
        const int row0 = 32 * g;                 // this workgroup's rows in the operand stacks
        float* X = buf;
        float* H1 = X + R * ldF;
        float* H2 = H1 + R * ldH;   // h2, then dZ2 in place
        float* DD = H2 + R * ldH;
        float* LOSS = DD + R;
        // parameter image: loads issued now (12 x 16 B per lane), committed to LDS after the row gather
        float4 wv[12];
        if (!GW) sg_stage_issue<12>(wv, a.params, d.total / 4);
        // this workgroup's 32 input rows were gathered into the right stack by the previous launch
        // (sg_disc_pregather): contiguous 16 B loads, no index indirection on the critical path
        for (int i = tid; i < R * (Fp / 4); i += blockDim.x) {
            const int r = i / (Fp / 4), c = 4 * (i % (Fp / 4));
            *reinterpret_cast<float4*>(X + r * ldF + c) = *reinterpret_cast<const float4*>(R1s + (size_t)(row0 + r) * ldF + c);
        }
        if (!GW) sg_stage_commit<12>(smem, wv, a.params, d.total / 4);
        SG_PHASE_SYNC(1);
        sg_layer_nt<2, GW>(X, ldF, W1, ldF, Fp, Hp, [&](int r, int c, float v) {
            const float h = sg_tanh(v + b1[c]);
            H1[r * ldH + c] = h;
            R2s[SG_STK(Kt, row0 + r, c)] = h;
        });
        SG_PHASE_SYNC(2);
        sg_layer_nt<2, GW>(H1, ldH, W2, ldH, Hp, Hp, [&](int r, int c, float v) { H2[r * ldH + c] = sg_tanh(v + b2[c]); });
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
        // dw3, db2, db3 and dZ2 (in place over H2, and into the left stack), one thread per hidden column
        for (int c = tid; c < Hp; c += blockDim.x) {
            const float w = w3[c];
            float gw = 0.f, gb = 0.f;
            for (int r = 0; r < R; ++r) {
                const float h = H2[r * ldH + c], dd = DD[r];
                gw += dd * h;
                const float dz = dd * w * (1.f - h * h);
                gb += dz;
                H2[r * ldH + c] = dz;
                L2s[SG_STK(Kt, row0 + r, c)] = dz;
            }
            part[2 * Hp + c] = gw;
            part[Hp + c] = gb;
        }
        if (tid >= 256 && tid < 272) {
            float s = 0.f;
            if (tid == 256) for (int r = 0; r < R; ++r) s += DD[r];               // db3
            if (tid == 257) for (int r = 0; r < 16; ++r) s += LOSS[r];            // sum expert BCE
            if (tid == 258) for (int r = 16; r < 32; ++r) s += LOSS[r];           // sum policy BCE
            part[3 * Hp + tid - 256] = s;
        }
        SG_PHASE_SYNC(5);
        // dZ1 = (dZ2 W2) * (1 - h1^2) straight to the left stack; db1 from the epilogue registers
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
                    L1s[SG_STK(Kt, row0 + row, c)] = z[i][r];
                }
            const float sb = sg_tile_colsum<2>(z);
            if (lq == 0) part[c] = sb;
        });
    } else {
        // ------------------------------------------------ mixup group: gradient penalty on 16 rows
        constexpr int R = 16;
        const int g = blockIdx.x - a.G;
        const int rowA = 2 * nb + 16 * g, rowB = 3 * nb + 16 * g;
        float* XM = buf;
        float* GX = XM + R * ldF;   // g, then gb
        float* H1 = GX + R * ldF;
        float* H2 = H1 + R * ldH;
        float* D2 = H2 + R * ldH;   // d2
        float* U1 = D2 + R * ldH;   // u1, then sb1
        float* D1 = U1 + R * ldH;   // d1
        float* BU1 = D1 + R * ldH;  // bu1
        float* Z2B = BU1 + R * ldH; // z2b
        float* ROWL = Z2B + R * ldH;
        float4 wv[12];
        if (!GW) sg_stage_issue<12>(wv, a.params, d.total / 4);
        for (int i = tid; i < R * (Fp / 4); i += blockDim.x) {   // pre-gathered mixup rows (see above)
            const int r = i / (Fp / 4), c = 4 * (i % (Fp / 4));
            *reinterpret_cast<float4*>(XM + r * ldF + c) = *reinterpret_cast<const float4*>(R1s + (size_t)(rowB + r) * ldF + c);
        }
        if (!GW) sg_stage_commit<12>(smem, wv, a.params, d.total / 4);
        SG_PHASE_SYNC(8);
        sg_layer_nt<1, GW>(XM, ldF, W1, ldF, Fp, Hp, [&](int r, int c, float v) {
            const float h = sg_tanh(v + b1[c]);
            H1[r * ldH + c] = h;
            R2s[SG_STK(Kt, rowB + r, c)] = h;
        });
        SG_PHASE_SYNC(9);
        sg_layer_nt<1, GW>(H1, ldH, W2, ldH, Hp, Hp, [&](int r, int c, float v) {
            const float h = sg_tanh(v + b2[c]);
            const float d2 = w3[c] * (1.f - h * h);
            H2[r * ldH + c] = h;
            D2[r * ldH + c] = d2;
            L2s[SG_STK(Kt, rowA + r, c)] = d2;
        });
        SG_PHASE_SYNC(10);
        sg_layer_nn<1>(D2, ldH, W2, ldH, Hp, Hp, [&](int r, int c, float v) {      // u1 = d2 W2
            const float h = H1[r * ldH + c];
            const float d1 = v * (1.f - h * h);
            U1[r * ldH + c] = v;
            D1[r * ldH + c] = d1;
            L1s[SG_STK(Kt, rowA + r, c)] = d1;
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
            for (int c = sub; c < Fp; c += L) {
                const float gb = GX[r * ldF + c] * cr;
                GX[r * ldF + c] = gb;
                R1t[SG_STK(Kt, rowA + r, c)] = gb;
            }
            if (sub == 0) ROWL[r] = valid ? (nn - 1.f) * (nn - 1.f) : 0.f;
        }
        SG_PHASE_SYNC(13);
        sg_layer_nt<1, GW>(GX, ldF, W1, ldF, Fp, Hp, [&](int r, int c, float v) {      // bd1 = gb W1^T
            const float h = H1[r * ldH + c];
            const float bu1 = v * (1.f - h * h);
            BU1[r * ldH + c] = bu1;
            R2s[SG_STK(Kt, rowA + r, c)] = bu1;
            U1[r * ldH + c] = v * U1[r * ldH + c];                                  // sb1 = bd1*u1
        });
        SG_PHASE_SYNC(14);
        sg_layer_nt_t<1, GW>(BU1, ldH, W2, ldH, Hp, Hp, [&](int tn, f32x4 (&acc)[1][1]) {  // bd2 = bu1 W2^T
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
                L2s[SG_STK(Kt, rowB + row, c)] = z[0][r];
            }
            const float sw = sg_tile_colsum<1>(t3), sb = sg_tile_colsum<1>(z);
            if (lq == 0) { part[2 * Hp + c] = sw; part[Hp + c] = sb; }
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
                L1s[SG_STK(Kt, rowB + row, c)] = z[0][r];
            }
            const float sb = sg_tile_colsum<1>(z);
            if (lq == 0) part[c] = sb;
        });
        if (tid >= 448 && tid < 464) {   // the idle last wave: scalars of this workgroup
            float s = 0.f;
            if (tid == 451) for (int r = 0; r < R; ++r) s += ROWL[r];             // sum (|g|-1)^2
            part[3 * Hp + tid - 448] = s;
        }
    }
    SG_PHASE_SYNC(31);
}

// ---------------------------------------------------------------------------------------------
// k_disc_chain4<KF,KH>: the same serial part of the step on 4-row blocks (sg_thin.hpp).
//   grid 12*G workgroups of max(KF,KH) waves (one per 16 output columns).  Workgroups [0,4G): 4 mixup rows (batch rows 4q..4q+3) through
//   the 7 dependent GEMMs of the gradient penalty; workgroups [4G,12G): the matching 4 expert rows (even)
//   or 4 policy rows (odd) through forward / BCE / backward-to-activations.  Wave w owns hidden (or
//   input) columns [16w, 16w+16) in every phase and holds its slice of W1, W2, W1^T and W2^T in
//   registers (27 x 16 B per lane at F=86, Hd=100), fetched once from L2 while the first phases
//   already run; LDS carries only the 4-row activations between phases.  Compared with
//   k_disc_chain the step uses 4x the CUs and a phase costs 28 MFMA issues per wave instead of 56
//   16x16x4 issues (4x the cycles each).  Operand stacks, partials and math are identical.
//   Requires compile-time KF, KH <= 8 (one wave per 16 columns, at most 8 waves; launched with exactly max(KF,KH) waves).
// Arguments are individual scalars / pointers (14 dwords), not a struct: the code object asks the command
// processor to preload them into SGPRs (-amdgpu-kernarg-preload-count), so the first global loads do not wait
// for a kernarg fetch from memory (~1 us after a fresh launch).
// A wave cannot run ahead of the vector-memory instructions it has issued, and a CU's memory pipe moves ~64 B per clock:
// a block whose waves each queue all four of their weight slices (190 KB) up front spends ~4k cycles issuing before its
// first GEMM.  The slices are therefore requested in consumption order with the LAST one (W1^T, first used by the
// fourth phase) deferred until the first GEMM has been issued, and a bare s_barrier (no counter wait: the loads stay in
// flight) after the W1 requests keeps the pipe -- which serves the oldest wave first -- from queueing wave 0's later
// matrices ahead of wave 6's first one.
#define SG4_ISSUE_FENCE()                      \
    do {                                       \
        __builtin_amdgcn_s_barrier();          \
        __builtin_amdgcn_sched_barrier(0);     \
    } while (0)

struct Chain4Args {
    const float *params, *wT;
    float *ops, *part;
    long long* dbg;
    int B, G;
    float inv_B, lambda_;
    unsigned* vlog = nullptr;   // SG_STEP4_VERIFY builds: [block][8 waves][8][64 lanes] xor checksums of what the wave computed with
};
template <int NW_>
__device__ __forceinline__ unsigned sg_xor4(const float4 (&w)[NW_]) {
    unsigned x = 0;
#pragma unroll
    for (int t = 0; t < NW_; ++t) x ^= __float_as_uint(w[t].x) ^ (__float_as_uint(w[t].y) * 3u) ^ (__float_as_uint(w[t].z) * 5u) ^ (__float_as_uint(w[t].w) * 7u);
    return x;
}
// A store another workgroup reads.  WT = false: a plain store (the reader is the NEXT launch; the end-of-kernel release
// writes the line back).  WT = true: a write-through (agent-scope, `sc1`) store for a reader inside the SAME launch
// (k_disc_step4: the weight-gradient workgroups poll a flag and read with L1-bypassing loads).
template <bool WT>
__device__ __forceinline__ void sg_pub(float* p, float v) {
    if (WT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// The serial part of one discriminator step on 4-row blocks (see k_disc_chain4 below); sm = 5*4*ldAH + 4*ldAF + 64 floats of LDS.
template <int KF, int KH, bool WT>
__device__ __forceinline__ void sg_chain4_body(const Chain4Args& a, float* sm) {
    constexpr int Fp = 16 * KF, Hp = 16 * KH, ldF = Fp + 4, ldH = Hp + 4;
    constexpr int ldAF = Fp + 8, ldAH = Hp + 8;   // LDS activation strides: rows 0..3 land in disjoint bank octets
    constexpr int o_b1 = Hp * ldF, o_b2 = o_b1 + Hp + Hp * ldH, o_w3 = o_b2 + Hp, o_b3 = o_w3 + Hp;
    const float* P = a.params;
    const float* I_W1 = a.wT;                    // images (sg_thin.hpp): W1 for x W1^T, W2 for h W2^T,
    const float* I_W2 = I_W1 + Hp * Fp;          // W2^T for d W2, W1^T for d W1
    const float* I_W2T = I_W2 + Hp * Hp;
    const float* I_W1T = I_W2T + Hp * Hp;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, s = lane >> 4, cl = lane & 15;
    // Column-tile ownership is rotated by the block index: the 12 blocks of an XCD would otherwise ask their L2 for the
    // same weight slices in the same order at the same moment (one hot channel at a time); rotated, wave 0 of
    // neighbouring blocks starts on different slices.
    constexpr int NWC_ = KF > KH ? KF : KH;
    const int tile = (wave + (int)(blockIdx.x >> 3)) % NWC_;
    const int col = 16 * tile + cl;
    const bool actH = tile < KH, actF = tile < KF;
    const int wvH = actH ? tile : KH - 1, colH = 16 * wvH + cl;
    const int nb = 16 * a.G, Kt = 4 * nb, G4 = 4 * a.G;
    const SgStacks stk = sg_disc_stacks(a.ops, Kt, Hp, Fp, ldF);
    float* L2s = stk.L2;
    float* R2s = stk.R2;
    float* L1s = stk.L1;
    float* R1s = stk.R1;     // row-major: pre-gathered input rows
    float* R1t = stk.R1t;    // tiled: gb rows for the weight gradient
    float* part = a.part + (size_t)blockIdx.x * (4 * Hp);   // db1 | db2 | dw3 | scalars
    if (a.dbg && tid == 0) { a.dbg[blockIdx.x * 32] = clock64(); a.dbg[blockIdx.x * 32 + 28] = wall_clock64(); }
    float4 w1[SG4_NW(Fp)], w2[SG4_NW(Hp)], w2t[SG4_NW(Hp)];
    float b1c = 0.f, b2c = 0.f, w3c = 0.f;

    if ((int)blockIdx.x >= G4) {
        // ------------------------------------------------ BCE group: 4 expert rows (even) or 4 policy rows (odd)
        const int bq = blockIdx.x - G4;
        const int bi0 = 4 * (bq >> 1);
        const bool is_policy = bq & 1;
        const int r0 = 32 * (bi0 >> 4) + (bi0 & 15) + (is_policy ? 16 : 0);   // this group's rows in the stacks
        float* H1 = sm;                 // [4][ldAH]
        float* DZ2 = H1 + 4 * ldAH;     // [4][ldAH]
        float* LP = DZ2 + 4 * ldAH;     // [8 waves][4 rows] logit partials
        float4 ax[1][SG4_NCH(Fp)];
        // loads are unconditional and in consumption order (a wave past KH, if the block has one, fetches a duplicate slice): memory
        // returns in order, so the first GEMM waits for its own operands only and the later matrices land
        // while it runs.  The scheduling barriers keep the compiler from sinking the small loads to the end.
        sg4_load_a<Fp, 1>(ax, R1s + (size_t)r0 * ldF, ldF, lane);   // pre-gathered by the previous launch
        sg4_load_w<Fp>(w1, I_W1, wvH, lane);
        __builtin_amdgcn_sched_barrier(0);
        b1c = P[o_b1 + colH]; b2c = P[o_b2 + colH]; w3c = P[o_w3 + colH];   // first needed by the first epilogue
        const float b3 = P[o_b3];
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_sched_barrier(0);
        sg4_load_w<Hp>(w2, I_W2, wvH, lane);
        __builtin_amdgcn_sched_barrier(0);
        sg4_load_w<Hp>(w2t, I_W2T, wvH, lane);
        __builtin_amdgcn_sched_barrier(0);
        float h1 = 0.f, h2 = 0.f;
        if (actH) {
            float o[1];
            sg4_mma<Fp, 1>(ax, w1, lane, o);
            h1 = sg_tanh(o[0] + b1c);
            H1[s * ldAH + col] = h1;
            sg_pub<WT>(&R2s[SG_STK(Kt, r0 + s, col)], h1);
        }
        SG_PHASE_SYNC(1);
        if (actH) {
            float4 ah[1][SG4_NCH(Hp)];
            float o[1];
            sg4_load_a<Hp, 1>(ah, H1, ldAH, lane);
            sg4_mma<Hp, 1>(ah, w2, lane, o);
            h2 = sg_tanh(o[0] + b2c);
            const float pl = sg4_rowsum16(h2 * w3c);
            if (cl == 0) LP[tile * 4 + s] = pl;
        }
        SG_PHASE_SYNC(2);
        {   // logit, BCE loss and dL/dd of this lane's row   (a2c/algo/gail.py:168-176)
            const bool valid = bi0 + s < a.B;
            float dd = b3;
#pragma unroll
            for (int w = 0; w < KH; ++w) dd += LP[w * 4 + s];
            // sigmoid and log-sigmoid from one exponential: e = exp(-|dd|)
            const float e = expf(-fabsf(dd)), inv = 1.f / (1.f + e), l1p = log1pf(e);
            const float sig = dd >= 0.f ? inv : e * inv, lsig = fminf(dd, 0.f) - l1p;
            float loss = 0.f, grad = 0.f;
            if (valid) {
                if (!is_policy) { loss = -lsig; grad = a.inv_B * (sig - 1.f); }
                else { loss = dd - lsig; grad = a.inv_B * sig; }
            }
            if (actH) {
                const float dz = grad * w3c * (1.f - h2 * h2);
                DZ2[s * ldAH + col] = dz;
                sg_pub<WT>(&L2s[SG_STK(Kt, r0 + s, col)], dz);
                const float gw = sg4_colsum(grad * h2), gb = sg4_colsum(dz);
                if (s == 0) { sg_pub<WT>(&part[2 * Hp + col], gw); sg_pub<WT>(&part[Hp + col], gb); }
            }
            if (wave == 0) {
                const float db3 = sg4_colsum(grad), ls = sg4_colsum(loss);
                if (lane < 4) sg_pub<WT>(&part[3 * Hp + lane], lane == 0 ? db3 : (lane == 1 && !is_policy) || (lane == 2 && is_policy) ? ls : 0.f);
            }
        }
        SG_PHASE_SYNC(3);
        if (actH) {   // dZ1 = (dZ2 W2) * (1 - h1^2) straight to the left stack; db1 = its column sums
            float4 ad[1][SG4_NCH(Hp)];
            float o[1];
            sg4_load_a<Hp, 1>(ad, DZ2, ldAH, lane);
            sg4_mma<Hp, 1>(ad, w2t, lane, o);
            const float dz1 = o[0] * (1.f - h1 * h1);
            sg_pub<WT>(&L1s[SG_STK(Kt, r0 + s, col)], dz1);
            const float sb = sg4_colsum(dz1);
            if (s == 0) sg_pub<WT>(&part[col], sb);
            if (a.vlog) {
                unsigned* o = a.vlog + ((size_t)blockIdx.x * 8 + wave) * 8 * 64 + lane;
                o[0] = sg_xor4(w1); o[64] = sg_xor4(w2); o[128] = sg_xor4(w2t); o[192] = __float_as_uint(h1); o[256] = __float_as_uint(dz1);
                o[320] = __float_as_uint(ad[0][0].x) ^ __float_as_uint(ad[0][SG4_NCH(Hp) - 1].y); o[384] = __float_as_uint(h2);
            }
        }
    } else {
        // ------------------------------------------------ mixup group: gradient penalty on 4 rows
        const int bi0 = 4 * blockIdx.x;   // the longest chain: these blocks are dispatched first
        const int rowA = 2 * nb + bi0, rowB = 3 * nb + bi0;
        float* H1 = sm;               // [4][ldAH] each
        float* D2 = H1 + 4 * ldAH;
        float* D1 = D2 + 4 * ldAH;
        float* BU1 = D1 + 4 * ldAH;
        float* Z2B = BU1 + 4 * ldAH;
        float* GX = sm + 5 * 4 * ldAH;   // [4][ldAF]
        float4 w1t[SG4_NW(Hp)];
        float4 ax[1][SG4_NCH(Fp)];
        sg4_load_a<Fp, 1>(ax, R1s + (size_t)rowB * ldF, ldF, lane);      // pre-gathered mixup rows
        sg4_load_w<Fp>(w1, I_W1, wvH, lane);
        __builtin_amdgcn_sched_barrier(0);
        b1c = P[o_b1 + colH]; b2c = P[o_b2 + colH]; w3c = P[o_w3 + colH];
        __builtin_amdgcn_sched_barrier(0);
        SG4_ISSUE_FENCE();
        sg4_load_w<Hp>(w2, I_W2, wvH, lane);
        __builtin_amdgcn_sched_barrier(0);
        float h1 = 0.f, h2 = 0.f, u1 = 0.f, sb1 = 0.f, gown = 0.f;
        if (actH) {
            float o[1];
            sg4_mma<Fp, 1>(ax, w1, lane, o);
            __builtin_amdgcn_sched_barrier(0);
            sg4_load_w<Hp>(w2t, I_W2T, wvH, lane);                         // deferred: first used two phases from here
            __builtin_amdgcn_sched_barrier(0);
            h1 = sg_tanh(o[0] + b1c);
            H1[s * ldAH + col] = h1;
            sg_pub<WT>(&R2s[SG_STK(Kt, rowB + s, col)], h1);
        }
        SG_PHASE_SYNC(8);
        if (actH) {
            float4 av[1][SG4_NCH(Hp)];
            float o[1];
            sg4_load_a<Hp, 1>(av, H1, ldAH, lane);
            sg4_mma<Hp, 1>(av, w2, lane, o);
            __builtin_amdgcn_sched_barrier(0);
            sg4_load_w<Hp>(w1t, I_W1T, tile < KF ? tile : KF - 1, lane);   // first used two phases from here
            __builtin_amdgcn_sched_barrier(0);
            h2 = sg_tanh(o[0] + b2c);
            const float d2 = w3c * (1.f - h2 * h2);
            D2[s * ldAH + col] = d2;
            sg_pub<WT>(&L2s[SG_STK(Kt, rowA + s, col)], d2);
        }
        SG_PHASE_SYNC(9);
        if (actH) {   // u1 = d2 W2
            float4 av[1][SG4_NCH(Hp)];
            float o[1];
            sg4_load_a<Hp, 1>(av, D2, ldAH, lane);
            sg4_mma<Hp, 1>(av, w2t, lane, o);
            u1 = o[0];
            const float d1 = u1 * (1.f - h1 * h1);
            D1[s * ldAH + col] = d1;
            sg_pub<WT>(&L1s[SG_STK(Kt, rowA + s, col)], d1);
        }
        SG_PHASE_SYNC(10);
        if (actF) {   // g = d1 W1
            float4 av[1][SG4_NCH(Hp)];
            float o[1];
            sg4_load_a<Hp, 1>(av, D1, ldAH, lane);
            sg4_mma<Hp, 1>(av, w1t, lane, o);
            gown = o[0];
            GX[s * ldAF + col] = gown;
        }
        SG_PHASE_SYNC(11);
        {   // per-row |g|, penalty coefficient c_r (a2c/algo/gail.py:88) and bd1 = (c_r g) W1^T.  The row scale
            // commutes with the GEMM, so g W1^T is issued straight after the barrier and scaled by this lane's own
            // c_r in the epilogue; the norms (every wave recomputes the four of them from LDS rather than spending a
            // barrier) overlap the MFMAs instead of preceding them.
            float4 av[1][SG4_NCH(Fp)];
            float o[1] = {0.f};
            if (actH) {
                sg4_load_a<Fp, 1>(av, GX, ldAF, lane);
                sg4_mma<Fp, 1>(av, w1, lane, o);
            }
            float ss = 0.f;
#pragma unroll
            for (int q = 0; q < KF; ++q) { const float v = GX[s * ldAF + 16 * q + cl]; ss += v * v; }
            ss = sg4_rowsum16(ss);
            const float nn = __builtin_amdgcn_sqrtf(ss);                      // v_sqrt_f32 / v_rcp_f32 (1 ulp): the correctly rounded
            const bool valid = bi0 + s < a.B;                                 // sequences are ~60 VALU instructions on this phase's path
            const float cr = (valid && nn > 0.f) ? a.lambda_ * 2.f * a.inv_B * (nn - 1.f) * __builtin_amdgcn_rcpf(nn) : 0.f;
            if (actF) sg_pub<WT>(&R1t[SG_STK(Kt, rowA + s, col)], gown * cr);                // gb
            if (wave == 0) {
                const float rl = sg4_colsum(valid ? (nn - 1.f) * (nn - 1.f) : 0.f);
                if (lane < 4) sg_pub<WT>(&part[3 * Hp + lane], lane == 3 ? rl : 0.f);         // sum (|g|-1)^2
            }
            if (actH) {
                const float bd1 = o[0] * cr;
                const float bu1 = bd1 * (1.f - h1 * h1);
                BU1[s * ldAH + col] = bu1;
                sg_pub<WT>(&R2s[SG_STK(Kt, rowA + s, col)], bu1);
                sb1 = bd1 * u1;
            }
        }
        SG_PHASE_SYNC(13);
        if (actH) {   // bd2 = bu1 W2^T
            float4 av[1][SG4_NCH(Hp)];
            float o[1];
            sg4_load_a<Hp, 1>(av, BU1, ldAH, lane);
            sg4_mma<Hp, 1>(av, w2, lane, o);
            const float s2 = 1.f - h2 * h2, bd2 = o[0];
            const float t3 = bd2 * s2;                                   // -> dw3
            const float z2b = (-2.f * h2 * (bd2 * w3c)) * s2;
            Z2B[s * ldAH + col] = z2b;
            sg_pub<WT>(&L2s[SG_STK(Kt, rowB + s, col)], z2b);
            const float sw = sg4_colsum(t3), sb = sg4_colsum(z2b);
            if (s == 0) { sg_pub<WT>(&part[2 * Hp + col], sw); sg_pub<WT>(&part[Hp + col], sb); }
        }
        SG_PHASE_SYNC(14);
        if (actH) {   // h1b = z2b W2
            float4 av[1][SG4_NCH(Hp)];
            float o[1];
            sg4_load_a<Hp, 1>(av, Z2B, ldAH, lane);
            sg4_mma<Hp, 1>(av, w2t, lane, o);
            const float z1b = (o[0] - 2.f * h1 * sb1) * (1.f - h1 * h1);
            sg_pub<WT>(&L1s[SG_STK(Kt, rowB + s, col)], z1b);
            const float sb = sg4_colsum(z1b);
            if (s == 0) sg_pub<WT>(&part[col], sb);
            if (a.vlog) {
                unsigned* o = a.vlog + ((size_t)blockIdx.x * 8 + wave) * 8 * 64 + lane;
                o[0] = sg_xor4(w1); o[64] = sg_xor4(w2); o[128] = sg_xor4(w2t); o[192] = __float_as_uint(h1); o[256] = __float_as_uint(z1b);
                o[320] = sg_xor4(w1t); o[384] = __float_as_uint(h2); o[448] = __float_as_uint(u1);
            }
        }
    }
    SG_PHASE_SYNC(31);
    if (a.dbg && tid == 0) a.dbg[blockIdx.x * 32 + 29] = wall_clock64();
}
#define SG_CHAIN4_LDS_FLOATS(KF, KH) (5 * 4 * (16 * (KH) + 8) + 4 * (16 * (KF) + 8) + 64)

template <int KF, int KH>
__global__ __launch_bounds__(512) void k_disc_chain4(const float* params, const float* wT, float* ops, float* part_base,
                                                    long long* dbg, int B, int G, float inv_B, float lambda_, PregatherArgs next) {
    // Workgroups past the 12G chain blocks copy the NEXT step's rows into the other parity's operand stacks (nothing of this
    // step touches those).  Such a block is one round trip to the epoch's row copies in HBM + the write-back of its stores,
    // ~2.5 us: riding in k_disc_wgrad (rounds 1-3) it was as long as that kernel's tile blocks (round 4: the weight-gradient
    // kernel with ONLY these blocks left in it cost the step 2.46 us, the whole kernel 2.9); beside the 4.3 us chain blocks
    // it is free.
    if ((int)blockIdx.x >= 12 * G) {
        if (next.ops) sg_disc_pregather(next, (int)blockIdx.x - 12 * G);
        return;
    }
    __shared__ __attribute__((aligned(16))) float sm[SG_CHAIN4_LDS_FLOATS(KF, KH)];
    const Chain4Args a{params, wT, ops, part_base, dbg, B, G, inv_B, lambda_};
    sg_chain4_body<KF, KH, false>(a, sm);
}

// The four weight images of k_disc_chain4 from the padded parameter vector (after sg_disc_set_params).
// positions of weight element (row, col) in the two images that hold it
__device__ __forceinline__ void sg_disc_img_pos(const SgDiscDesc& d, bool is_w2, int row, int col, int& i0, int& i1) {
    const int Hp = d.Hp, Fp = d.Fp;
    if (is_w2) {
        i0 = Hp * Fp + sg4_img_index(row, col, Hp);                       // W2   (n = row, k = col)
        i1 = Hp * Fp + Hp * Hp + sg4_img_index(col, row, Hp);             // W2^T (n = col, k = row)
    } else {
        i0 = sg4_img_index(row, col, Fp);                                 // W1
        i1 = Hp * Fp + 2 * Hp * Hp + sg4_img_index(col, row, Hp);         // W1^T
    }
}
__device__ __forceinline__ void sg_disc_img_store(const SgDiscDesc& d, float* img, bool is_w2, int row, int col, float p) {
    int i0, i1;
    sg_disc_img_pos(d, is_w2, row, col, i0, i1);
    img[i0] = p;
    img[i1] = p;
}
__global__ __launch_bounds__(256) void k_disc_images(SgDiscDesc d, const float* params, float* img) {
    const int n1 = d.Hp * d.Fp, n2 = d.Hp * d.Hp;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n1 + n2; i += gridDim.x * blockDim.x) {
        const bool w2 = i >= n1;
        const int j = w2 ? i - n1 : i, nc = w2 ? d.Hp : d.Fp, r = j / nc, c = j % nc;
        sg_disc_img_store(d, img, w2, r, c, params[(w2 ? d.w2 : d.w1) + r * (w2 ? d.ldH : d.ldF) + c]);
    }
}

// ---------------------------------------------------------------------------------------------
struct WgradArgs {
    SgDiscDesc d;
    const float* ops;
    const float* part;
    int G;
    float *params, *m, *v;
    float* grad_out;          // data-parallel mode: write the gradient here instead of applying Adam
    const SgOptState* st;
    float eps, inv_B, lambda_;
    double* loss_acc;
    PregatherArgs next;       // next step's inputs (next.ops == NULL on the last step)
    int nparts;               // workgroups of the chain kernel (rows of `part`)
    int k1;                   // 1-based step index within the epoch: Adam step t = st->t0 + k1, scalars in slot t & 1
    float* wT;                // k_disc_chain4's weight images to keep in step (NULL: not maintained)
    long long* dbg;           // optional wall-clock stamps per block (test hook), NULL in production
    int xcd_map;              // tiles grouped by row panel per XCD (grid = align8(2G) + 8 * (th + tf) blocks)
};

// torch.optim.Adam single-tensor math (a2c/algo/gail.py:48,186-188: lr 1e-3, eps 1e-8)
__device__ __forceinline__ void sg_adam_apply(float* p, float* m, float* v, float g, float step_size,
                                              float bc2_sqrt, float eps) {
    float mi = *m, vi = *v;
    mi = mi + (g - mi) * (float)(1.0 - 0.9);
    vi = vi * (float)0.999 + (float)(1.0 - 0.999) * g * g;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    *p = *p - step_size * (mi / denom);
    *m = mi;
    *v = vi;
}

#define SG_WGRAD_THREADS 512

// Everything a weight-tile block and a vector block read before their last stores rides in the preloaded scalars (the
// code object asks the command processor to preload the leading arguments into SGPRs: fourteen dwords for this signature):
// six pointers, the two padded extents in one dword, and {G, flags, the step's index} in another; the per-workgroup partials
// sit at a fixed place behind the two operand stacks, so their address follows from c_ops and the step's parity.  The struct
// behind them serves the blocks that copy the next step's rows (16-row chain kernel only), the data-parallel gradient output
// and the loss scalars of one lane.  (Round 4 measured what this buys: nothing -- a kernarg field or the hidden blockDim.x
// read at the head of every block costs <= 0.04 us per launch, tools/probes/launch_floor.hip; wall-clock stamps that seemed to
// show 0.8 us between a tile block's first instruction and its first operand load were showing the stamps' own s_memrealtime
// round trips.  What DID bound this kernel was found by removing block bodies one at a time: the blocks copying the next
// step's rows were as long as the tile blocks, 2.5 against 2.9 us over an empty launch -- they now ride beside the chain.)
#define SG_WGRAD_PACK_K1_BITS 18
__host__ __device__ __forceinline__ int sg_wgrad_pack(int G, int flags, int k1) { return G | (flags << 10) | (k1 << 14); }
constexpr float SG_DISC_ADAM_EPS = 1e-8f;   // a2c/algo/gail.py:48 (torch.optim.Adam default)
__global__ __launch_bounds__(SG_WGRAD_THREADS) void k_disc_wgrad(const float* c_ops, float* c_params, float* c_m, float* c_v,
                                                                const SgOptState* c_st, float* c_wT, int c_HpFp /* Hp | Fp << 16 */,
                                                                int c_pack /* G | flags << 10 | k1 << 14; flags 1: xcd_map, 2: grad_out set, 4: dbg set, 8: 4-row chain */,
                                                                WgradArgs a) {
    __shared__ float red[8][256];
    SgDiscDesc d;   // same arithmetic as sg_make_disc_desc, from the preloaded extents
    d.F = 0; d.Hd = 0;
    d.Hp = c_HpFp & 0xffff; d.Fp = (int)((unsigned)c_HpFp >> 16); d.ldF = d.Fp + 4; d.ldH = d.Hp + 4;
    d.w1 = 0; d.b1 = d.Hp * d.ldF; d.w2 = d.b1 + d.Hp; d.b2 = d.w2 + d.Hp * d.ldH;
    d.w3 = d.b2 + d.Hp; d.b3 = d.w3 + d.Hp; d.total = d.b3 + 16;
    const int c_G = c_pack & 1023, c_flags = (c_pack >> 10) & 15, c_k1 = (int)((unsigned)c_pack >> 14);
    const bool xcd_map = c_flags & 1, has_grad_out = c_flags & 2, has_dbg = c_flags & 4;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lq = lane >> 4;
    constexpr int nw = SG_WGRAD_THREADS / 64;
    const int th = d.Hp >> 4, tf = d.Fp >> 4;
    const int T2 = th * th, T1 = th * tf;
    const int Kt = 64 * c_G;
    // test hook: wall clock (100 MHz) at block start / after the operand loads / after the LDS reduce / end
    long long* stamp = (has_dbg && threadIdx.x == 0) ? a.dbg + 32 * 256 + 4 * blockIdx.x : nullptr;
    if (stamp) stamp[0] = wall_clock64();
    // Role of this workgroup: 0 weight tile b, 1 vector block b, 2 everything short (copy the next step's rows and / or
    // evaluate the next step's Adam scalars, or nothing); the double-precision pow() of the Adam scalars is instantiated once.
    int role, b = 0, j = 0;
    bool prepare = false, gather = false;
    if (xcd_map) {
        // Workgroups go to the 8 XCDs round-robin by index and every XCD has its own L2.  All tiles of one row
        // panel (same 16 rows of W2 / W1, hence the same left slab) are given to one XCD, so an XCD pulls one
        // left slab plus the right slabs from the memory side instead of nearly all of both stacks; the spare XCD
        // column(s) take the vector blocks and, in the first slot left over, the lane that evaluates the next
        // step's Adam scalars.  Behind the 16-row chain kernel, the 2G blocks that copy the next step's rows come last.
        const int ntv = 8 * (th + tf), NV = (3 * d.Hp + 4 + 63) / 64;
        const bool has_spare = (8 - th) * (th + tf) > NV;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        if ((int)blockIdx.x >= ntv) { role = 2; j = blockIdx.x - ntv; prepare = !has_spare && j == 0; gather = !(c_flags & 8); }
        else if (xcd < th) { role = 0; b = slot < th ? xcd * th + slot : T2 + xcd * tf + (slot - th); }
        else {
            const int vid = (xcd - th) * (th + tf) + slot;
            if (vid < NV) { role = 1; b = T2 + T1 + vid; }
            else { role = 2; prepare = vid == NV; }
        }
    } else {
        // linear order: the 2G blocks that prepare the next step first (one lane also evaluates its Adam scalars)
        const int ng = (c_flags & 8) ? 1 : 2 * c_G;   // the 4-row chain kernel copies the next step's rows itself
        if ((int)blockIdx.x < ng) { role = 2; j = blockIdx.x; prepare = j == 0; gather = !(c_flags & 8); }
        else { b = blockIdx.x - ng; role = b < T2 + T1 ? 0 : 1; }
    }
    if (__builtin_expect(role == 2, 0)) {
        if (prepare && tid == 0) {
            sg_opt_prepare(const_cast<SgOptState*>(c_st), c_st->t0 + c_k1 + 1);
            if (stamp) stamp[3] = wall_clock64();
        }
        if (gather && a.next.ops) sg_disc_pregather(a.next, j);
        return;
    }
    if (role == 0) {
        const bool w2 = b < T2;
        const int t = w2 ? b : b - T2;
        const int tm = w2 ? t / th : t / tf, tn = w2 ? t % th : t % tf;
        const int ldp = w2 ? d.ldH : d.ldF;
        // the tile's 256 parameters and moments: requested first so they arrive with the operands
        const int idx = (w2 ? d.w2 : d.w1) + (tm * 16 + ((tid & 255) >> 4)) * ldp + tn * 16 + (tid & 15);
        float p0 = 0.f, m0 = 0.f, v0 = 0.f;
        float4 sc = float4{0.f, 0.f, 1.f, 1.f};   // both slots of the Adam scalars; the step's parity picks one at the end
        int t0 = 0;
        if (tid < 256 && !has_grad_out) {
            p0 = c_params[idx]; m0 = c_m[idx]; v0 = c_v[idx];
            sc = *reinterpret_cast<const float4*>(c_st->step_size2);
            t0 = c_st->t0;
        }
        const SgStacks stk = sg_disc_stacks(const_cast<float*>(c_ops), Kt, d.Hp, d.Fp, d.ldF);
        // the two [Kt x 16] slabs this tile contracts: contiguous in the tiled stacks
        const float* L = (w2 ? stk.L2 : stk.L1) + (size_t)tm * Kt * 16;
        const float* Rr = (w2 ? stk.R2 : stk.R1t) + (size_t)tn * Kt * 16;
        // waves split the stacked rows in 16-row chunks dealt round-robin; within a chunk MFMA step s takes
        // rows 4s..4s+3 (k = lq), so one load instruction of a wave covers 256 contiguous bytes
        const int n_chunks = Kt >> 4;
        const __amdgpu_buffer_rsrc_t rL = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(L), 0, Kt * 64, 0x00020000);
        const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Rr), 0, Kt * 64, 0x00020000);
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, alt = acc;
        for (int c0 = wave; c0 < n_chunks; c0 += 4 * nw) {       // up to 4 chunks = 32 loads in flight per lane
            float x[4][4], y[4][4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int c = c0 + cc * nw;
                    const int r = 16 * c + 4 * s + lq;   // chunks past the slab read as zero (buffer range check)
                    x[cc][s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rL, (r * 16 + li) * 4, 0, 0));
                    y[cc][s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rR, (r * 16 + li) * 4, 0, 0));
                }
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if (s & 1) alt = sg_mfma(x[cc][s], y[cc][s], alt);
                    else acc = sg_mfma(x[cc][s], y[cc][s], acc);
                }
        }
        acc += alt;
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][(4 * lq + r) * 16 + li] = acc[r];
        if (stamp) stamp[1] = wall_clock64();
        __syncthreads();
        if (stamp) stamp[2] = wall_clock64();
        if (tid < 256) {
            // every address is formed before the first store: an index computed after a store would reuse the store's
            // data registers and make the compiler wait for the store to be acknowledged (a memory round trip)
            int img0 = 0, img1 = 0;
            sg_disc_img_pos(d, w2, tm * 16 + (tid >> 4), tn * 16 + (tid & 15), img0, img1);
            float g = 0.f;
            {   // the eight partials are read with independent LDS loads, then added in a fixed order
                float r8[SG_WGRAD_THREADS / 64];
#pragma unroll
                for (int w = 0; w < SG_WGRAD_THREADS / 64; ++w) r8[w] = red[w][tid];
#pragma unroll
                for (int w = 0; w < SG_WGRAD_THREADS / 64; ++w) g += r8[w];
            }
            if (has_grad_out) a.grad_out[idx] = g;
            else {
                const bool odd = (t0 + c_k1) & 1;
                const float step_size = odd ? sc.y : sc.x, bc2_sqrt = odd ? sc.w : sc.z;
                m0 = m0 + (g - m0) * (float)(1.0 - 0.9);
                v0 = v0 * (float)0.999 + (float)(1.0 - 0.999) * g * g;
                const float denom = sqrtf(v0) / bc2_sqrt + SG_DISC_ADAM_EPS;
                p0 = p0 - step_size * (m0 / denom);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_nontemporal_store(p0, c_params + idx);
                __builtin_nontemporal_store(m0, c_m + idx);
                __builtin_nontemporal_store(v0, c_v + idx);
                if (c_wT) { __builtin_nontemporal_store(p0, c_wT + img0); __builtin_nontemporal_store(p0, c_wT + img1); }
            }
        }
        if (stamp) stamp[3] = wall_clock64();
    } else {
        // vectors: db1 | db2 | dw3 | db3 and the three loss sums from the per-workgroup partials.  Vector block vb
        // owns elements [64 vb, 64 vb + 64); its 8 waves each sum an eighth of the partials (independent
        // loads, 4 in flight per lane), then combine through LDS.
        const int Hp = d.Hp, nparts = ((c_flags & 8) ? 12 : 2) * c_G, stride = 4 * Hp, NE = 3 * Hp + 4;
        // scratch = stacks[0] | stacks[1] | partials (disc_update_core); this step's stacks are stacks[(k1 - 1) & 1]
        const float* c_part = c_ops + (size_t)(2 - ((c_k1 - 1) & 1)) * ((size_t)Kt * (3 * d.Hp + d.ldF + d.Fp));
        const int i = 64 * (b - (T2 + T1)) + lane;
        const float4 sc = *reinterpret_cast<const float4*>(c_st->step_size2);
        const bool odd = (c_st->t0 + c_k1) & 1;
        const float step_size = odd ? sc.y : sc.x, bc2_sqrt = odd ? sc.w : sc.z;
        // the element's parameter and moments are requested with the partials: one memory round trip per block
        const bool is_param = wave == 0 && i < 3 * Hp + 1 && !has_grad_out;
        const int pidx = i < Hp ? d.b1 + i : i < 2 * Hp ? d.b2 + (i - Hp) : i < 3 * Hp ? d.w3 + (i - 2 * Hp) : d.b3;
        float pv = 0.f, pm = 0.f, pvv = 0.f;
        if (is_param) { pv = c_params[pidx]; pm = c_m[pidx]; pvv = c_v[pidx]; }
        float g = 0.f;
        if (i < NE) {
            for (int s0 = wave; s0 < nparts; s0 += 16 * nw) {   // 16 loads in flight per lane: one round trip at batch 128
                float t[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int sidx = s0 + u * nw;
                    t[u] = sidx < nparts ? c_part[(size_t)sidx * stride + i] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) g += t[u];
            }
        }
        red[wave][lane] = g;
        __syncthreads();
        if (wave == 0 && i < NE) {
            g = 0.f;
            {
                float r8[SG_WGRAD_THREADS / 64];
#pragma unroll
                for (int w = 0; w < SG_WGRAD_THREADS / 64; ++w) r8[w] = red[w][lane];
#pragma unroll
                for (int w = 0; w < SG_WGRAD_THREADS / 64; ++w) g += r8[w];
            }
            if (i < 3 * Hp + 1) {
                if (has_grad_out) a.grad_out[pidx] = g;
                else {
                    pm = pm + (g - pm) * (float)(1.0 - 0.9);
                    pvv = pvv * (float)0.999 + (float)(1.0 - 0.999) * g * g;
                    const float denom = sqrtf(pvv) / bc2_sqrt + SG_DISC_ADAM_EPS;
                    c_params[pidx] = pv - step_size * (pm / denom);
                    c_m[pidx] = pm;
                    c_v[pidx] = pvv;
                }
            }
            // loss_expert, loss_policy, loss_gp sums sit in elements 3Hp+1..3 = lanes l0+1..l0+3 of the last block
            const int l0 = (3 * Hp) & 63;
            const float el_s = __shfl(g, l0 + 1), pl_s = __shfl(g, l0 + 2), gp_s = __shfl(g, l0 + 3);
            if (i == 3 * Hp) {
                if (has_grad_out) {
                    a.grad_out[d.total + 0] = el_s; a.grad_out[d.total + 1] = pl_s; a.grad_out[d.total + 2] = gp_s;
                } else {
                    // a2c/algo/gail.py:181-184: loss.item() etc. are float32, accumulated in Python doubles
                    const float el = el_s * a.inv_B, pl = pl_s * a.inv_B, gp = a.lambda_ * (gp_s * a.inv_B);
                    a.loss_acc[0] += (double)(el + pl + gp);
                    a.loss_acc[1] += (double)el;
                    a.loss_acc[2] += (double)pl;
                }
            }
        }
    }
}

// Data-parallel mode only: Adam from the all-reduced flat gradient (+ loss sums in its tail).
__global__ __launch_bounds__(256) void k_disc_adam_flat(float* params, float* m, float* v, const float* grad, int total,
                                                        const SgOptState* st, float eps, float inv_B, float lambda_,
                                                        double* loss_acc, SgDiscDesc d, float* wT, int k1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        const int t = st->t0 + k1;
        sg_adam_apply(params + i, m + i, v + i, grad[i], st->step_size2[t & 1], st->bc2_sqrt2[t & 1], eps);
        if (wT) {
            if (i < d.b1) { const int r = i / d.ldF, c = i % d.ldF; if (c < d.Fp) sg_disc_img_store(d, wT, false, r, c, params[i]); }
            else if (i >= d.w2 && i < d.b2) { const int j = i - d.w2, r = j / d.ldH, c = j % d.ldH; if (c < d.Hp) sg_disc_img_store(d, wT, true, r, c, params[i]); }
        }
    }
    if (i == 0) {
        const float el = grad[total] * inv_B, pl = grad[total + 1] * inv_B, gp = lambda_ * (grad[total + 2] * inv_B);
        loss_acc[0] += (double)(el + pl + gp);
        loss_acc[1] += (double)el;
        loss_acc[2] += (double)pl;
    }
}
