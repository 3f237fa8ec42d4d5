// sg_gemm.hpp -- LDS-resident fp32 tile GEMMs on the gfx950 matrix cores.
//
// Every dense contraction of the GAIL+PPO update (the obs x W products of the 64/100-unit tanh
// MLPs, their transposed backward products and the dW = dY^T X reductions) runs through the three
// wave-level routines below.  Operands live in LDS as row-major matrices whose dimensions are
// padded to multiples of 16 and whose leading dimension is  ld = pad16(cols) + 4  floats, i.e.
// ld == 4 (mod 8).  With that one stride rule all three access patterns are LDS-bank-conflict free:
//
//   "K-contiguous" operand (NT: both; NN: A)   lane (i = lane&15, q = lane>>4) reads 2 (ds_read_b64)
//        or 4 (ds_read_b128) consecutive floats of row i; ld/2 == 2 (mod 4) spreads the 32 lanes of
//        a b64 lane group over all 64 banks.
//   "K-strided" operand (TN: both; NN: B)      lane (j, q) reads element j of rows 4q+s, s = 0..3
//        (ds_read_b32); 4*ld == 16 (mod 32) puts the two 16-lane halves of a lane group on
//        disjoint bank halves.
//
// v_mfma_f32_16x16x4_f32 consumes k = lane>>4 (4 k-values per instruction).  Because a dot product
// does not care about the order of its terms, each routine assigns reduction indices to (q, step)
// in whatever order makes the LDS reads wide -- the SAME assignment for the A and the B operand:
//        NT:  k = 8c + 2q + s   (s = 0,1)        NN, TN:  k = 16c + 4q + s   (s = 0..3)
// (all K extents are multiples of 16).  The result is exact fp32 (each MFMA is a k-ordered fmaf
// chain), only the summation order differs from a scalar loop.
//
// Latency structure: these GEMMs are short (K <= 128, a handful of tiles per wave), so instead of
// a software-pipelined K loop each routine issues ALL LDS reads of a K block (up to 8 chunks of
// 16) first and then streams the MFMAs behind them -- one LDS latency per GEMM instead of one per
// K step.  K/16 is a runtime value dispatched to compile-time-unrolled bodies.  A lone 16x16 tile
// (MT*NT == 1) alternates between two accumulators so consecutive MFMAs never wait on the
// 40-cycle dependent-accumulator latency.
//
// C/D fragment of the 16x16 tile: acc[r] holds row 4*(lane>>4) + r, column lane&15.
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SG_PAD16(x) (((x) + 15) & ~15)
#define SG_LD(x) (SG_PAD16(x) + 4)

__device__ __forceinline__ f32x4 sg_mfma(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// 8-byte LDS read that stays a ds_read_b64.  Left alone, the compiler fuses two of them into
// ds_read2_b64, which is banked modulo 32 dwords in 16-lane groups (2-way conflict on the
// ld == 4 (mod 8) layout, and half the bytes per clock); ds_read_b64 is banked modulo 64 in
// 32-lane groups and is conflict-free on it.  volatile forbids the fusion; the compiler still
// tracks the read in its lgkmcnt bookkeeping.
__device__ __forceinline__ float2 sg_lds_read_b64(const float* p) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    typedef const volatile v2f __attribute__((address_space(3))) * lds_v2f_ptr;  // explicit LDS pointer:
    const v2f v = *(lds_v2f_ptr)(p);  // a volatile generic pointer would become flat_load
    return float2{v.x, v.y};
}

// tanh for activations: 1 - 2/(exp(2x)+1) on the fast exp/rcp units, with the odd Taylor
// polynomial below |x| = 0.1 where that form would cancel.  Absolute error < 2e-7 (fp32 round-off
// class), far inside the 1e-4 parity budget; ~4x cheaper than the libm call in the epilogues.
__device__ __forceinline__ float sg_tanh(float x) {
    const float x2 = x * x;
    const float poly = x * (1.f + x2 * (-0.33333334f + x2 * (0.13333334f + x2 * (-0.053968254f + x2 * 0.021869488f))));
    const float t = __expf(2.f * x);
    const float big = 1.f - 2.f * __builtin_amdgcn_rcpf(t + 1.f);   // v_rcp_f32 (1 ulp), not the IEEE division sequence
    return fabsf(x) < 0.1f ? poly : big;
}

template <int MT, int NT>
__device__ __forceinline__ void sg_acc_zero(f32x4 (&acc)[MT][NT]) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// Dispatch a runtime chunk count (K/16) to bodies unrolled for 1..8 chunks.
#define SG_KC_DISPATCH(kc, BODY)                                   \
    do {                                                           \
        int _kc = (kc);                                            \
        while (_kc > 8) { BODY(8); _kc -= 8; }                     \
        switch (_kc) {                                             \
            case 8: BODY(8); break; case 7: BODY(7); break;        \
            case 6: BODY(6); break; case 5: BODY(5); break;        \
            case 4: BODY(4); break; case 3: BODY(3); break;        \
            case 2: BODY(2); break; case 1: BODY(1); break;        \
            default: break;                                        \
        }                                                          \
    } while (0)

// ---- NT:  C[m][n] += sum_k A[m][k] * B[n][k].  A -> row m0 of [.][lda], B -> row n0 of [.][ldb].
// (activations x weight^T: nn.Linear forward; also g x W1^T in the gradient-penalty backward)
template <int MT, int NT, int KC>
__device__ __forceinline__ void sg_mma_nt_blk(const float*& ap, int lda, const float*& bp, int ldb,
                                              f32x4 (&acc)[MT][NT], f32x4& alt) {
    constexpr bool DUAL = (MT * NT == 1);
    float2 a[KC][2][MT], b[KC][2][NT];
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < MT; ++i) a[c][h][i] = sg_lds_read_b64(ap + i * 16 * lda + 16 * c + 8 * h);
#pragma unroll
            for (int j = 0; j < NT; ++j) b[c][h][j] = sg_lds_read_b64(bp + j * 16 * ldb + 16 * c + 8 * h);
        }
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = sg_mfma(a[c][h][i].x, b[c][h][j].x, acc[i][j]);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if (DUAL) alt = sg_mfma(a[c][h][i].y, b[c][h][j].y, alt);
                    else acc[i][j] = sg_mfma(a[c][h][i].y, b[c][h][j].y, acc[i][j]);
                }
        }
    ap += 16 * KC;
    bp += 16 * KC;
}

template <int MT, int NT>
__device__ __forceinline__ void sg_mma_nt(const float* A, int lda, const float* B, int ldb, int K,
                                          f32x4 (&acc)[MT][NT]) {
    const int lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
    const float* ap = A + li * lda + 2 * lq;
    const float* bp = B + li * ldb + 2 * lq;
    f32x4 alt = f32x4{0.f, 0.f, 0.f, 0.f};
#define SG_BODY(N) sg_mma_nt_blk<MT, NT, N>(ap, lda, bp, ldb, acc, alt)
    SG_KC_DISPATCH(K >> 4, SG_BODY);
#undef SG_BODY
    if (MT * NT == 1) acc[0][0] += alt;
}

// ---- NT, B operand in GLOBAL memory ("GW": weights read straight from L2 instead of an LDS-resident parameter image --
// the general-shape path for networks whose parameter block does not fit a CU's 160 KB of LDS, sg_policy.hip /
// sg_ppo.hip / sg_disc.hip).  Same contract as sg_mma_nt.  Reduction indices k = 16c + 4q + s (s = 0..3) for BOTH operands:
// A (LDS, K-contiguous) is one ds_read_b128 per chunk, B one 16-byte global load per chunk (rows are 16-byte aligned:
// ld is a multiple of 4); all loads of a K block are issued before its first MFMA.  NN with a global B needs no twin:
// sg_mma_nn reads B through plain pointers.
template <int MT, int NT, int KC>
__device__ __forceinline__ void sg_mma_nt_g_blk(const float*& ap, int lda, const float*& bp, int ldb,
                                                f32x4 (&acc)[MT][NT], f32x4& alt) {
    constexpr bool DUAL = (MT * NT == 1);
    float4 a[KC][MT], b[KC][NT];
#pragma unroll
    for (int c = 0; c < KC; ++c) {
#pragma unroll
        for (int j = 0; j < NT; ++j) b[c][j] = *reinterpret_cast<const float4*>(bp + (size_t)j * 16 * ldb + 16 * c);
#pragma unroll
        for (int i = 0; i < MT; ++i) a[c][i] = *reinterpret_cast<const float4*>(ap + i * 16 * lda + 16 * c);
    }
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const float av = s == 0 ? a[c][i].x : s == 1 ? a[c][i].y : s == 2 ? a[c][i].z : a[c][i].w;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const float bv = s == 0 ? b[c][j].x : s == 1 ? b[c][j].y : s == 2 ? b[c][j].z : b[c][j].w;
                    if (DUAL && (s & 1)) alt = sg_mfma(av, bv, alt);
                    else acc[i][j] = sg_mfma(av, bv, acc[i][j]);
                }
            }
    ap += 16 * KC;
    bp += 16 * KC;
}

template <int MT, int NT>
__device__ __forceinline__ void sg_mma_nt_g(const float* A, int lda, const float* B, int ldb, int K,
                                            f32x4 (&acc)[MT][NT]) {
    const int lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
    const float* ap = A + li * lda + 4 * lq;
    const float* bp = B + (size_t)li * ldb + 4 * lq;
    f32x4 alt = f32x4{0.f, 0.f, 0.f, 0.f};
#define SG_BODY(N) sg_mma_nt_g_blk<MT, NT, N>(ap, lda, bp, ldb, acc, alt)
    SG_KC_DISPATCH(K >> 4, SG_BODY);
#undef SG_BODY
    if (MT * NT == 1) acc[0][0] += alt;
}

// ---- NN:  C[m][n] += sum_k A[m][k] * B[k][n].  A -> row m0 of [.][lda]; B -> column n0 of [K][ldb].
// (dY x W: back-propagation through nn.Linear to its input)
template <int MT, int NT, int KC>
__device__ __forceinline__ void sg_mma_nn_blk(const float*& ap, int lda, const float*& bp, int ldb,
                                              f32x4 (&acc)[MT][NT], f32x4& alt) {
    constexpr bool DUAL = (MT * NT == 1);
    float4 a[KC][MT];
    float b[KC][4][NT];
#pragma unroll
    for (int c = 0; c < KC; ++c) {
#pragma unroll
        for (int i = 0; i < MT; ++i) a[c][i] = *reinterpret_cast<const float4*>(ap + i * 16 * lda + 16 * c);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < NT; ++j) b[c][s][j] = bp[(16 * c + s) * ldb + j * 16];
    }
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const float av = s == 0 ? a[c][i].x : s == 1 ? a[c][i].y : s == 2 ? a[c][i].z : a[c][i].w;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if (DUAL && (s & 1)) alt = sg_mfma(av, b[c][s][j], alt);
                    else acc[i][j] = sg_mfma(av, b[c][s][j], acc[i][j]);
                }
            }
    ap += 16 * KC;
    bp += 16 * KC * ldb;
}

template <int MT, int NT>
__device__ __forceinline__ void sg_mma_nn(const float* A, int lda, const float* B, int ldb, int K,
                                          f32x4 (&acc)[MT][NT]) {
    const int lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
    const float* ap = A + li * lda + 4 * lq;
    const float* bp = B + (4 * lq) * ldb + li;
    f32x4 alt = f32x4{0.f, 0.f, 0.f, 0.f};
#define SG_BODY(N) sg_mma_nn_blk<MT, NT, N>(ap, lda, bp, ldb, acc, alt)
    SG_KC_DISPATCH(K >> 4, SG_BODY);
#undef SG_BODY
    if (MT * NT == 1) acc[0][0] += alt;
}

// ---- TN:  C[m][n] += sum_r A[r][m] * B[r][n].  A -> column m0 of [K][lda]; B -> column n0 of [K][ldb].
// (dW = dY^T X: the weight-gradient reduction over the rows of a tile)
template <int MT, int NT, int KC>
__device__ __forceinline__ void sg_mma_tn_blk(const float*& ap, int lda, const float*& bp, int ldb,
                                              f32x4 (&acc)[MT][NT], f32x4& alt) {
    constexpr bool DUAL = (MT * NT == 1);
    float a[KC][4][MT], b[KC][4][NT];
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int i = 0; i < MT; ++i) a[c][s][i] = ap[(16 * c + s) * lda + i * 16];
#pragma unroll
            for (int j = 0; j < NT; ++j) b[c][s][j] = bp[(16 * c + s) * ldb + j * 16];
        }
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if (DUAL && (s & 1)) alt = sg_mfma(a[c][s][i], b[c][s][j], alt);
                    else acc[i][j] = sg_mfma(a[c][s][i], b[c][s][j], acc[i][j]);
                }
    ap += 16 * KC * lda;
    bp += 16 * KC * ldb;
}

template <int MT, int NT>
__device__ __forceinline__ void sg_mma_tn(const float* A, int lda, const float* B, int ldb, int K,
                                          f32x4 (&acc)[MT][NT]) {
    const int lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
    const float* ap = A + (4 * lq) * lda + li;
    const float* bp = B + (4 * lq) * ldb + li;
    f32x4 alt = f32x4{0.f, 0.f, 0.f, 0.f};
#define SG_BODY(N) sg_mma_tn_blk<MT, NT, N>(ap, lda, bp, ldb, acc, alt)
    SG_KC_DISPATCH(K >> 4, SG_BODY);
#undef SG_BODY
    if (MT * NT == 1) acc[0][0] += alt;
}

// Visit the 4 accumulator elements this lane owns in the tile whose top-left is (row0, col0).
template <typename F>
__device__ __forceinline__ void sg_tile_foreach(const f32x4& acc, int row0, int col0, F&& f) {
    const int lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) f(row0 + 4 * lq + r, col0 + li, acc[r]);
}

// Column sum over all 16*MT rows of a column block: v[i][r] are this lane's (post-epilogue) values
// of row tile i, register r.  Every lane returns the total of column (lane & 15).
template <int MT>
__device__ __forceinline__ float sg_tile_colsum(const float (&v)[MT][4]) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    return s;
}

// ---------------------------------------------------------------------------------------------
// Workgroup-level layer routines.  All waves of the workgroup call them; column tiles of the
// output are dealt round-robin to waves.  R = 16*MT rows are processed per call.
// The *_t variants hand the epilogue a whole column block: ep(tn, acc[MT][1]).
// ---------------------------------------------------------------------------------------------

// GW = true: W points into global memory (sg_mma_nt_g); the NN forms read W through plain pointers either way.
template <int MT, bool GW = false, typename EP>
__device__ __forceinline__ void sg_layer_nt_t(const float* in, int ldi, const float* W, int ldw,
                                              int K, int Np, EP&& ep) {
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int tn = wave; tn * 16 < Np; tn += nw) {
        f32x4 acc[MT][1];
        sg_acc_zero(acc);
        if (GW) sg_mma_nt_g<MT, 1>(in, ldi, W + (size_t)tn * 16 * ldw, ldw, K, acc);
        else sg_mma_nt<MT, 1>(in, ldi, W + tn * 16 * ldw, ldw, K, acc);
        ep(tn, acc);
    }
}

template <int MT, typename EP>
__device__ __forceinline__ void sg_layer_nn_t(const float* dY, int ldy, const float* W, int ldw,
                                              int K, int Np, EP&& ep) {
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int tn = wave; tn * 16 < Np; tn += nw) {
        f32x4 acc[MT][1];
        sg_acc_zero(acc);
        sg_mma_nn<MT, 1>(dY, ldy, W + tn * 16, ldw, K, acc);
        ep(tn, acc);
    }
}

// out[r][c] = ep(r, c, sum_k in[r][k] * W[c][k])  for c < Np   (forward layer / "NT")
template <int MT, bool GW = false, typename EP>
__device__ __forceinline__ void sg_layer_nt(const float* in, int ldi, const float* W, int ldw,
                                            int K, int Np, EP&& ep) {
    sg_layer_nt_t<MT, GW>(in, ldi, W, ldw, K, Np, [&](int tn, f32x4 (&acc)[MT][1]) {
#pragma unroll
        for (int i = 0; i < MT; ++i) sg_tile_foreach(acc[i][0], i * 16, tn * 16, ep);
    });
}

// out[r][c] = ep(r, c, sum_k dY[r][k] * W[k][c])  for c < Np   (backward to the layer input / "NN")
template <int MT, typename EP>
__device__ __forceinline__ void sg_layer_nn(const float* dY, int ldy, const float* W, int ldw,
                                            int K, int Np, EP&& ep) {
    sg_layer_nn_t<MT>(dY, ldy, W, ldw, K, Np, [&](int tn, f32x4 (&acc)[MT][1]) {
#pragma unroll
        for (int i = 0; i < MT; ++i) sg_tile_foreach(acc[i][0], i * 16, tn * 16, ep);
    });
}

// Unit-distributed forms: a task is ONE 16x16 output tile (row tile i, column tile tn), dealt round-robin to the waves.
// With 8 waves and 32 rows x 64 columns every wave gets one tile (half the MFMA chain of the column-strip forms above).
template <int MT, bool GW = false, typename EP>
__device__ __forceinline__ void sg_layer_nt_u(const float* in, int ldi, const float* W, int ldw, int K, int Np, EP&& ep) {
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6, ntl = Np >> 4;
    for (int u = wave; u < MT * ntl; u += nw) {
        const int i = u / ntl, tn = u - i * ntl;
        f32x4 acc[1][1];
        sg_acc_zero(acc);
        if (GW) sg_mma_nt_g<1, 1>(in + i * 16 * ldi, ldi, W + (size_t)tn * 16 * ldw, ldw, K, acc);
        else sg_mma_nt<1, 1>(in + i * 16 * ldi, ldi, W + tn * 16 * ldw, ldw, K, acc);
        sg_tile_foreach(acc[0][0], i * 16, tn * 16, ep);
    }
}
template <int MT, typename EP>
__device__ __forceinline__ void sg_layer_nn_u(const float* dY, int ldy, const float* W, int ldw, int K, int Np, EP&& ep) {
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6, ntl = Np >> 4;
    for (int u = wave; u < MT * ntl; u += nw) {
        const int i = u / ntl, tn = u - i * ntl;
        f32x4 acc[1][1];
        sg_acc_zero(acc);
        sg_mma_nn<1, 1>(dY + i * 16 * ldy, ldy, W + tn * 16, ldw, K, acc);
        sg_tile_foreach(acc[0][0], i * 16, tn * 16, ep);
    }
}

// ---- weight gradient ("TN"):  G[m][n] (+)= sum_{r<R} dY[r][m] X[r][n]  (+ sum_r dY2[r][m] X2[r][n])
// written to global memory.  Each wave owns whole 16-row panels of G: it reads its A fragment
// (16 columns of dY) once, all B fragments of up to 8 column tiles, issues every LDS read before
// the first MFMA, accumulates both contributions in registers and stores the panel once.  The
// optional second operand pair lets a gradient with two terms (e.g. dW2 = d2^T bu1 + z2b^T h1 in
// the gradient penalty) be formed without a read-modify-write of G.   R = 16*KC rows.
template <int KC, int NTN, bool TWO, bool NT = true>
__device__ __forceinline__ void sg_tn_panel(const float* ap, const float* ap2, int lda, const float* bp,
                                            const float* bp2, int ldb, float* G, int ldg, bool accumulate) {
    float a[TWO ? 2 : 1][KC][4], b[TWO ? 2 : 1][KC][4][NTN];
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            a[0][c][s] = ap[(16 * c + s) * lda];
#pragma unroll
            for (int j = 0; j < NTN; ++j) b[0][c][s][j] = bp[(16 * c + s) * ldb + 16 * j];
            if (TWO) {
                a[TWO ? 1 : 0][c][s] = ap2[(16 * c + s) * lda];
#pragma unroll
                for (int j = 0; j < NTN; ++j) b[TWO ? 1 : 0][c][s][j] = bp2[(16 * c + s) * ldb + 16 * j];
            }
        }
    f32x4 acc[NTN];
#pragma unroll
    for (int j = 0; j < NTN; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int src = 0; src < (TWO ? 2 : 1); ++src)
#pragma unroll
        for (int c = 0; c < KC; ++c)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < NTN; ++j) acc[j] = sg_mfma(b[src][c][s][j], a[src][c][s], acc[j]);
    // The operands are SWAPPED (the products commute, the sums and their order are the same): the matrix core then leaves
    // the tile transposed in the accumulators -- lane (li, lq) holds G[li][16 j + 4 lq .. + 3], four consecutive floats of one
    // row -- so a tile goes out as one 16-byte store per lane (16 row segments of 64 bytes per instruction) instead of four
    // 4-byte stores.  Round 3: the weight-gradient phases of k_ppo_bwd were bound by the issue of their dword stores.
    const int lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
    typedef float sg_f4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int j = 0; j < NTN; ++j) {
        sg_f4* p = reinterpret_cast<sg_f4*>(G + li * ldg + 16 * j + 4 * lq);
        sg_f4 v = sg_f4{acc[j][0], acc[j][1], acc[j][2], acc[j][3]};
        // a fresh gradient panel is written once and read by a later kernel: a streaming store lets it drain
        // to memory while the kernel runs instead of sitting dirty in L2 until the end-of-kernel write-back
        // (NT = false: G is an LDS buffer -- k_ppo_small keeps the gradient on chip -- where a streaming store does not exist)
        if (accumulate) *p = *p + v;
        else if (NT) __builtin_nontemporal_store(v, p);
        else {   // (element by element: a plain 16-byte vector store of MFMA results does not get through this compiler's back end)
            float* q = G + li * ldg + 16 * j + 4 * lq;
            q[0] = acc[j][0]; q[1] = acc[j][1]; q[2] = acc[j][2]; q[3] = acc[j][3];
        }
    }
}

// NW_HINT: the number of waves the caller launches with when that is more than 4 (a compile-time hint, so that with
// compile-time Mp / Np the strip width and the panel switch below still fold to one case; 0 = 8-tile strips).
template <int KC, int NW_HINT = 0, bool NT = true>
__device__ __forceinline__ void sg_grad_tn2(const float* dY, int ldy, const float* X, int ldx,
                                            const float* dY2, const float* X2, int Mp, int Np,
                                            float* G, int ldg, bool accumulate) {
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
    // column tiles per wave-task: 8 (one A fragment feeds the whole strip), narrower when that would leave waves idle
    const int tm_n = Mp >> 4, tn_n = Np >> 4;
    int bw = 8;
    if (NW_HINT > 4) while (bw > 1 && tm_n * ((tn_n + bw - 1) / bw) < NW_HINT) bw >>= 1;   // (no gain for 4-wave workgroups, measured)
    const int blocks_n = (tn_n + bw - 1) / bw;
    for (int t = wave; t < tm_n * blocks_n; t += nw) {
        const int tm = t / blocks_n, tb = t % blocks_n;
        const int ntn = tn_n - bw * tb < bw ? tn_n - bw * tb : bw;
        const float* ap = dY + (4 * lq) * ldy + tm * 16 + li;
        const float* bp = X + (4 * lq) * ldx + tb * bw * 16 + li;
        const float* ap2 = dY2 ? dY2 + (4 * lq) * ldy + tm * 16 + li : ap;
        const float* bp2 = X2 ? X2 + (4 * lq) * ldx + tb * bw * 16 + li : bp;
        float* g = G + (size_t)(tm * 16) * ldg + tb * bw * 16;
#define SG_PANEL(N)                                                                              \
    do {                                                                                         \
        if (dY2) sg_tn_panel<KC, N, true, NT>(ap, ap2, ldy, bp, bp2, ldx, g, ldg, accumulate);   \
        else sg_tn_panel<KC, N, false, NT>(ap, ap2, ldy, bp, bp2, ldx, g, ldg, accumulate);      \
    } while (0)
        switch (ntn) {
            case 8: SG_PANEL(8); break; case 7: SG_PANEL(7); break; case 6: SG_PANEL(6); break;
            case 5: SG_PANEL(5); break; case 4: SG_PANEL(4); break; case 3: SG_PANEL(3); break;
            case 2: SG_PANEL(2); break; case 1: SG_PANEL(1); break; default: break;
        }
#undef SG_PANEL
    }
}

template <int KC, int NW_HINT = 0, bool NT = true>
__device__ __forceinline__ void sg_grad_tn(const float* dY, int ldy, const float* X, int ldx, int Mp,
                                           int Np, float* G, int ldg, bool accumulate) {
    sg_grad_tn2<KC, NW_HINT, NT>(dY, ldy, X, ldx, nullptr, nullptr, Mp, Np, G, ldg, accumulate);
}

// sum over the 16 lanes of one DPP row, result in all 16 lanes: quad butterflies, then the half-row and row mirrors
// (every lane of a quad already holds the quad's sum) -- four VALU instructions, no LDS crossbar
template <int CTRL>
__device__ __forceinline__ float sg_dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float sg_rowsum16(float v) {
    v = sg_dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
    v = sg_dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
    v = sg_dpp_add<0x141>(v);   // row_half_mirror
    v = sg_dpp_add<0x140>(v);   // row_mirror
    return v;
}

// g[c] (+)= sum_{r<R} M[r][c]  for c < Np  (bias gradients).  16 lanes per column (one DPP row), each summing every
// 16th row with independent LDS reads, then a row reduction: a column costs R/16 reads of latency instead of R.
__device__ __forceinline__ void sg_colsum(const float* M, int ldm, int R, int Np, float* g,
                                          bool accumulate) {
    const int sub = threadIdx.x & 15;
    for (int c = threadIdx.x >> 4; c < Np; c += blockDim.x >> 4) {
        float s = 0.f;
        for (int r = sub; r < R; r += 16) s += M[r * ldm + c];
        s = sg_rowsum16(s);
        if (sub == 0) g[c] = accumulate ? g[c] + s : s;
    }
}

// Linear copy global -> LDS, 16 bytes per lane (n4 = number of float4), 8 loads in flight per
// lane per round (all rounds when n4 is a compile-time constant and the loop unrolls).
typedef unsigned int sg_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 sg_buffer_load4(__amdgpu_buffer_rsrc_t rsrc, int byte_offset) {
    const sg_u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_offset, 0, 0);
    return float4{__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w)};
}
__device__ __forceinline__ void sg_stage(float* lds, const float* __restrict__ g, int n4) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g), 0, n4 * 16, 0x00020000);
    float4* dst = reinterpret_cast<float4*>(lds);
    const int nt = blockDim.x;
#pragma unroll 2
    for (int base = threadIdx.x; base < n4; base += 8 * nt) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = sg_buffer_load4(rsrc, (base + u * nt) * 16);   // range-checked: zeros past n4
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * nt;
            if (i < n4) dst[i] = v[u];
        }
    }
}

// Split form of sg_stage: issue the global loads of the first U*blockDim float4 into registers
// now, commit them to LDS later, so the memory round trip overlaps whatever is placed in between
// (the dependent index->row gather of a minibatch).  The tail beyond U*blockDim, if any, is copied
// by sg_stage_commit itself.
// The loads are buffer loads: the hardware range-checks every lane against the n4*16-byte extent and returns
// zeros beyond it, so there is no per-load guard for the compiler to turn into a branch (which would make
// each request wait for the previous one).
template <int U>
__device__ __forceinline__ void sg_stage_issue(float4 (&v)[U], const float* __restrict__ g, int n4) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g), 0, n4 * 16, 0x00020000);
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = sg_buffer_load4(rsrc, (int)(threadIdx.x + u * blockDim.x) * 16);
}
template <int U>
__device__ __forceinline__ void sg_stage_commit(float* lds, const float4 (&v)[U], const float* __restrict__ g, int n4) {
    float4* dst = reinterpret_cast<float4*>(lds);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = threadIdx.x + u * blockDim.x;
        if (i < n4) dst[i] = v[u];
    }
    const int done = U * blockDim.x;
    if (n4 > done) sg_stage(lds + 4 * done, g + 4 * done, n4 - done);
}

__device__ __forceinline__ float sg_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double sg_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
