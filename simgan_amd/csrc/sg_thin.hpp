// sg_thin.hpp -- thin-row GEMM engine: out[4*RG rows][16 cols per wave] = in[4*RG][K] . M[n][k]^T
// on v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 outer products per instruction, exact fp32).
//
// Why: a discriminator step at batch 128 is a chain of dependent [rows x 112] x [112 x 112] GEMMs.
// With 16x16x4 tiles the smallest workgroup owns 16 rows, so the whole step runs on 16 of the 256
// CUs and each phase is bound by one CU's MFMA pipe.  With 4-row blocks the same step spreads over
// 4x the CUs, each wave keeps ITS slice of every weight matrix in registers for the whole kernel
// (loaded once from L2, never staged through LDS), and LDS only carries the 4-row activations.
//
// Lane roles (lane l of a wave that owns output columns [n0, n0+16)):
//     s = l >> 4      K slice: this lane multiplies k in [s*K/4, (s+1)*K/4)
//     c = (l >> 2)&3  column quad, j = l & 3: B operand / result column n0 + 4c + j = n0 + (l & 15)
//     i = l & 3       A operand row
// The MFMA's block index is l >> 2 = 4s + c.  With CBSZ = 2 the A operand of block ABID of each
// group of four blocks (= one K slice) is broadcast to the group, so ONE A register carries four
// different k (one per column-quad position) and a single 16-byte LDS read per lane feeds 16
// MFMAs.  After the K loop the four slices' partial sums sit in lanes l, l^16, l^32, l^48.
#pragma once
#include "sg_gemm.hpp"

#define SG4_NCH(K) (((K) / 4 + 15) / 16)   // 16-k chunks per slice
#define SG4_NW(K) ((K) / 16)               // float4 weight registers per lane

template <int ABID>
__device__ __forceinline__ f32x4 sg4_mfma(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 2, ABID, 0);
}

// This lane's B operand for the whole kernel: row n of the row-major matrix M (leading dimension
// ld floats, a multiple of 4), K slice s.  M[n][k] multiplies in[.][k]: for y = x W^T pass W, for
// y = d W pass the stored transpose of W.
template <int K>
__device__ __forceinline__ void sg4_load_w(float4 (&w)[SG4_NW(K)], const float* M, int ld, int n, int lane) {
    const float4* p = reinterpret_cast<const float4*>(M + (size_t)n * ld + (lane >> 4) * (K / 4));
#pragma unroll
    for (int t = 0; t < SG4_NW(K); ++t) w[t] = p[t];
}

// A operand of RG row groups from X (LDS or global, leading dimension ldx, 16-byte aligned rows).
// Row group rg starts at row rg * rg_rows of X.
template <int K, int RG>
__device__ __forceinline__ void sg4_load_a(float4 (&a)[RG][SG4_NCH(K)], const float* X, int ldx, int lane, int rg_rows = 4) {
    constexpr int KS = K / 4;
    const int s = lane >> 4, cp = (lane >> 2) & 3, i = lane & 3;
#pragma unroll
    for (int rg = 0; rg < RG; ++rg)
#pragma unroll
        for (int tc = 0; tc < SG4_NCH(K); ++tc) {
            const int off = 16 * tc + 4 * cp;
            a[rg][tc] = (off < KS) ? *reinterpret_cast<const float4*>(X + (size_t)(rg_rows * rg + i) * ldx + s * KS + off)
                                   : float4{0.f, 0.f, 0.f, 0.f};
        }
}

#define SG4_STEP(CC)                                                                 \
    if (4 * tc + (CC) < SG4_NW(K)) {                                                 \
        const float4 wq = w[4 * tc + (CC) < SG4_NW(K) ? 4 * tc + (CC) : 0];          \
        _Pragma("unroll") for (int rg = 0; rg < RG; ++rg) {                          \
            acc[rg][0] = sg4_mfma<CC>(a[rg][tc].x, wq.x, acc[rg][0]);                \
            acc[rg][1] = sg4_mfma<CC>(a[rg][tc].y, wq.y, acc[rg][1]);                \
            acc[rg][0] = sg4_mfma<CC>(a[rg][tc].z, wq.z, acc[rg][0]);                \
            acc[rg][1] = sg4_mfma<CC>(a[rg][tc].w, wq.w, acc[rg][1]);                \
        }                                                                            \
    }

// out[rg] = this lane's result element: row 4*rg + (lane >> 4), column n0 + (lane & 15).
template <int K, int RG>
__device__ __forceinline__ void sg4_mma(const float4 (&a)[RG][SG4_NCH(K)], const float4 (&w)[SG4_NW(K)], int lane,
                                        float (&out)[RG]) {
    f32x4 acc[RG][2];
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) acc[rg][0] = acc[rg][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tc = 0; tc < SG4_NCH(K); ++tc) {
        SG4_STEP(0)
        SG4_STEP(1)
        SG4_STEP(2)
        SG4_STEP(3)
    }
    const int s = lane >> 4;
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
        f32x4 v = acc[rg][0] + acc[rg][1];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[r] += __shfl_xor(v[r], 16);
            v[r] += __shfl_xor(v[r], 32);
        }
        out[rg] = s == 0 ? v[0] : s == 1 ? v[1] : s == 2 ? v[2] : v[3];
    }
}

// sum of one value per lane over the four K-slice lanes that hold the four rows of a column
__device__ __forceinline__ float sg4_colsum(float v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
// sum over the 16 lanes that hold one row's 16 columns of this wave
__device__ __forceinline__ float sg4_rowsum16(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    return v;
}
