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

// Weight images.  A matrix M[n][k] (n < 16*waves, K a multiple of 16) that multiplies in[.][k] is
// kept in global memory in the order the lanes consume it: float4 number ((n>>4)*K/16 + t)*64 + lane
// holds M[n][s*K/4 + 4t .. +3] for lane = 16 s + (n & 15).  One load instruction of a wave is then one
// contiguous KiB (8 cache lines) instead of 64 scattered 16-byte pieces.  For y = x W^T the image is
// built from W, for y = d W from W^T; k_disc_wgrad writes both images next to the canonical layout.
__host__ __device__ __forceinline__ int sg4_img_index(int n, int k, int K) {
    const int KS = K >> 2, s = (k >= KS) + (k >= 2 * KS) + (k >= 3 * KS), ko = k - s * KS;   // s = k / KS without a division
    return ((((n >> 4) * (K >> 4) + (ko >> 2)) * 64 + 16 * s + (n & 15)) << 2) + (ko & 3);
}

// This lane's B operand for the whole kernel: wave-th 16-column slice of an image.
template <int K>
__device__ __forceinline__ void sg4_load_w(float4 (&w)[SG4_NW(K)], const float* img, int wave, int lane) {
#if defined(SG4_W_SC1) && SG4_W_SC1   // diagnostic builds: the weight slices past the CU's L1 (and, with sc1, past stale L2 copies)
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(img), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int t = 0; t < SG4_NW(K); ++t) {
        const u32x4_ v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)((((size_t)wave * SG4_NW(K) + t) * 64 + lane) * 16), 0, 16);
        w[t] = float4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
    }
#else
    const float4* p = reinterpret_cast<const float4*>(img) + (size_t)wave * SG4_NW(K) * 64 + lane;
#pragma unroll
    for (int t = 0; t < SG4_NW(K); ++t) w[t] = p[t * 64];
#endif
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

#ifndef SG4_TIMING_SKIP
#define SG4_TIMING_SKIP 0        // diagnostic builds (WRONG results, timing only): 1 = every other group of four MFMAs is not issued
#endif
#define SG4_STEP(CC)                                                                 \
    if (SG4_TIMING_SKIP && ((CC) & 1)) {   /* the registers stay loaded, the MFMAs are not issued */ \
        if (4 * tc + (CC) < SG4_NW(K)) { const float4 wk = w[4 * tc + (CC) < SG4_NW(K) ? 4 * tc + (CC) : 0]; asm volatile("" ::"v"(wk.x), "v"(wk.y), "v"(wk.z), "v"(wk.w)); } \
    } else if (4 * tc + (CC) < SG4_NW(K)) {                                          \
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
    // Sum the four K slices (lanes l, l^16, l^32, l^48) and leave row (lane >> 4) in each lane, with three
    // cross-lane swaps on the VALU instead of eight LDS permutes: after the 32-lane swap the lower half
    // carries rows 0/1 and the upper half rows 2/3; the 16-lane row swap finishes both sums at once.
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
        const f32x4 v = acc[rg][0] + acc[rg][1];
        const auto p02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[0]), __float_as_uint(v[2]), false, false);
        const auto p13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[1]), __float_as_uint(v[3]), false, false);
        const float e = __uint_as_float(p02[0]) + __uint_as_float(p02[1]);   // [row0 | row2] summed over s, s^2
        const float o = __uint_as_float(p13[0]) + __uint_as_float(p13[1]);   // [row1 | row3]
        const auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(e), __float_as_uint(o), false, false);
        out[rg] = __uint_as_float(q[0]) + __uint_as_float(q[1]);
    }
    (void)lane;
}

// sum of one value per lane over the four K-slice lanes (l, l^16, l^32, l^48) that hold the four rows of a
// column: two VALU lane swaps (upper/lower half, then odd/even 16-lane rows), result in all four lanes
__device__ __forceinline__ float sg4_colsum(float v) {
    const auto p = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(p[0]) + __uint_as_float(p[1]);
    const auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
// sum over the 16 lanes (one DPP row) that hold one row's 16 columns of this wave, result in all 16 lanes (sg_gemm.hpp)
__device__ __forceinline__ float sg4_rowsum16(float v) { return sg_rowsum16(v); }
