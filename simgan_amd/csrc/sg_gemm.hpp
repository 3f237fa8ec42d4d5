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
// The result is exact fp32 (each MFMA is a k-ordered fmaf chain), only the summation order differs
// from a scalar loop.
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

template <int MT, int NT>
__device__ __forceinline__ void sg_acc_zero(f32x4 (&acc)[MT][NT]) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// C[m][n] += sum_k A[m][k] * B[n][k].   A -> row m0 of [.][lda], B -> row n0 of [.][ldb]; K % 8 == 0.
// (activations x weight^T: nn.Linear forward; also g x W1^T in the gradient-penalty backward)
template <int MT, int NT>
__device__ __forceinline__ void sg_mma_nt(const float* A, int lda, const float* B, int ldb, int K,
                                          f32x4 (&acc)[MT][NT]) {
    const int lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
    const float* ap = A + li * lda + 2 * lq;
    const float* bp = B + li * ldb + 2 * lq;
    for (int k = 0; k < K; k += 8) {
        float2 a[MT], b[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const float2*>(ap + i * 16 * lda + k);
#pragma unroll
        for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const float2*>(bp + j * 16 * ldb + k);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = sg_mfma(a[i].x, b[j].x, acc[i][j]);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = sg_mfma(a[i].y, b[j].y, acc[i][j]);
    }
}

// C[m][n] += sum_k A[m][k] * B[k][n].   A -> row m0 of [.][lda]; B -> column n0 of [K][ldb]; K % 16 == 0.
// (dY x W: back-propagation through nn.Linear to its input)
template <int MT, int NT>
__device__ __forceinline__ void sg_mma_nn(const float* A, int lda, const float* B, int ldb, int K,
                                          f32x4 (&acc)[MT][NT]) {
    const int lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
    const float* ap = A + li * lda + 4 * lq;
    const float* bp = B + (4 * lq) * ldb + li;
    for (int k = 0; k < K; k += 16) {
        float4 a[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const float4*>(ap + i * 16 * lda + k);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float b[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) b[j] = bp[(k + s) * ldb + j * 16];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const float av = s == 0 ? a[i].x : s == 1 ? a[i].y : s == 2 ? a[i].z : a[i].w;
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = sg_mfma(av, b[j], acc[i][j]);
            }
        }
    }
}

// C[m][n] += sum_r A[r][m] * B[r][n].   A -> column m0 of [K][lda]; B -> column n0 of [K][ldb]; K % 16 == 0.
// (dW = dY^T X: the weight-gradient reduction over the rows of a tile)
template <int MT, int NT>
__device__ __forceinline__ void sg_mma_tn(const float* A, int lda, const float* B, int ldb, int K,
                                          f32x4 (&acc)[MT][NT]) {
    const int lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
    const float* ap = A + (4 * lq) * lda + li;
    const float* bp = B + (4 * lq) * ldb + li;
    for (int k = 0; k < K; k += 16) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float a[MT], b[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) a[i] = ap[(k + s) * lda + i * 16];
#pragma unroll
            for (int j = 0; j < NT; ++j) b[j] = bp[(k + s) * ldb + j * 16];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = sg_mfma(a[i], b[j], acc[i][j]);
        }
    }
}

// Visit the 4 accumulator elements this lane owns in the tile whose top-left is (row0, col0).
template <typename F>
__device__ __forceinline__ void sg_tile_foreach(const f32x4& acc, int row0, int col0, F&& f) {
    const int lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) f(row0 + 4 * lq + r, col0 + li, acc[r]);
}

// ---------------------------------------------------------------------------------------------
// Workgroup-level layer routines.  All waves of the workgroup call them; column tiles of the
// output are dealt round-robin to waves.  R = 16*MT rows are processed per call.
// ---------------------------------------------------------------------------------------------

// out[r][c] = ep(r, c, sum_k in[r][k] * W[c][k])  for c < Np   (forward layer / "NT")
template <int MT, typename EP>
__device__ __forceinline__ void sg_layer_nt(const float* in, int ldi, const float* W, int ldw,
                                            int K, int Np, EP&& ep) {
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int tn = wave; tn * 16 < Np; tn += nw) {
        f32x4 acc[MT][1];
        sg_acc_zero(acc);
        sg_mma_nt<MT, 1>(in, ldi, W + tn * 16 * ldw, ldw, K, acc);
#pragma unroll
        for (int i = 0; i < MT; ++i) sg_tile_foreach(acc[i][0], i * 16, tn * 16, ep);
    }
}

// out[r][c] = ep(r, c, sum_k dY[r][k] * W[k][c])  for c < Np   (backward to the layer input / "NN")
template <int MT, typename EP>
__device__ __forceinline__ void sg_layer_nn(const float* dY, int ldy, const float* W, int ldw,
                                            int K, int Np, EP&& ep) {
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int tn = wave; tn * 16 < Np; tn += nw) {
        f32x4 acc[MT][1];
        sg_acc_zero(acc);
        sg_mma_nn<MT, 1>(dY, ldy, W + tn * 16, ldw, K, acc);
#pragma unroll
        for (int i = 0; i < MT; ++i) sg_tile_foreach(acc[i][0], i * 16, tn * 16, ep);
    }
}

// G[m][n] (+)= sum_{r<R} dY[r][m] * X[r][n]  for m < Mp, n < Np, written to global memory
// (weight gradient / "TN").  Tiles are dealt to waves in pairs along n to share the A fragment.
__device__ __forceinline__ void sg_grad_tn(const float* dY, int ldy, const float* X, int ldx, int R,
                                           int Mp, int Np, float* G, int ldg, bool accumulate) {
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int tm_n = Mp >> 4, tn_n = Np >> 4, pairs_n = (tn_n + 1) >> 1;
    for (int t = wave; t < tm_n * pairs_n; t += nw) {
        const int tm = t / pairs_n, tp = t % pairs_n;
        const int tn0 = tp * 2, tn1 = (tn0 + 1 < tn_n) ? tn0 + 1 : tn0;  // odd tail: redo tn0, skip store
        f32x4 acc[1][2];
        sg_acc_zero(acc);
        {
            const int lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
            const float* ap = dY + (4 * lq) * ldy + tm * 16 + li;
            const float* b0 = X + (4 * lq) * ldx + tn0 * 16 + li;
            const float* b1 = X + (4 * lq) * ldx + tn1 * 16 + li;
            for (int k = 0; k < R; k += 16) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float a = ap[(k + s) * ldy];
                    acc[0][0] = sg_mfma(a, b0[(k + s) * ldx], acc[0][0]);
                    acc[0][1] = sg_mfma(a, b1[(k + s) * ldx], acc[0][1]);
                }
            }
        }
        auto st = [&](int r, int c, float v) {
            float* p = G + r * ldg + c;
            *p = accumulate ? *p + v : v;
        };
        sg_tile_foreach(acc[0][0], tm * 16, tn0 * 16, st);
        if (tn1 != tn0) sg_tile_foreach(acc[0][1], tm * 16, tn1 * 16, st);
    }
}

// g[c] (+)= sum_{r<R} M[r][c]  for c < Np  (bias gradients), one thread per column.
__device__ __forceinline__ void sg_colsum(const float* M, int ldm, int R, int Np, float* g,
                                          bool accumulate) {
    for (int c = threadIdx.x; c < Np; c += blockDim.x) {
        float s = 0.f;
        for (int r = 0; r < R; ++r) s += M[r * ldm + c];
        g[c] = accumulate ? g[c] + s : s;
    }
}

// Linear copy global -> LDS, 16 bytes per lane (n4 = number of float4).
__device__ __forceinline__ void sg_stage(float* lds, const float* __restrict__ g, int n4) {
    const float4* src = reinterpret_cast<const float4*>(g);
    float4* dst = reinterpret_cast<float4*>(lds);
    for (int i = threadIdx.x; i < n4; i += blockDim.x) dst[i] = src[i];
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
