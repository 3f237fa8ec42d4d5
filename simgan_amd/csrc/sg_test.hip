// sg_test.hip -- test hook for the LDS/MFMA tile engine (tests/test_gemm_engine.py).
#include <vector>

#include "sg_common.h"

struct GemmTestArgs {
    int mode, M, N, K;
    const float *A, *B;
    float* C;
};

// mode 0 (NT): A[M,K] B[N,K];  mode 1 (NN): A[M,K] B[K,N];  mode 2 (TN): A[K,M] B[K,N];  C[M,N]
template <int MT>
__global__ __launch_bounds__(256) void k_gemm_test(GemmTestArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int ar = a.mode == 2 ? a.K : a.M, ac = a.mode == 2 ? a.M : a.K;
    const int br = a.mode == 0 ? a.N : a.K, bc = a.mode == 0 ? a.K : a.N;
    const int lda = SG_LD(ac), ldb = SG_LD(bc), ldc = SG_LD(a.N);
    float* As = smem;
    float* Bs = As + ar * lda;
    float* Cs = Bs + br * ldb;
    for (int i = threadIdx.x; i < ar * ac; i += blockDim.x) As[(i / ac) * lda + i % ac] = a.A[i];
    for (int i = threadIdx.x; i < br * bc; i += blockDim.x) Bs[(i / bc) * ldb + i % bc] = a.B[i];
    __syncthreads();
    if (a.mode == 0)
        sg_layer_nt<MT>(As, lda, Bs, ldb, a.K, a.N, [&](int r, int c, float v) { Cs[r * ldc + c] = v; });
    else if (a.mode == 1)
        sg_layer_nn<MT>(As, lda, Bs, ldb, a.K, a.N, [&](int r, int c, float v) { Cs[r * ldc + c] = v; });
    else {
        sg_grad_tn(As, lda, Bs, ldb, a.K, a.M, a.N, a.C, a.N, false);
        sg_grad_tn(As, lda, Bs, ldb, a.K, a.M, a.N, a.C, a.N, true);   // exercises the accumulate path: C = 2*A^T B
        return;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < a.M * a.N; i += blockDim.x) a.C[i] = Cs[(i / a.N) * ldc + i % a.N];
}

extern "C" int sg_test_gemm(sg_ctx* ctx, int mode, int M, int N, int K, const float* A, const float* B, float* C) {
    SG_REQUIRE(ctx && A && B && C, "sg_test_gemm: NULL argument");
    SG_REQUIRE(mode >= 0 && mode <= 2, "sg_test_gemm: mode must be 0 (NT), 1 (NN) or 2 (TN)");
    SG_REQUIRE(M % 16 == 0 && N % 16 == 0 && K % 16 == 0 && M > 0 && N > 0 && K > 0, "sg_test_gemm: dims must be positive multiples of 16");
    SG_REQUIRE(mode == 2 || M == 16 || M == 32 || M == 64, "sg_test_gemm: M must be 16, 32 or 64 for NT/NN");
    SG_CHECK(hipSetDevice(ctx->device));
    const int ar = mode == 2 ? K : M, ac = mode == 2 ? M : K, br = mode == 0 ? N : K, bc = mode == 0 ? K : N;
    const size_t lds = sizeof(float) * ((size_t)ar * SG_LD(ac) + (size_t)br * SG_LD(bc) + (size_t)M * SG_LD(N));
    SG_REQUIRE(lds <= (size_t)ctx->lds_bytes, "sg_test_gemm: %zu bytes of LDS needed, %d available", lds, ctx->lds_bytes);
    float *dA, *dB, *dC;
    SG_CHECK(hipMalloc((void**)&dA, sizeof(float) * ar * ac));
    SG_CHECK(hipMalloc((void**)&dB, sizeof(float) * br * bc));
    SG_CHECK(hipMalloc((void**)&dC, sizeof(float) * M * N));
    SG_CHECK(hipMemcpy(dA, A, sizeof(float) * ar * ac, hipMemcpyHostToDevice));
    SG_CHECK(hipMemcpy(dB, B, sizeof(float) * br * bc, hipMemcpyHostToDevice));
    GemmTestArgs a{mode, M, N, K, dA, dB, dC};
    const int MT = mode == 2 ? 1 : M / 16;
    if (MT == 4) hipLaunchKernelGGL(k_gemm_test<4>, dim3(1), dim3(256), lds, ctx->stream, a);
    else if (MT == 2) hipLaunchKernelGGL(k_gemm_test<2>, dim3(1), dim3(256), lds, ctx->stream, a);
    else hipLaunchKernelGGL(k_gemm_test<1>, dim3(1), dim3(256), lds, ctx->stream, a);
    SG_CHECK(hipGetLastError());
    SG_CHECK(hipStreamSynchronize(ctx->stream));
    SG_CHECK(hipMemcpy(C, dC, sizeof(float) * M * N, hipMemcpyDeviceToHost));
    (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC);
    return 0;
}
