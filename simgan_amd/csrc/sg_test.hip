// sg_test.hip -- libsimgan_hip_test.so: test hooks and probe kernels (engine unit tests, phase-timestamp switches, RNG
// dumps, latency / fetch / counter-calibration probes).  Built from this file alone and loaded only by tests/ and tools/
// (simgan_amd/_lib.py:load_test); the product library exports none of it.  The hooks reach into handles created by the
// product library through the struct definitions of sg_common.h -- both libraries are built from the same headers and
// share the process's HIP runtime.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "sg_common.h"
#include "sg_rng.hpp"
#include "sg_test_api.h"
#include "sg_thin.hpp"

// this library's own error slot (the product's sg_set_error is not exported)
static thread_local char g_test_err[1024] = "";
void sg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_test_err, sizeof g_test_err, fmt, ap);
    va_end(ap);
}
extern "C" const char* sg_test_last_error(void) { return g_test_err; }

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
        // TN: K (= tile rows R) is a compile-time chunk count in the product kernels
        if (a.K == 16) { sg_grad_tn<1>(As, lda, Bs, ldb, a.M, a.N, a.C, a.N, false); sg_grad_tn<1>(As, lda, Bs, ldb, a.M, a.N, a.C, a.N, true); }
        else if (a.K == 32) { sg_grad_tn<2>(As, lda, Bs, ldb, a.M, a.N, a.C, a.N, false); sg_grad_tn<2>(As, lda, Bs, ldb, a.M, a.N, a.C, a.N, true); }
        else { sg_grad_tn<4>(As, lda, Bs, ldb, a.M, a.N, a.C, a.N, false); sg_grad_tn<4>(As, lda, Bs, ldb, a.M, a.N, a.C, a.N, true); }   // exercises the accumulate path: C = 2*A^T B
        return;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < a.M * a.N; i += blockDim.x) a.C[i] = Cs[(i / a.N) * ldc + i % a.N];
}

// mode 3: thin-row engine (sg_thin.hpp).  A[M,K] with M = 4 or 8, B[N,K] with N <= 128; one wave per 16 columns.
template <int K, int RG>
__global__ __launch_bounds__(512) void k_thin_test(GemmTestArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (16 * wave >= a.N) return;
    float4 w[SG4_NW(K)];
    float4 x[RG][SG4_NCH(K)];
    float out[RG];
    sg4_load_w<K>(w, a.B, wave, lane);   // a.B is the image built by k_thin_swizzle
    sg4_load_a<K, RG>(x, a.A, K, lane);
    sg4_mma<K, RG>(x, w, lane, out);
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) a.C[(size_t)(4 * rg + (lane >> 4)) * a.N + 16 * wave + (lane & 15)] = out[rg];
}

__global__ void k_thin_swizzle(const float* M, int N, int K, float* img) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N * K; i += gridDim.x * blockDim.x) img[sg4_img_index(i / K, i % K, K)] = M[i];
}

template <int K>
static void launch_thin_test(sg_ctx* ctx, const GemmTestArgs& a) {
    if (a.M == 8) hipLaunchKernelGGL((k_thin_test<K, 2>), dim3(1), dim3(512), 0, ctx->stream, a);
    else hipLaunchKernelGGL((k_thin_test<K, 1>), dim3(1), dim3(512), 0, ctx->stream, a);
}

// raw v_mfma_f32_4x4x1 with CBSZ=2: d[lane][0..3] for given per-lane a, b and ABID (documents the operand layout)
__global__ void k_mfma_probe(const float* av, const float* bv, int abid, float* d) {
    const int l = threadIdx.x;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (abid == 0) acc = sg4_mfma<0>(av[l], bv[l], acc);
    else if (abid == 1) acc = sg4_mfma<1>(av[l], bv[l], acc);
    else if (abid == 2) acc = sg4_mfma<2>(av[l], bv[l], acc);
    else acc = sg4_mfma<3>(av[l], bv[l], acc);
    for (int r = 0; r < 4; ++r) d[4 * l + r] = acc[r];
}

extern "C" int sg_test_mfma_probe(sg_ctx* ctx, int abid, const float* a, const float* b, float* d) {
    SG_REQUIRE(ctx && a && b && d && abid >= 0 && abid < 4, "sg_test_mfma_probe: bad argument");
    SG_CHECK(hipSetDevice(ctx->device));
    float *da, *db, *dd;
    SG_CHECK(hipMalloc((void**)&da, 256)); SG_CHECK(hipMalloc((void**)&db, 256)); SG_CHECK(hipMalloc((void**)&dd, 1024));
    SG_CHECK(hipMemcpy(da, a, 256, hipMemcpyHostToDevice));
    SG_CHECK(hipMemcpy(db, b, 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_mfma_probe, dim3(1), dim3(64), 0, ctx->stream, da, db, abid, dd);
    SG_CHECK(hipGetLastError());
    SG_CHECK(hipStreamSynchronize(ctx->stream));
    SG_CHECK(hipMemcpy(d, dd, 1024, hipMemcpyDeviceToHost));
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dd);
    return 0;
}

extern "C" int sg_test_gemm(sg_ctx* ctx, int mode, int M, int N, int K, const float* A, const float* B, float* C) {
    SG_REQUIRE(ctx && A && B && C, "sg_test_gemm: NULL argument");
    if (mode == 3) {
        SG_REQUIRE((M == 4 || M == 8) && N % 16 == 0 && N > 0 && N <= 128 && (K == 16 || K == 32 || K == 96 || K == 112),
                   "sg_test_gemm: thin mode needs M in {4,8}, N a multiple of 16 <= 128, K in {16,32,96,112}");
        SG_CHECK(hipSetDevice(ctx->device));
        float *dA, *dB, *dC;
        SG_CHECK(hipMalloc((void**)&dA, sizeof(float) * M * K));
        SG_CHECK(hipMalloc((void**)&dB, sizeof(float) * N * K));
        SG_CHECK(hipMalloc((void**)&dC, sizeof(float) * M * N));
        SG_CHECK(hipMemcpy(dA, A, sizeof(float) * M * K, hipMemcpyHostToDevice));
        SG_CHECK(hipMemcpy(dB, B, sizeof(float) * N * K, hipMemcpyHostToDevice));
        float* dImg;
        SG_CHECK(hipMalloc((void**)&dImg, sizeof(float) * N * K));
        hipLaunchKernelGGL(k_thin_swizzle, dim3(16), dim3(256), 0, ctx->stream, dB, N, K, dImg);
        GemmTestArgs a{mode, M, N, K, dA, dImg, dC};
        if (K == 16) launch_thin_test<16>(ctx, a);
        else if (K == 32) launch_thin_test<32>(ctx, a);
        else if (K == 96) launch_thin_test<96>(ctx, a);
        else launch_thin_test<112>(ctx, a);
        SG_CHECK(hipGetLastError());
        SG_CHECK(hipStreamSynchronize(ctx->stream));
        SG_CHECK(hipMemcpy(C, dC, sizeof(float) * M * N, hipMemcpyDeviceToHost));
        (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC); (void)hipFree(dImg);
        return 0;
    }
    SG_REQUIRE(mode >= 0 && mode <= 2, "sg_test_gemm: mode must be 0 (NT), 1 (NN), 2 (TN) or 3 (thin NT)");
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

// ------------------------------------------------------------------------------------------
// Engine micro-benchmark (tools/gemm_bench.py): one workgroup repeats a layer GEMM on LDS-resident
// operands `iters` times (barrier between repetitions, like a real phase) and reports shader clocks.
struct GemmBenchArgs {
    int mode, K, Np, iters, epi;
    long long* out;
    float* sink;
};

template <int MT>
__global__ __launch_bounds__(512) void k_gemm_bench(GemmBenchArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int R = 16 * MT;
    const int ldk = SG_LD(a.K), ldn = SG_LD(a.Np);
    // NT: in [R][ldk], W [Np][ldk];  NN: in [R][ldk], W [K][ldn];  TN: dY [R][ldk'] ...
    float* In = smem;
    float* Wt = In + R * (ldk > ldn ? ldk : ldn);
    float* Out = Wt + (a.K > a.Np ? a.K : a.Np) * (ldk > ldn ? ldk : ldn);
    const int total = R * (ldk > ldn ? ldk : ldn) + (a.K > a.Np ? a.K : a.Np) * (ldk > ldn ? ldk : ldn) + R * ldn;
    for (int i = threadIdx.x; i < total; i += blockDim.x) smem[i] = 0.001f * (float)((i * 7) % 13);
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < a.iters; ++it) {
        if (a.mode == 0) {
            if (a.epi)
                sg_layer_nt<MT>(In, ldk, Wt, ldk, a.K, a.Np, [&](int r, int c, float v) { Out[r * ldn + c] = sg_tanh(v); });
            else
                sg_layer_nt<MT>(In, ldk, Wt, ldk, a.K, a.Np, [&](int r, int c, float v) { Out[r * ldn + c] = v; });
        } else if (a.mode == 1) {
            sg_layer_nn<MT>(In, ldk, Wt, ldn, a.K, a.Np, [&](int r, int c, float v) { Out[r * ldn + c] = v; });
        } else {
            sg_grad_tn<MT>(In, ldk, Out, ldn, a.K, a.Np, a.sink, a.Np, false);
        }
        __syncthreads();
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) a.out[0] = t1 - t0;
    if (a.sink && threadIdx.x == 0) a.sink[0] = Out[0];
}

extern "C" int sg_test_gemm_bench(sg_ctx* ctx, int mode, int MT, int K, int Np, int threads, int iters, int epi,
                                  long long* cycles) {
    SG_REQUIRE(ctx && cycles, "sg_test_gemm_bench: NULL argument");
    SG_CHECK(hipSetDevice(ctx->device));
    const int R = 16 * MT, ldk = SG_LD(K), ldn = SG_LD(Np), ldm = ldk > ldn ? ldk : ldn, big = K > Np ? K : Np;
    const size_t lds = sizeof(float) * ((size_t)R * ldm + (size_t)big * ldm + (size_t)R * ldn);
    SG_REQUIRE(lds <= (size_t)ctx->lds_bytes, "sg_test_gemm_bench: LDS %zu > %d", lds, ctx->lds_bytes);
    long long* d_out;
    float* d_sink;
    SG_CHECK(hipMalloc((void**)&d_out, 64));
    SG_CHECK(hipMalloc((void**)&d_sink, sizeof(float) * (size_t)(K > 16 ? K : 16) * (Np > 16 ? Np : 16)));
    GemmBenchArgs a{mode, K, Np, iters, epi, d_out, d_sink};
    if (MT == 4) hipLaunchKernelGGL(k_gemm_bench<4>, dim3(1), dim3(threads), lds, ctx->stream, a);
    else if (MT == 2) hipLaunchKernelGGL(k_gemm_bench<2>, dim3(1), dim3(threads), lds, ctx->stream, a);
    else hipLaunchKernelGGL(k_gemm_bench<1>, dim3(1), dim3(threads), lds, ctx->stream, a);
    SG_CHECK(hipGetLastError());
    SG_CHECK(hipStreamSynchronize(ctx->stream));
    SG_CHECK(hipMemcpy(cycles, d_out, sizeof(long long), hipMemcpyDeviceToHost));
    (void)hipFree(d_out); (void)hipFree(d_sink);
    return 0;
}

// ------------------------------------------------------------------------------------------
// Producer/consumer flag latency inside one launch (tools/flag_probe.py): np producer blocks write `words`
// floats each, release-fence and bump a device-scope counter; nc consumer blocks spin on it, acquire, then
// read everything back.  Wall clock (100 MHz) stamps: producer [before fence, after atomic], consumer
// [saw the counter, finished reading].  Spins are bounded so a logic error cannot hang the GPU.
template <int MODE>   // 0: plain stores + __threadfence (L2 write-back); 1: agent-scope (write-through) stores + one counter;
                      // 2: write-through stores + one flag word per producer, consumers poll all flags with coalesced loads
__global__ __launch_bounds__(512) void k_flag_probe(float* data, int words, int np, int nc, unsigned* counter, long long* stamps,
                                                    float* sums) {
    const int b = blockIdx.x, tid = threadIdx.x;
    if (b < np) {
        for (int i = tid; i < words; i += blockDim.x) {
            if (MODE == 0) data[(size_t)b * words + i] = (float)(b + 1);
            else __hip_atomic_store(data + (size_t)b * words + i, (float)(b + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();   // s_waitcnt vmcnt(0) + barrier: every store of the block has been acknowledged
        if (tid == 0) {
            stamps[2 * b] = wall_clock64();
            if (MODE == 0) {
                __threadfence();
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else if (MODE == 1) {
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                __hip_atomic_store(counter + 16 + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            stamps[2 * b + 1] = wall_clock64();
        }
    } else {
        __shared__ int ok;
        if (MODE == 2) {
            if (tid < 64) {   // one wave polls every producer's flag: ceil(np/64) coalesced loads per round
                int spins = 0;
                bool all = false;
                while (!all && spins < (1 << 20)) {
                    bool mine = true;
                    for (int j = tid; j < np; j += 64)
                        mine = mine && __hip_atomic_load(counter + 16 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
                    all = __all(mine);
                    if (!all) __builtin_amdgcn_s_sleep(1);
                    ++spins;
                }
                if (tid == 0) { ok = spins < (1 << 20); stamps[2 * b] = wall_clock64(); }
            }
        } else if (tid == 0) {
            int spins = 0;
            if (MODE == 0) {
                while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)np && spins < (1 << 20)) {
                    __builtin_amdgcn_s_sleep(1);
                    ++spins;
                }
            } else {
                while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)np && spins < (1 << 20)) {
                    __builtin_amdgcn_s_sleep(1);
                    ++spins;
                }
            }
            ok = spins < (1 << 20);
            stamps[2 * b] = wall_clock64();
        }
        __syncthreads();
        float s = 0.f;
        if (ok)
            for (int i = tid; i < np * words; i += blockDim.x)
                s += MODE == 0 ? data[i] : __hip_atomic_load(data + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s = sg_wave_sum(s);
        __shared__ float red[8];
        if ((tid & 63) == 0) red[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) {
            float t = 0.f;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
            sums[b - np] = ok ? t : -1.f;
            stamps[2 * b + 1] = wall_clock64();
        }
    }
}

// MODE 3: the same hand-off confined to ONE XCD.  8 x (np + nc) blocks are launched; a block reads the XCC id the hardware
// reports and leaves unless it runs on XCC 0; the survivors draw tickets (producers first).  Producers write with PLAIN stores
// (the line stays in the XCD's L2), wait for the acknowledgements and set a plain flag; consumers poll and read with
// L1-bypassing (sc1) loads, which the shared L2 serves: no fabric round trip on either leg.
__device__ __forceinline__ unsigned sg_xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 15u;
}
__global__ __launch_bounds__(512) void k_flag_probe_xcd(float* data, int words, int np, int nc, unsigned* counter, long long* stamps,
                                                        float* sums) {
    __shared__ int role;
    const int tid = threadIdx.x;
    if (tid == 0) {
        role = -1;
        if (sg_xcc_id() == 0u) role = (int)__hip_atomic_fetch_add(counter + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const int b = role;
    if (b < 0 || b >= np + nc) return;
    if (b < np) {
        for (int i = tid; i < words; i += blockDim.x) data[(size_t)b * words + i] = (float)(b + 1);
        __syncthreads();   // s_waitcnt vmcnt(0): every store of the block has been acknowledged by the L2
        if (tid == 0) {
            stamps[2 * b] = wall_clock64();
            *reinterpret_cast<volatile unsigned*>(counter + 16 + b) = 1u;
            stamps[2 * b + 1] = wall_clock64();
        }
    } else {
        __shared__ int ok;
        if (tid < 64) {
            int spins = 0;
            bool all = false;
            while (!all && spins < (1 << 20)) {
                bool mine = true;
                for (int j = tid; j < np; j += 64)
                    mine = mine && __hip_atomic_load(counter + 16 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
                all = __all(mine);
                ++spins;
            }
            if (tid == 0) { ok = spins < (1 << 20); stamps[2 * b] = wall_clock64(); }
        }
        __syncthreads();
        float s = 0.f;
        if (ok)
            for (int i = tid; i < np * words; i += blockDim.x)
                s += __hip_atomic_load(data + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s = sg_wave_sum(s);
        __shared__ float red[8];
        if ((tid & 63) == 0) red[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) {
            float t = 0.f;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
            sums[b - np] = ok ? t : -1.f;
            stamps[2 * b + 1] = wall_clock64();
        }
    }
}

extern "C" int sg_test_flag_probe(sg_ctx* ctx, int mode, int np, int nc, int words, long long* stamps, float* sums) {
    SG_REQUIRE(ctx && stamps && sums && np > 0 && nc > 0 && np + nc <= 256 && words > 0 && (mode != 3 || np + nc <= 32), "sg_test_flag_probe: bad argument");
    SG_CHECK(hipSetDevice(ctx->device));
    float *d_data, *d_sums;
    unsigned* d_cnt;
    long long* d_st;
    SG_CHECK(hipMalloc((void**)&d_data, sizeof(float) * (size_t)np * words));
    SG_CHECK(hipMalloc((void**)&d_sums, sizeof(float) * nc));
    SG_CHECK(hipMalloc((void**)&d_cnt, 4096));
    SG_CHECK(hipMalloc((void**)&d_st, sizeof(long long) * 2 * (np + nc)));
    for (int rep = 0; rep < 3; ++rep) {   // last repetition is the one reported (warm code, warm TLB)
        SG_CHECK(hipMemsetAsync(d_cnt, 0, 4096, ctx->stream));
        SG_CHECK(hipMemsetAsync(d_data, 0, sizeof(float) * (size_t)np * words, ctx->stream));
        if (mode == 0) hipLaunchKernelGGL(k_flag_probe<0>, dim3(np + nc), dim3(512), 0, ctx->stream, d_data, words, np, nc, d_cnt, d_st, d_sums);
        else if (mode == 1) hipLaunchKernelGGL(k_flag_probe<1>, dim3(np + nc), dim3(512), 0, ctx->stream, d_data, words, np, nc, d_cnt, d_st, d_sums);
        else if (mode == 2) hipLaunchKernelGGL(k_flag_probe<2>, dim3(np + nc), dim3(512), 0, ctx->stream, d_data, words, np, nc, d_cnt, d_st, d_sums);
        else hipLaunchKernelGGL(k_flag_probe_xcd, dim3(8 * (np + nc)), dim3(512), 0, ctx->stream, d_data, words, np, nc, d_cnt, d_st, d_sums);
    }
    SG_CHECK(hipGetLastError());
    SG_CHECK(hipStreamSynchronize(ctx->stream));
    SG_CHECK(hipMemcpy(stamps, d_st, sizeof(long long) * 2 * (np + nc), hipMemcpyDeviceToHost));
    SG_CHECK(hipMemcpy(sums, d_sums, sizeof(float) * nc, hipMemcpyDeviceToHost));
    (void)hipFree(d_data); (void)hipFree(d_sums); (void)hipFree(d_cnt); (void)hipFree(d_st);
    return 0;
}

#include "sg_test_pstep.hpp"

extern "C" int sg_test_raise_handoff_error(sg_disc* d, sg_ppo* a) {
    const unsigned one = 1u;
    if (d) {
        SG_CHECK(hipSetDevice(d->ctx->device));
        SG_CHECK(hipMemcpyAsync(reinterpret_cast<unsigned*>(d->d_state) + SG_STEP4_ERR_WORD, &one, sizeof one, hipMemcpyHostToDevice, d->ctx->stream));
        SG_CHECK(hipStreamSynchronize(d->ctx->stream));
    }
    if (a) {
        SG_REQUIRE(a->d_pair, "sg_test_raise_handoff_error: this PPO object has no pair-mode state");
        SG_CHECK(hipSetDevice(a->ctx->device));
        SG_CHECK(hipMemcpyAsync(a->d_pair + SG_PAIR_ERR_WORD, &one, sizeof one, hipMemcpyHostToDevice, a->ctx->stream));
        SG_CHECK(hipStreamSynchronize(a->ctx->stream));
    }
    return 0;
}

// Test hook: how the last update of each object was issued -- 0 direct launches, 1 replayed graph, 2 a capture was
// refused once and the object fell back to direct launches (tests/test_gpu_comm.py).
// ---------------------------------------------------------------------------------------------------------------
// Fetch probe (tools/fetch_probe.py): how fast ONE CU takes in a weight image the way k_disc_chain4 does -- every wave
// of a block requests NL 16-byte-per-lane loads (1 KiB per instruction) of a buffer another kernel has just written,
// all blocks the same addresses (mode bit 0 clear) or each block its own copy (bit 0 set); bit 1: 4-byte loads of the
// same bytes (4x the instructions); bit 2: only the lower 32 lanes of every wave load.  A second pass inside the same
// launch re-reads the data (now in this XCD's L2); bit 3: the writer uses nontemporal stores.
// out[block] = {cycles first pass, cycles second pass, cycles until the first load of the first pass has returned}.
__global__ void k_fetch_fill(float* p, size_t n, int nt) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (nt) __builtin_nontemporal_store(1.0f, p + i);
        else p[i] = 1.0f;
    }
}
template <int NL>
__global__ __launch_bounds__(512) void k_fetch_probe(const float* src, int mode, size_t block_stride_f, long long* out, float* sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float* base = src + ((mode & 1) ? (size_t)blockIdx.x * block_stride_f : 0) + (size_t)wave * NL * 256;
    float acc = 0.f;
    long long cyc[2], first = 0;
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
        const long long t0 = clock64();
        if (!(mode & 4) || lane < 32) {
            if (mode & 2) {
                float v[NL][4];
#pragma unroll
                for (int t = 0; t < NL; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[t][j] = __builtin_nontemporal_load(base + t * 256 + j * 64 + lane) * 0.f + base[t * 256 + j * 64 + lane];
#pragma unroll
                for (int t = 0; t < NL; ++t) acc += (v[t][0] + v[t][1]) + (v[t][2] + v[t][3]);
            } else {
                float4 v[NL];
                // one wave-uniform descriptor, immediate offsets: the loads issue back to back (sg_stage_issue's pattern)
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 0x7fffffff, 0x00020000);
                const int boff = (int)((base - src) * 4) + lane * 16;
#pragma unroll
                for (int t = 0; t < NL; ++t) v[t] = sg_buffer_load4(rsrc, boff + t * 1024);
                acc += v[0].x;   // the first load's round trip
                __builtin_amdgcn_sched_barrier(0);
                if (pass == 0 && threadIdx.x == 0) first = clock64() - t0;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < NL; ++t) acc += (v[t].x + v[t].y) + (v[t].z + v[t].w);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cyc[pass] = clock64() - t0;
    }
    if (threadIdx.x == 0) { out[3 * blockIdx.x] = cyc[0]; out[3 * blockIdx.x + 1] = cyc[1]; out[3 * blockIdx.x + 2] = first; }
    if (acc == 12345.678f) sink[0] = acc;
}

extern "C" int sg_test_fetch_probe(sg_ctx* ctx, int n_blocks, int waves, int mode, long long* out) {
    SG_REQUIRE(ctx && out && n_blocks > 0 && n_blocks <= 1024 && waves >= 1 && waves <= 8, "sg_test_fetch_probe: bad argument");
    SG_CHECK(hipSetDevice(ctx->device));
    constexpr int NL = 27;
    const size_t per_block_f = (size_t)8 * NL * 256;
    const size_t n = per_block_f * ((mode & 1) ? (size_t)n_blocks : 1);
    float *d_src, *d_sink;
    long long* d_out;
    SG_CHECK(hipMalloc((void**)&d_src, sizeof(float) * n));
    SG_CHECK(hipMalloc((void**)&d_sink, 64));
    SG_CHECK(hipMalloc((void**)&d_out, sizeof(long long) * 3 * n_blocks));
    for (int rep = 0; rep < 6; ++rep) {   // last repetition is the one reported (warm code, warm TLB); the data is rewritten before each
        hipLaunchKernelGGL(k_fetch_fill, dim3(256), dim3(256), 0, ctx->stream, d_src, n, (mode >> 3) & 1);
        hipLaunchKernelGGL(k_fetch_probe<NL>, dim3(n_blocks), dim3(64 * waves), 0, ctx->stream, d_src, mode, per_block_f, d_out, d_sink);
    }
    SG_CHECK(hipGetLastError());
    SG_CHECK(hipStreamSynchronize(ctx->stream));
    SG_CHECK(hipMemcpy(out, d_out, sizeof(long long) * 3 * n_blocks, hipMemcpyDeviceToHost));
    (void)hipFree(d_src); (void)hipFree(d_sink); (void)hipFree(d_out);
    return 0;
}

extern "C" int sg_test_graph_state(sg_ppo* a, sg_disc* d, int out[2]) {
    SG_REQUIRE(out, "sg_test_graph_state: NULL argument");
    out[0] = a ? (a->graph_refused ? 2 : (a->steps_graph ? 1 : 0)) : -1;
    out[1] = d ? (d->graph_refused ? 2 : (d->epoch_graph ? 1 : 0)) : -1;
    return 0;
}

// ------------------------------------------------------------------------------------------------------------
// Test hook: calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on this chip in THIS library's access patterns
// (MI355X_MICROARCH.md, section HBM: FETCH_SIZE reports half the bytes of a 16-byte-per-lane coalesced stream on gfx950;
// "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern").
// Each kernel below moves a KNOWN number of bytes of a buffer larger than the 256 MiB Infinity Cache, once, fully
// coalesced, at one access width; tools/pmc_calibrate.py runs them under `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE`
// and writes counter bytes / true bytes per width to profiles/, which tools/make_traffic.py applies per kernel.
template <int W>   // bytes per lane per load instruction: 4, 8 or 16
__global__ __launch_bounds__(256) void k_calib_read(const float* src, size_t n_floats, float* sink) {
    constexpr int V = W / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x * V;
    float acc = 0.f;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * V; i + V <= n_floats; i += stride) {
        if (V == 4) { const float4 v = *reinterpret_cast<const float4*>(src + i); acc += v.x + v.y + v.z + v.w; }
        else if (V == 2) { const float2 v = *reinterpret_cast<const float2*>(src + i); acc += v.x + v.y; }
        else acc += src[i];
    }
    if (acc == 12345.678f) sink[0] = acc;   // keeps the loads alive
}
template <int W>
__global__ __launch_bounds__(256) void k_calib_write(float* dst, size_t n_floats, float val) {
    constexpr int V = W / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x * V;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * V; i + V <= n_floats; i += stride) {
        if (V == 4) *reinterpret_cast<float4*>(dst + i) = float4{val, val, val, val};
        else if (V == 2) *reinterpret_cast<float2*>(dst + i) = float2{val, val};
        else dst[i] = val;
    }
}
// The weight-gradient kernel's pattern: range-checked 4-byte buffer loads, a wave's 64 lanes covering 4 rows of 64 bytes
// that sit 64 bytes apart in a [K][16] slab, i.e. one contiguous 256 bytes per load instruction.
__global__ __launch_bounds__(256) void k_calib_read_buf4(const float* src, size_t n_floats, float* sink) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (int)(n_floats * 4 > 0x7fffffffull ? 0x7fffffff : n_floats * 4), 0x00020000);
    float acc = 0.f;
    const size_t total = n_floats * 4 > 0x7fffffffull ? (size_t)0x7fffffff / 4 : n_floats;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        acc += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(i * 4), 0, 0));
    if (acc == 12345.678f) sink[0] = acc;
}

extern "C" int sg_test_pmc_calibrate(sg_ctx* ctx, int64_t mbytes) {
    SG_REQUIRE(ctx && mbytes >= 16 && mbytes <= 8192, "sg_test_pmc_calibrate: bad argument");
    SG_CHECK(hipSetDevice(ctx->device));
    const size_t n = (size_t)mbytes * (1 << 20) / 4;
    float *a = nullptr, *b = nullptr, *sink = nullptr;
    SG_CHECK(hipMalloc((void**)&a, n * 4));
    SG_CHECK(hipMalloc((void**)&b, n * 4));
    SG_CHECK(hipMalloc((void**)&sink, 64));
    SG_CHECK(hipMemsetAsync(a, 0, n * 4, ctx->stream));
    SG_CHECK(hipMemsetAsync(b, 0, n * 4, ctx->stream));   // evicts `a` from the Infinity Cache (buffers exceed it)
    const dim3 grid(4096), block(256);
    hipLaunchKernelGGL((k_calib_read<16>), grid, block, 0, ctx->stream, a, n, sink);
    hipLaunchKernelGGL((k_calib_write<16>), grid, block, 0, ctx->stream, b, n, 1.0f);
    hipLaunchKernelGGL((k_calib_read<8>), grid, block, 0, ctx->stream, a, n, sink);
    hipLaunchKernelGGL((k_calib_write<8>), grid, block, 0, ctx->stream, b, n, 2.0f);
    hipLaunchKernelGGL((k_calib_read<4>), grid, block, 0, ctx->stream, a, n, sink);
    hipLaunchKernelGGL((k_calib_write<4>), grid, block, 0, ctx->stream, b, n, 3.0f);
    hipLaunchKernelGGL(k_calib_read_buf4, grid, block, 0, ctx->stream, a, n, sink);
    SG_CHECK(hipGetLastError());
    SG_CHECK(hipStreamSynchronize(ctx->stream));
    (void)hipFree(a); (void)hipFree(b); (void)hipFree(sink);
    return 0;
}

// Test hook: enable/read per-phase shader-clock timestamps of k_disc_chain (tools/phase_times.py).
extern "C" int sg_test_disc_phase_times(sg_disc* d, int enable, long long* out, int n_blocks) {
    SG_REQUIRE(d, "sg_test_disc_phase_times: NULL argument");
    SG_CHECK(hipStreamSynchronize(d->ctx->stream));
    if (enable && !d->d_dbg) {
        SG_CHECK(hipMalloc((void**)&d->d_dbg, sizeof(long long) * 32 * 512));
        SG_CHECK(hipMemsetAsync(d->d_dbg, 0, sizeof(long long) * 32 * 512, d->ctx->stream)); SG_CHECK(hipStreamSynchronize(d->ctx->stream));
    }
    if (out && d->d_dbg) {
        SG_REQUIRE(n_blocks <= 512, "sg_test_disc_phase_times: at most 512 blocks");
        SG_COPY_SYNC(d->ctx, out, d->d_dbg, sizeof(long long) * 32 * n_blocks, hipMemcpyDeviceToHost);   // n_blocks = 512: + k_disc_wgrad stamps
    }
    if (!enable && d->d_dbg) { SG_CHECK(hipFree(d->d_dbg)); d->d_dbg = nullptr; }
    return 0;
}

// Test hook: per-phase shader-clock timestamps of k_ppo_fwd, row groups [0, n_blocks) (tools/ppo_phase_times.py).
// k_disc_step4's wall-clock stamps (library built with -DSG_STEP4_STAMPS=1): 16 per workgroup, of the epoch's last-but-one step (the last has no next step's rows to copy)
extern "C" int sg_test_disc_step4_times(sg_disc* d, int enable, long long* out, int n_blocks) {
    SG_REQUIRE(d && n_blocks >= 0 && n_blocks <= 512, "sg_test_disc_step4_times: bad argument");
    SG_CHECK(hipSetDevice(d->ctx->device));
    SG_CHECK(hipStreamSynchronize(d->ctx->stream));
    if (enable && !d->d_dbg_step4) {
        SG_CHECK(hipMalloc((void**)&d->d_dbg_step4, sizeof(long long) * 16 * 512));
        SG_CHECK(hipMemset(d->d_dbg_step4, 0, sizeof(long long) * 16 * 512));
    }
    if (out && d->d_dbg_step4) SG_CHECK(hipMemcpy(out, d->d_dbg_step4, sizeof(long long) * 16 * n_blocks, hipMemcpyDeviceToHost));
    if (!enable && d->d_dbg_step4) { SG_CHECK(hipFree(d->d_dbg_step4)); d->d_dbg_step4 = nullptr; }
    return 0;
}

extern "C" int sg_test_disc_gathers(sg_disc* d, long long* out) {
    SG_REQUIRE(d && out, "sg_test_disc_gathers: NULL argument");
    *out = (long long)d->n_gathers;
    return 0;
}

extern "C" int sg_test_ppo_phase_times(sg_ppo* a, int enable, long long* out, int n_blocks) {
    SG_REQUIRE(a, "sg_test_ppo_phase_times: NULL argument");
    SG_CHECK(hipStreamSynchronize(a->ctx->stream));
    if (enable && !a->d_dbg) {
        SG_CHECK(hipMalloc((void**)&a->d_dbg, sizeof(long long) * 16 * 1024));
        SG_CHECK(hipMemsetAsync(a->d_dbg, 0, sizeof(long long) * 16 * 1024, a->ctx->stream)); SG_CHECK(hipStreamSynchronize(a->ctx->stream));
    }
    if (out && a->d_dbg) {
        SG_REQUIRE(n_blocks <= 1024, "sg_test_ppo_phase_times: at most 1024 blocks");
        SG_COPY_SYNC(a->ctx, out, a->d_dbg, sizeof(long long) * 16 * n_blocks, hipMemcpyDeviceToHost);
    }
    if (!enable && a->d_dbg) { SG_CHECK(hipFree(a->d_dbg)); a->d_dbg = nullptr; }
    return 0;
}

// The permutation exactly as the product draws it (sg_ppo.hip:sg_fill_perm: key and round count come from sg_rng.hpp).
__global__ void k_test_fill_perm(int64_t* perm, int64_t n, int half_bits, uint64_t key) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) perm[i] = sg_perm_at(i, n, half_bits, key);
}

// Test hook: the library's counter-based generators as host arrays (tests/test_gpu_fullsize.py).
// kind 0: random permutation of [0, n) -> int64 out;  1: uniform [0,1) -> float out;  2: normal -> float out.
__global__ void k_test_rng(float* out, int64_t n, uint64_t seed, int kind) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = kind == 1 ? sg_uniform(seed, 7, (uint64_t)i) : sg_normal(seed, 7, (uint64_t)i);
}

extern "C" int sg_test_rng(sg_ctx* ctx, int kind, int64_t n, uint64_t seed, void* out) {
    SG_REQUIRE(ctx && out && n > 0 && kind >= 0 && kind <= 2, "sg_test_rng: bad argument");
    SG_CHECK(hipSetDevice(ctx->device));
    void* dev = nullptr;
    SG_CHECK(hipMalloc(&dev, (size_t)n * 8));
    if (kind == 0) {
        hipLaunchKernelGGL(k_test_fill_perm, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (int64_t*)dev, n,
                           sg_perm_half_bits((uint64_t)n), sg_key(seed, 0x5045524Dull, 3));
        SG_CHECK(hipStreamSynchronize(ctx->stream));
        SG_COPY_SYNC(ctx, out, dev, (size_t)n * 8, hipMemcpyDeviceToHost);
    } else {
        hipLaunchKernelGGL(k_test_rng, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (float*)dev, n, seed, kind);
        SG_CHECK(hipStreamSynchronize(ctx->stream));
        SG_COPY_SYNC(ctx, out, dev, (size_t)n * 4, hipMemcpyDeviceToHost);
    }
    (void)hipFree(dev);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Tear probe (tools/tear_probe.py): can ONE lane's write-through store of W bytes be seen half-written by a lane of another
// XCD?  Writer workgroups (XCDs 0-3) store {t, t, ...} with t = 1, 2, ... into their lanes' slots as fast as they can; reader
// workgroups (XCDs 4-7) load the same slots with L1-bypassing loads and count the words whose dwords disagree.
// mode 0: 16-byte word, 16-byte aligned;  1: 16-byte word at an 8-byte offset (straddles a 16-byte boundary);
// mode 2: 8-byte word, 8-byte aligned;    3: 8-byte word at a 4-byte offset;
// positive controls: 4: 8-byte word across a 64-byte boundary (offset 60); 5: 16-byte word across a 64-byte boundary (offset 56);
//                    6: 16-byte word across a 128-byte line (offset 120); 7: 8-byte word across a 128-byte line (offset 124).
// out = {torn words seen, words read, largest t seen, reader lanes that saw at least two different t}.
typedef unsigned int sg_tu4 __attribute__((ext_vector_type(4)));
typedef unsigned int sg_tu2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_tear_probe(unsigned* words, int n_bytes, int iters, int mode, unsigned long long* out, unsigned* done) {
    const int xcd = blockIdx.x & 7, tid = threadIdx.x;
    const bool writer = xcd < 4;
    const int pair = (int)(blockIdx.x >> 3) * 4 + (xcd & 3);
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(words, 0, n_bytes, 0x00020000);
    const int offs[8] = {0, 8, 0, 4, 60, 56, 120, 124};
    const int off = (pair * 256 + tid) * 256 + offs[mode & 7];   // one 256-byte slot per lane
    const bool wide = mode < 2 || mode == 5 || mode == 6;
    if (writer) {
        for (int t = 1; t <= iters; ++t) {
            const unsigned u = (unsigned)t;
            if (wide) __builtin_amdgcn_raw_buffer_store_b128(sg_tu4{u, u, u, u}, r, off, 0, 16);
            else __builtin_amdgcn_raw_buffer_store_b64(sg_tu2{u, u}, r, off, 0, 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) atomicAdd(done, 1u);
        return;
    }
    unsigned long long torn = 0, reads = 0;
    unsigned tmax = 0, tfirst = 0, changed = 0;
    const unsigned n_writers = gridDim.x / 2;
    for (int it = 0;; ++it) {
        unsigned a, b, c, d;
        if (wide) { const sg_tu4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16); a = v.x; b = v.y; c = v.z; d = v.w; }
        else { const sg_tu2 v = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 16); a = v.x; b = v.y; c = a; d = b; }
        reads += 1;
        torn += (a != b) | (a != c) | (a != d);
        const unsigned m = a > b ? a : b;
        tmax = m > tmax ? m : tmax;
        if (tfirst == 0 && a != 0) tfirst = a;
        changed |= (tfirst != 0 && a != tfirst);
        if ((it & 63) == 63 && (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n_writers || it > 64 * iters)) break;
    }
    atomicAdd(out + 0, torn);
    atomicAdd(out + 1, reads);
    atomicMax(out + 2, (unsigned long long)tmax);
    atomicAdd(out + 3, (unsigned long long)changed);
}

extern "C" int sg_test_tear_probe(sg_ctx* ctx, int mode, int pairs4, int iters, long long* out4) {
    SG_REQUIRE(ctx && out4 && mode >= 0 && mode <= 7 && pairs4 > 0 && pairs4 <= 16 && iters > 0, "sg_test_tear_probe: bad argument");
    SG_CHECK(hipSetDevice(ctx->device));
    const int n_bytes = pairs4 * 4 * 256 * 256 + 256;
    unsigned *d_words = nullptr, *d_done = nullptr;
    unsigned long long* d_out = nullptr;
    SG_CHECK(hipMalloc((void**)&d_words, (size_t)n_bytes));
    SG_CHECK(hipMalloc((void**)&d_done, 256));
    SG_CHECK(hipMalloc((void**)&d_out, 64));
    SG_CHECK(hipMemsetAsync(d_words, 0, (size_t)n_bytes, ctx->stream));
    SG_CHECK(hipMemsetAsync(d_done, 0, 256, ctx->stream));
    SG_CHECK(hipMemsetAsync(d_out, 0, 64, ctx->stream));
    hipLaunchKernelGGL(k_tear_probe, dim3(8 * pairs4), dim3(256), 0, ctx->stream, d_words, n_bytes, iters, mode, d_out, d_done);
    SG_CHECK(hipGetLastError());
    SG_CHECK(hipStreamSynchronize(ctx->stream));
    SG_CHECK(hipMemcpy(out4, d_out, 32, hipMemcpyDeviceToHost));
    (void)hipFree(d_words); (void)hipFree(d_done); (void)hipFree(d_out);
    return 0;
}
