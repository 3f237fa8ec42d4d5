// launch_floor.hip -- what a dependent kernel launch costs inside a replayed hipGraph, by kernel flavour.
// Build: hipcc -O3 --offload-arch=gfx950 [-mllvm -amdgpu-kernarg-preload-count=16] -o launch_floor launch_floor.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
struct Big { float* p[30]; int n[20]; };
__global__ void k_empty() {}
__global__ void k_args(float* a, float* b, float* c, float* d, float* e, int x, int y, int z, int w) { if (x == -12345) a[0] = 1.f; }
__global__ void k_big(float* a, int x, Big s) { if (x == -12345) a[0] = (float)s.n[3]; }
__global__ __launch_bounds__(512) void k_big_use(float* a, int x, Big s) { a[blockIdx.x * 512 + threadIdx.x] = (float)(s.n[3] + x); }
__global__ __launch_bounds__(512) void k_args_use(float* a, int x, int y) { a[blockIdx.x * 512 + threadIdx.x] = (float)(y + x); }
__global__ __launch_bounds__(512) void k_bdim_use(float* a, int x, int y) { a[blockIdx.x * blockDim.x + threadIdx.x] = (float)(y + x); }
__global__ __launch_bounds__(512) void k_lds(float* a, int x) { __shared__ float sm[2048]; if (x == -12345) { sm[threadIdx.x] = 1.f; __syncthreads(); a[0] = sm[(threadIdx.x + 1) & 511]; } }
__global__ __launch_bounds__(512) void k_store(float* a, int x) { a[blockIdx.x * 512 + threadIdx.x] = (float)x; }
__global__ __launch_bounds__(512) void k_load(const float* a, float* o, int x) { float v = a[blockIdx.x * 512 + threadIdx.x]; if (v == -12345.f) o[0] = v; }

template <typename F>
static double run(const char* name, hipStream_t st, int n, F&& launch2) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    for (int i = 0; i < n; ++i) launch2(i);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    const auto t0 = std::chrono::steady_clock::now();
    const int R = 5;
    for (int r = 0; r < R; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (R * (double)n);
    printf("%-64s %7.3f us per launch\n", name, us);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return us;
}
int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float* buf; CK(hipMalloc(&buf, 1 << 24)); CK(hipMemset(buf, 0, 1 << 24));
    Big big; memset(&big, 0, sizeof big);
    const int n = 2000;
    for (int grid : {1, 96, 120, 256, 1024}) {
        for (int block : {64, 512}) {
            char nm[128];
            snprintf(nm, sizeof nm, "empty, no args                      grid %4d x %3d", grid, block);
            run(nm, st, n, [&](int) { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(block), 0, st); });
        }
    }
    for (int grid : {96, 120}) {
        char nm[128];
        snprintf(nm, sizeof nm, "9 scalar/pointer args (preloadable) grid %4d x 512", grid);
        run(nm, st, n, [&](int i) { hipLaunchKernelGGL(k_args, dim3(grid), dim3(512), 0, st, buf, buf, buf, buf, buf, i, 1, 2, 3); });
        snprintf(nm, sizeof nm, "2 args + 320-byte struct            grid %4d x 512", grid);
        run(nm, st, n, [&](int i) { hipLaunchKernelGGL(k_big, dim3(grid), dim3(512), 0, st, buf, i, big); });
        snprintf(nm, sizeof nm, "store of a struct field (kernarg fetch) grid %4d x 512", grid);
        run(nm, st, n, [&](int i) { hipLaunchKernelGGL(k_big_use, dim3(grid), dim3(512), 0, st, buf, i, big); });
        snprintf(nm, sizeof nm, "store of a preloadable scalar        grid %4d x 512", grid);
        run(nm, st, n, [&](int i) { hipLaunchKernelGGL(k_args_use, dim3(grid), dim3(512), 0, st, buf, i, 3); });
        snprintf(nm, sizeof nm, "same, indexed with blockDim.x        grid %4d x 512", grid);
        run(nm, st, n, [&](int i) { hipLaunchKernelGGL(k_bdim_use, dim3(grid), dim3(512), 0, st, buf, i, 3); });
        snprintf(nm, sizeof nm, "8 KB static LDS                     grid %4d x 512", grid);
        run(nm, st, n, [&](int i) { hipLaunchKernelGGL(k_lds, dim3(grid), dim3(512), 0, st, buf, i); });
        snprintf(nm, sizeof nm, "one 4-byte store per thread         grid %4d x 512", grid);
        run(nm, st, n, [&](int i) { hipLaunchKernelGGL(k_store, dim3(grid), dim3(512), 0, st, buf, i); });
        snprintf(nm, sizeof nm, "one 4-byte load per thread          grid %4d x 512", grid);
        run(nm, st, n, [&](int i) { hipLaunchKernelGGL(k_load, dim3(grid), dim3(512), 0, st, buf, buf + (1 << 20), i); });
        snprintf(nm, sizeof nm, "store kernel then load kernel (pair) grid %4d x 512", grid);
        run(nm, st, n, [&](int i) { if (i & 1) hipLaunchKernelGGL(k_load, dim3(grid), dim3(512), 0, st, buf, buf + (1 << 20), i); else hipLaunchKernelGGL(k_store, dim3(grid), dim3(512), 0, st, buf, i); });
    }
    return 0;
}
