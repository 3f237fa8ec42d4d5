// sg_test_pstep.hpp -- probe (tools/pstep_probe.py): what ONE persistent launch would make of a discriminator epoch.
//
// The shipped step is two launches (k_disc_chain4 -> k_disc_wgrad) with a ~1.8 us dependent-launch boundary behind each.
// This skeleton keeps the real dataflow and byte counts of a step inside one launch and replaces the GEMMs by timed spins:
//
//   C blocks (NC = 96: 32 "mixup" blocks of 7 phases, 64 "BCE" blocks of 4): wait for weight version k from the W blocks,
//       take the four weight images (186 KB) into registers in consumption order, spin a phase's cycles per phase, write
//       their rows of the operand stacks (4 bytes per lane per phase, write-through), publish "step k done";
//   W blocks (NW: 91 tile blocks + vector blocks): wait for every C block's "step k done", read their two operand slabs
//       (64 KB, L2-bypassing loads), spin, write 2 x 256 floats of weight version k+1 (write-through), publish.
//
// Every word that crosses carries its step number and is checked by the reader (stale = counted).  Flags are monotonic
// step counts (no resets), one word per producer, polled by one wave with coalesced sc1 loads.  Every spin is bounded by the
// wall clock.  mode bit 0: weight images live in a 2-slot ring and are read with sc1 loads (else: one fresh slot per step,
// plain loads -- an address an L2 has never seen cannot be stale in it); bit 1: W blocks fetch the BCE half of their operands
// as soon as the BCE blocks are done; bit 2: C blocks wait for the W1 tiles (+ vectors) first and for the W2 tiles after W1
// has been requested.
#pragma once

struct PStepArgs {
    float* wring;
    float* stacks;
    unsigned* flagsC;
    unsigned* flagsW;
    long long* stamps;   // [3 blocks][S][4]
    int* err;            // [0] stale words seen by C, [1] stale words seen by W, [2] time-outs
    int NC, NW, S, mode, nslots;
    int cyc_phase, cyc_w;
};

#define PS_WF (2 * (112 * 96 + 112 * 112))   // floats of the four images
#define PS_SF (96 * 448 * 6)                 // floats of one stack version (6 phases x 448 lanes per C block)
#define PS_TIMEOUT_TICKS 150000000ll         // 1.5 s of the 100 MHz wall clock

__device__ __forceinline__ unsigned ps_ld_flag(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ps_st_flag(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ps_spin(long long cycles) {
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(1);
}
// one wave waits until flags[lo..hi) >= want; false on time-out
__device__ __forceinline__ bool ps_wait(const unsigned* flags, int lo, int hi, unsigned want, long long deadline, int lane) {
    for (;;) {
        bool ok = true;
        for (int j = lo + lane; j < hi; j += 64) ok = ok && ps_ld_flag(flags + j) >= want;
        if (__all(ok)) return true;
        if (wall_clock64() > deadline) return false;
        __builtin_amdgcn_s_sleep(2);
    }
}

__global__ __launch_bounds__(512) void k_pstep_probe(PStepArgs a) {
    __shared__ int sh_ok[2];
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long long deadline = wall_clock64() + PS_TIMEOUT_TICKS;
    const bool is_c = b < a.NC;
    const int n_w2 = 49, n_tiles = 91;
    long long* st = nullptr;
    if (tid == 0) {
        if (b == 0) st = a.stamps;
        else if (b == a.NC - 1) st = a.stamps + (size_t)a.S * 4;
        else if (b == a.NC) st = a.stamps + (size_t)2 * a.S * 4;
    }
    if (tid < 2) sh_ok[tid] = 1;
    __syncthreads();
    if (is_c) {
        const bool mixup = b < 32;
        const int nph = mixup ? 7 : 4;
        for (int k = 0; k < a.S; ++k) {
            const float* slot = a.wring + (size_t)(k % a.nslots) * PS_WF;
            const float want = (float)k;
            // ---- wait for weight version k
            if (wave == 7) {
                bool ok;
                if (a.mode & 4) ok = ps_wait(a.flagsW, n_w2, a.NW, (unsigned)k, deadline, lane);
                else ok = ps_wait(a.flagsW, 0, a.NW, (unsigned)k, deadline, lane);
                if (!ok && lane == 0) sh_ok[0] = 0;
            }
            __syncthreads();
            if (!sh_ok[0]) { if (tid == 0) atomicAdd(a.err + 2, 1); return; }
            if (st) st[k * 4 + 0] = wall_clock64();
            int bad = 0;
            float4 w1[6], w2[7], w2t[7], w1t[7];
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(slot), 0, PS_WF * 4, 0x00020000);
            auto ld = [&](int f4) -> float4 {
                const sg_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, f4 * 16, 0, 16 /* sc1 */);
                return float4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
            };
            const int wv = wave < 7 ? wave : 6;
            // W1 image: 7 waves x 6 KiB
            if (wave < 7) {
#pragma unroll
                for (int t = 0; t < 6; ++t) w1[t] = (a.mode & 1) ? ld((wv * 6 + t) * 64 + lane) : reinterpret_cast<const float4*>(slot)[(wv * 6 + t) * 64 + lane];
            }
            if (a.mode & 4) {
                if (wave == 7) { if (!ps_wait(a.flagsW, 0, n_w2, (unsigned)k, deadline, lane) && lane == 0) sh_ok[0] = 0; }
                __syncthreads();
                if (!sh_ok[0]) { if (tid == 0) atomicAdd(a.err + 2, 1); return; }
            }
            const int o2 = 112 * 96 / 4, o3 = o2 + 112 * 112 / 4, o4 = o3 + 112 * 112 / 4;
            if (wave < 7) {
#pragma unroll
                for (int t = 0; t < 7; ++t) w2[t] = (a.mode & 1) ? ld(o2 + (wv * 7 + t) * 64 + lane) : reinterpret_cast<const float4*>(slot)[o2 + (wv * 7 + t) * 64 + lane];
                float s = 0.f;
#pragma unroll
                for (int t = 0; t < 6; ++t) { bad += (w1[t].x != want) + (w1[t].y != want) + (w1[t].z != want) + (w1[t].w != want); s += w1[t].x; }
                if (s == 12345.f) a.err[3] = 1;
            }
            float* stk = a.stacks + (size_t)(k & 3) * PS_SF + (size_t)b * 448 * 6;
            int nst = 0;
            for (int ph = 0; ph < nph; ++ph) {
                if (wave < 7) {
                    if (ph == 1) {
#pragma unroll
                        for (int t = 0; t < 7; ++t) w2t[t] = (a.mode & 1) ? ld(o3 + (wv * 7 + t) * 64 + lane) : reinterpret_cast<const float4*>(slot)[o3 + (wv * 7 + t) * 64 + lane];
#pragma unroll
                        for (int t = 0; t < 7; ++t) bad += (w2[t].x != want) + (w2[t].y != want) + (w2[t].z != want) + (w2[t].w != want);
                    }
                    if (ph == 2) {
#pragma unroll
                        for (int t = 0; t < 7; ++t) w1t[t] = (a.mode & 1) ? ld(o4 + (wv * 6 + (t < 6 ? t : 5)) * 64 + lane) : reinterpret_cast<const float4*>(slot)[o4 + (wv * 6 + (t < 6 ? t : 5)) * 64 + lane];
#pragma unroll
                        for (int t = 0; t < 7; ++t) bad += (w2t[t].x != want) + (w2t[t].y != want) + (w2t[t].z != want) + (w2t[t].w != want);
                    }
                    if (ph == 3) {
#pragma unroll
                        for (int t = 0; t < 7; ++t) bad += (w1t[t].x != want) + (w1t[t].y != want) + (w1t[t].z != want) + (w1t[t].w != want);
                    }
                    ps_spin(a.cyc_phase);
                    if (ph != 3) {   // this phase's rows of the operand stacks: 4 bytes per lane, write-through
                        for (int r = 0; r < (mixup ? 1 : 2) && nst < 6; ++r, ++nst)
                            __hip_atomic_store(stk + nst * 448 + wave * 64 + lane, (float)(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                if (st && ph == 0) st[k * 4 + 1] = wall_clock64();
                __syncthreads();
            }
            if (st) st[k * 4 + 2] = wall_clock64();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) ps_st_flag(a.flagsC + b, (unsigned)(k + 1));
            if (st) st[k * 4 + 3] = wall_clock64();
            if (bad) atomicAdd(a.err + 0, bad);
        }
    } else {
        const int j = b - a.NC;
        for (int k = 0; k < a.S; ++k) {
            const float* stk = a.stacks + (size_t)(k & 3) * PS_SF;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(stk), 0, PS_SF * 4, 0x00020000);
            // the 64 KB this block contracts: 32 dwords per lane, half from the BCE blocks' rows, half from the mixup blocks';
            // tile blocks of one row panel share their left slab (as the XCD map of k_disc_wgrad arranges)
            const int base_bce = 32 * 448 * 6 + ((j * 8192) % (64 * 448 * 6 - 8192));
            const int base_mix = (j * 8192) % (32 * 448 * 6 - 8192);
            float x[32];
            const float want = (float)(k + 1);
            if (wave == 0) { if (!ps_wait(a.flagsC, 32, a.NC, (unsigned)(k + 1), deadline, lane) && lane == 0) sh_ok[0] = 0; }
            if (a.mode & 2) {
                __syncthreads();
                if (!sh_ok[0]) { if (tid == 0) atomicAdd(a.err + 2, 1); return; }
#pragma unroll
                for (int u = 0; u < 16; ++u) x[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (base_bce + u * 512 + tid) * 4, 0, 16));
            }
            if (wave == 0) { if (!ps_wait(a.flagsC, 0, 32, (unsigned)(k + 1), deadline, lane) && lane == 0) sh_ok[0] = 0; }
            __syncthreads();
            if (!sh_ok[0]) { if (tid == 0) atomicAdd(a.err + 2, 1); return; }
            if (st) st[k * 4 + 0] = wall_clock64();
            if (!(a.mode & 2)) {
#pragma unroll
                for (int u = 0; u < 16; ++u) x[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (base_bce + u * 512 + tid) * 4, 0, 16));
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) x[16 + u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (base_mix + u * 512 + tid) * 4, 0, 16));
            int bad = 0;
#pragma unroll
            for (int u = 0; u < 32; ++u) bad += x[u] != want;
            if (st) st[k * 4 + 1] = wall_clock64();
            ps_spin(a.cyc_w);
            __syncthreads();
            if (j < n_tiles) {
                float* slot = a.wring + (size_t)((k + 1) % a.nslots) * PS_WF + (size_t)j * 512;
                __hip_atomic_store(slot + tid, (float)(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (st) st[k * 4 + 2] = wall_clock64();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) ps_st_flag(a.flagsW + j, (unsigned)(k + 1));
            if (st) st[k * 4 + 3] = wall_clock64();
            if (bad) atomicAdd(a.err + 1, bad);
        }
    }
}

__global__ void k_pstep_fill(float* p, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

// stamps: [3][S][4] wall-clock ticks (10 ns) of C block 0 (mixup), C block NC-1 (BCE) and W block 0:
//   C: weights visible | first phase done | last phase done | published;  W: all flags seen | operands in | stored | published
extern "C" int sg_test_pstep_probe(sg_ctx* ctx, int mode, int S, int NW, int cyc_phase, int cyc_w, long long* stamps, int* err4) {
    SG_REQUIRE(ctx && stamps && err4 && S > 0 && S <= 4096 && NW >= 91 && NW <= 150, "sg_test_pstep_probe: bad argument");
    SG_CHECK(hipSetDevice(ctx->device));
    PStepArgs a;
    memset(&a, 0, sizeof a);
    a.NC = 96; a.NW = NW; a.S = S; a.mode = mode; a.cyc_phase = cyc_phase; a.cyc_w = cyc_w;
    a.nslots = (mode & 1) ? 2 : S + 1;
    int* d_err;
    SG_CHECK(hipMalloc((void**)&a.wring, sizeof(float) * (size_t)a.nslots * PS_WF));
    SG_CHECK(hipMalloc((void**)&a.stacks, sizeof(float) * 4 * PS_SF));
    SG_CHECK(hipMalloc((void**)&a.flagsC, 4096));
    SG_CHECK(hipMalloc((void**)&a.stamps, sizeof(long long) * 3 * S * 4));
    SG_CHECK(hipMalloc((void**)&d_err, 64));
    a.flagsW = a.flagsC + 512;
    a.err = d_err;
    for (int rep = 0; rep < 2; ++rep) {   // the second repetition is reported (warm code, warm TLB)
        SG_CHECK(hipMemsetAsync(a.flagsC, 0, 4096, ctx->stream));
        SG_CHECK(hipMemsetAsync(d_err, 0, 64, ctx->stream));
        SG_CHECK(hipMemsetAsync(a.stamps, 0, sizeof(long long) * 3 * S * 4, ctx->stream));
        hipLaunchKernelGGL(k_pstep_fill, dim3(1024), dim3(256), 0, ctx->stream, a.wring, (size_t)a.nslots * PS_WF, -1.0f);
        hipLaunchKernelGGL(k_pstep_fill, dim3(64), dim3(256), 0, ctx->stream, a.wring, (size_t)PS_WF, 0.0f);
        hipLaunchKernelGGL(k_pstep_fill, dim3(256), dim3(256), 0, ctx->stream, a.stacks, (size_t)4 * PS_SF, -1.0f);
        hipLaunchKernelGGL(k_pstep_probe, dim3(a.NC + a.NW), dim3(512), 0, ctx->stream, a);
    }
    SG_CHECK(hipGetLastError());
    SG_CHECK(hipStreamSynchronize(ctx->stream));
    SG_CHECK(hipMemcpy(stamps, a.stamps, sizeof(long long) * 3 * S * 4, hipMemcpyDeviceToHost));
    SG_CHECK(hipMemcpy(err4, d_err, 16, hipMemcpyDeviceToHost));
    (void)hipFree(a.wring); (void)hipFree(a.stacks); (void)hipFree(a.flagsC); (void)hipFree(a.stamps); (void)hipFree(d_err);
    return 0;
}
