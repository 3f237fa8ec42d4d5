// sg_disc_step4.hpp -- one discriminator optimizer step as ONE launch: k_disc_chain4's row blocks and k_disc_wgrad's tile /
// vector blocks side by side, joined by a one-way hand-off inside the launch.
//
// Why (round 4, measured with bodies removed, ms per update -> us per step): two EMPTY dependent launches cost a step 3.8 us,
// the chain blocks add 3.3 and the weight-gradient blocks 2.9 -- of which ~1 us is the first-touch latency of a freshly
// launched block and ~1 us the transfer of 64 KB of operands the other XCDs have just written back.  A step is
//     weights(k) -> chain -> operand stacks -> weight gradient + Adam -> weights(k+1),
// two all-to-all edges.  Putting BOTH inside a persistent launch loses (tools/pstep_probe.py: 11 us per step with no
// compute at all: two polled hand-offs cost more than two launch boundaries).  Putting only the FIRST edge inside wins:
//   * the weight-gradient blocks are resident from the start of the launch: their code is fetched, their tile's parameter
//     and moments and the Adam scalars are in registers, and they sit in a poll loop when the chain ends;
//   * the chain blocks publish their rows of the operand stacks with write-through (`sc1`) stores, drain them, and raise one
//     flag word each (the Adam step number: monotonic, never reset inside an epoch or across replays of the epoch's graph);
//     the other side polls the 12G words with coalesced L1-bypassing loads from one wave and then reads its two operand
//     slabs with `sc1` loads (MI355X: write-through stores + sc1 loads are the cross-XCD-coherent pair);
//   * the dependence is ONE-WAY -- no chain block ever waits for anything -- so whatever the dispatcher does (blocks are
//     dispatched in index order, the chain blocks first) the launch cannot deadlock on residency; the spin is bounded by the
//     wall clock all the same and a time-out is reported through an error word the host reads at its next synchronisation;
//   * the second edge (new weights -> the next step's chain blocks) stays a launch boundary, which is also what keeps every
//     L2 coherent for the plain loads of the next launch.
// Arithmetic, summation order and stores are those of the two-launch step: results are bit-identical (tests/test_gpu_parity.py).
//
// Round 5 tried the hand-off WITHOUT flags (every operand word poisoned one launch ahead and validated by its reader, no drain, no
// flag: `SG_STEP4_POISON`): measured negative (profiles/r05_step4_poison_handoff_negative.txt) and removed from the library in
// round 6 (git show 9df2324:simgan_amd/csrc/sg_disc_step4.hpp has it).
#pragma once
#ifndef SG_ABL
#define SG_ABL 0
#endif

struct Step4Args {          // not preloaded: read by the workgroups that copy the next step's rows and by one vector lane at its end
    PregatherArgs next;
    double* loss_acc;
    long long* dbg;         // SG_STEP4_STAMPS builds only (tools/step4_times.py): wall-clock stamps of a few workgroups
};
#define SG_STEP4_STAMP_SLOTS 16   // long longs per workgroup in Step4Args::dbg (SG_STEP4_STAMPS builds)
#ifndef SG_STEP4_VERIFY
#define SG_STEP4_VERIFY 0     // 1 (debug builds, batch 128): every tile wave re-reads its operands ~2 us after it consumed them and records differences
#endif
#ifndef SG_STEP4_NO_DRAIN
#define SG_STEP4_NO_DRAIN 0   // 1: the hand-off WITHOUT its store drain (to see tests/test_gpu_fullsize.py::..._under_load fail)
#endif
#ifndef SG_STEP4_NOSLEEP
#define SG_STEP4_NOSLEEP 0
#endif
#ifndef SG_STEP4_STAMPS
#define SG_STEP4_STAMPS 0
#endif
#ifndef SG_STEP4_WSTORE
#define SG_STEP4_WSTORE 0     // how the tile blocks store the new weights: 0 streaming (kept), 1 write-through, 2 plain (A/B, round 4)
#endif
#if SG_STEP4_STAMPS
#define SG_STAMP(var) const long long var = wall_clock64()
#else
#define SG_STAMP(var) const long long var = 0
#endif

// (SG_STEP4_FLAG_WORD0 / _MAX_FLAGS / _FLAG_STRIDE / _ERR_WORD / _STATE_BYTES: sg_common.h, next to sg_disc::d_state)
#define SG_STEP4_TIMEOUT_TICKS 300000000ll   // 3 s of the 100 MHz wall clock

__device__ __forceinline__ float sg_ld_sc1(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 16 /* sc1: served past this CU's L1 */));
}
// How the flag form's tile workgroups read the operand stacks once the flags are up (A/B, round 5): 16 = sc1 (past this
// XCD's L2: every tile workgroup pulls its 64 KB through the fabric), 0 = plain (the 13 tiles of an XCD share a 32 KB slab
// through its L2), 1 = sc0.
#ifndef SG_STEP4_OPLOAD
#define SG_STEP4_OPLOAD 16
#endif
__device__ __forceinline__ float sg_ld_op(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, SG_STEP4_OPLOAD));
}
#ifndef SG_STEP4_FIRST_PLAIN
#define SG_STEP4_FIRST_PLAIN 1   // the timed FIRST request of an operand word is a plain load (shared through the XCD's L2); 0: sc1 like the re-requests
#endif
__device__ __forceinline__ float sg_ld_first(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, SG_STEP4_FIRST_PLAIN ? 0 : 16));
}

// One wave waits until the flags of workgroups [lo, hi) all hold `want`; false on time-out.
// Measured and removed, round 4 (north-star, ms per update; 26.19-26.24 with this plain loop):
//  * FOUR polls kept in flight per waiting wave, issued 0.15 us apart and examined oldest first (asm loads + `s_waitcnt
//    vmcnt(6)`: the compiler's wait-count pass drains such a pipe every fourth poll): 27.59 -- 91 waiting workgroups x 96 flag
//    lines x 4 is traffic the chain blocks' write-through stores and the flags themselves queue behind (stores drained 0.3 us
//    later, flags seen 1.28 us after the last one was raised instead of 0.96);
//  * ONE sentinel flag per waiting workgroup polled first, the whole set only once it is up: 26.57 (a serial round trip more
//    at the moment that counts); a further 0.3 us of sleep between the BCE and the mixup flags: 27.6;
//  * no poll before 3.4 us after the workgroup's start (the chain cannot be done sooner): 26.08-26.20, i.e. the early polls
//    do not disturb the chain blocks measurably.
// The hand-off is two fabric round trips -- data acknowledged (0.36-0.40 us), flag visible and sampled (0.96 us) -- and a third
// for the operands (0.68 us with their transfer); tools/step4_times.py prints them.
__device__ __forceinline__ bool sg_step4_wait(const unsigned* flags, int lo, int hi, unsigned want, int lane) {
    const long long deadline = wall_clock64() + SG_STEP4_TIMEOUT_TICKS;
    for (int it = 0;; ++it) {
        bool ok = true;
        for (int j = lo + lane; j < hi; j += 64) ok = ok & (__hip_atomic_load(flags + j * SG_STEP4_FLAG_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == want);
        if (__all(ok)) return true;
        if ((it & 31) == 31 && wall_clock64() > deadline) return false;
#if !SG_STEP4_NOSLEEP
        __builtin_amdgcn_s_sleep(1);
#endif
    }
}

template <int KF, int KH>
__global__ __launch_bounds__(512) void k_disc_step4(float* c_params, float* c_m, float* c_v, float* c_wT, float* c_ops,
                                                   SgOptState* c_st, int c_pack /* G | k1 << 14 (sg_wgrad_pack) */,
                                                   int c_B, Step4Args a) {
    constexpr int Fp = 16 * KF, Hp = 16 * KH, ldF = Fp + 4, ldH = Hp + 4;
    constexpr int NWC = KF > KH ? KF : KH;
    constexpr int W_LDS = 8 * 256, C_LDS = SG_CHAIN4_LDS_FLOATS(KF, KH);
    __shared__ __attribute__((aligned(16))) float sm[(C_LDS > W_LDS ? C_LDS : W_LDS) + 4];
    const int c_G = c_pack & 1023, c_k1 = (int)((unsigned)c_pack >> 14);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int Kt = 64 * c_G, n_chain = 12 * c_G, w_base = (n_chain + 7) & ~7;
    unsigned* flags = reinterpret_cast<unsigned*>(c_st) + SG_STEP4_FLAG_WORD0;
    // scratch = stacks[0] | stacks[1] | partials (disc_update_core); this step's stacks are stacks[(k1 - 1) & 1]
    float* c_part = c_ops + (size_t)(2 - ((c_k1 - 1) & 1)) * ((size_t)Kt * (3 * Hp + ldF + Fp));

    if ((int)blockIdx.x < n_chain) {
        // ------------------------------------------------------------------ the chain: k_disc_chain4's body, published
        if (wave >= NWC) return;   // the launch has 8 waves per workgroup for the tile blocks; the chain uses one per column tile
        int t0 = 0;
        if (tid == 0) t0 = c_st->t0;
        SG_STAMP(ts0);
        Chain4Args ca{c_params, c_wT, c_ops, c_part, nullptr, c_B, c_G, 1.0f / (float)c_B, 10.0f};   // (time stamps: the two-launch path)
#if SG_STEP4_VERIFY
        if (a.dbg) ca.vlog = reinterpret_cast<unsigned*>(reinterpret_cast<float*>(a.dbg + 8 * 512) + (size_t)128 * 8 * 32 * 64);
        long long* vst = a.dbg ? reinterpret_cast<long long*>(ca.vlog + 96 * 8 * 8 * 64) + 4 * blockIdx.x : nullptr;
        if (vst && tid == 0) vst[0] = wall_clock64();
#endif
        sg_chain4_body<KF, KH, (SG_ABL & 2) ? false : true>(ca, sm);
        SG_STAMP(ts1);
#if SG_STEP4_VERIFY
        if (vst && tid == 0) vst[1] = wall_clock64();
#endif
        // every wave drains its write-through stores BEFORE the barrier the flag store sits behind.  The wait has to be spelt
        // out: __syncthreads() is a workgroup-scope fence, for which gfx950 needs no vmcnt wait (the waves of a workgroup share
        // their CU's L1), and the compiler emits none -- the flag then overtakes the data under load (replicas of 8 contexts
        // sharing one GPU diverged in the 6th digit: tests/test_gpu_world.py).
#if !SG_STEP4_NO_DRAIN
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        __syncthreads();
        SG_STAMP(ts2);
        if (tid == 0) __hip_atomic_store(flags + blockIdx.x * SG_STEP4_FLAG_STRIDE, (unsigned)(t0 + c_k1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (SG_STEP4_STAMPS && tid == 0 && a.dbg) { long long* o = a.dbg + SG_STEP4_STAMP_SLOTS * blockIdx.x; o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = wall_clock64(); }
        return;
    }
    const int wb = (int)blockIdx.x - w_base;
    constexpr int th = KH, tf = KF, T2 = th * th, ntv = 8 * (th + tf), NV = (3 * Hp + 4 + 63) / 64;
    static_assert((8 - th) * (th + tf) > NV, "k_disc_step4: a spare slot must exist for the Adam-scalar lane");
    if (wb < 0) return;            // padding up to a multiple of 8: the tile map below counts XCDs from w_base
    if (wb >= ntv) {               // the next step's rows (see k_disc_chain4)
        SG_STAMP(tg0);
        if (a.next.ops) sg_disc_pregather(a.next, wb - ntv);
        if (SG_STEP4_STAMPS && a.dbg) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (SG_STEP4_STAMPS && tid == 0 && a.dbg) { long long* o = a.dbg + SG_STEP4_STAMP_SLOTS * blockIdx.x; o[0] = tg0; o[1] = wall_clock64(); }
        return;
    }
    SgDiscDesc d;
    d.F = 0; d.Hd = 0; d.Hp = Hp; d.Fp = Fp; d.ldF = ldF; d.ldH = ldH;
    d.w1 = 0; d.b1 = Hp * ldF; d.w2 = d.b1 + Hp; d.b2 = d.w2 + Hp * ldH; d.w3 = d.b2 + Hp; d.b3 = d.w3 + Hp; d.total = d.b3 + 16;
    float (*red)[256] = reinterpret_cast<float (*)[256]>(sm);
    int* sh_ok = reinterpret_cast<int*>(sm + W_LDS);
    unsigned* err = reinterpret_cast<unsigned*>(c_st) + SG_STEP4_ERR_WORD;
    // tile -> XCD map of k_disc_wgrad: all tiles of one 16-row weight panel on one XCD (workgroups go to XCDs round-robin)
    const int xcd = wb & 7, slot = wb >> 3;
    const int li = lane & 15, lq = lane >> 4;
    constexpr int nw = 8;
    if (xcd < th) {
        // -------------------------------------------------------------- a 16 x 16 weight tile: k_disc_wgrad's tile body
        const int b = slot < th ? xcd * th + slot : T2 + xcd * tf + (slot - th);
        const bool w2 = b < T2;
        const int t = w2 ? b : b - T2;
        const int tm = w2 ? t / th : t / tf, tn = w2 ? t % th : t % tf;
        const int ldp = w2 ? ldH : ldF;
        const int idx = (w2 ? d.w2 : d.w1) + (tm * 16 + ((tid & 255) >> 4)) * ldp + tn * 16 + (tid & 15);
        // everything that does not depend on the chain is requested before the wait: the tile's parameters and moments, the
        // Adam scalars of both parities, the step base
        float p0 = 0.f, m0 = 0.f, v0 = 0.f;
        float4 sc = float4{0.f, 0.f, 1.f, 1.f};
        int t0 = 0;
        if (tid < 256) {
            p0 = c_params[idx]; m0 = c_m[idx]; v0 = c_v[idx];
            sc = *reinterpret_cast<const float4*>(c_st->step_size2);
        }
        if (tid < 256 || wave == 7) t0 = c_st->t0;
        const SgStacks stk = sg_disc_stacks(c_ops, Kt, Hp, Fp, ldF);
        const float* L = (w2 ? stk.L2 : stk.L1) + (size_t)tm * Kt * 16;
        const float* Rr = (w2 ? stk.R2 : stk.R1t) + (size_t)tn * Kt * 16;
        const int n_chunks = Kt >> 4;
        const __amdgpu_buffer_rsrc_t rL = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(L), 0, Kt * 64, 0x00020000);
        const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Rr), 0, Kt * 64, 0x00020000);
        int img0 = 0, img1 = 0;
        sg_disc_img_pos(d, w2, tm * 16 + ((tid & 255) >> 4), tn * 16 + (tid & 15), img0, img1);
        if (tid == 0) *sh_ok = 1;
        SG_STAMP(ts0);
        __syncthreads();
        // The stacked rows [0, 32G) come from the BCE workgroups (chain blocks [4G, 12G)), which finish ~0.7 us before the mixup
        // workgroups ([0, 4G): seven dependent GEMMs against three): their half of the two slabs is requested and contracted
        // while the mixup half is still being computed.  Waves take 16-row chunks round-robin, two chunks (16 loads per
        // lane) per batch; at batch 128 this is the chunk order of k_disc_wgrad, so the sums are bit-identical to it.
        const unsigned want = (unsigned)(t0 + c_k1);
        // (a time-out is sticky: once the error word is up, the waiting workgroups of every later launch give up at once instead
        // of spinning out their own three seconds -- the update ends in its normal time, with NaN losses and the error reported)
        if (!(SG_ABL & 1) && wave == 7 && (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || !sg_step4_wait(flags, 4 * c_G, n_chain, want, lane)) && lane == 0) *sh_ok = 0;
        __syncthreads();
        if (!*sh_ok) { if (tid == 0) { atomicOr(err, 1u); a.loss_acc[0] = __builtin_nanf(""); } return; }   // the host sees NaN losses, then the error word
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, alt = acc;
        const int half_chunks = n_chunks >> 1;
        // contract chunks [c_lo, c_hi): this wave's share, two chunks per batch; `between` runs once, after the first batch's
        // loads have been issued and before they are consumed (the polling wave looks for the mixup flags there)
        auto contract = [&](int c_lo, int c_hi, auto&& between) {
            bool first = true;
            for (int c0 = c_lo + wave; c0 < c_hi || first; c0 += 2 * nw) {
                float x[2][4], y[2][4];
#pragma unroll
                for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const int c = c0 + cc * nw;
                        const int r = c < c_hi ? 16 * c + 4 * s + lq : Kt;   // past this half: the buffer's range check returns zero
                        x[cc][s] = sg_ld_op(rL, (r * 16 + li) * 4);
                        y[cc][s] = sg_ld_op(rR, (r * 16 + li) * 4);
                    }
                if (first) { between(); first = false; }
#pragma unroll
                for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        if (s & 1) alt = sg_mfma(x[cc][s], y[cc][s], alt);
                        else acc = sg_mfma(x[cc][s], y[cc][s], acc);
                    }
            }
        };
        contract(0, half_chunks, [&]() {
            if (!(SG_ABL & 1) && wave == 7 && !sg_step4_wait(flags, 0, 4 * c_G, want, lane) && lane == 0) *sh_ok = 0;
        });
        __syncthreads();
        SG_STAMP(ts1);
        if (!*sh_ok) { if (tid == 0) { atomicOr(err, 1u); a.loss_acc[0] = __builtin_nanf(""); } return; }   // the host sees NaN losses, then the error word
        contract(half_chunks, n_chunks, []() {});
        acc += alt;
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][(4 * lq + r) * 16 + li] = acc[r];
        SG_STAMP(ts2);
        __syncthreads();
        if (SG_STEP4_STAMPS && tid == 0 && a.dbg) { long long* o = a.dbg + SG_STEP4_STAMP_SLOTS * blockIdx.x; o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = wall_clock64(); }
        if (tid < 256) {
            float g = 0.f;
            {
                float r8[8];
#pragma unroll
                for (int w = 0; w < 8; ++w) r8[w] = red[w][tid];
#pragma unroll
                for (int w = 0; w < 8; ++w) g += r8[w];
            }
            const bool odd = (t0 + c_k1) & 1;
            const float step_size = odd ? sc.y : sc.x, bc2_sqrt = odd ? sc.w : sc.z;
            m0 = m0 + (g - m0) * (float)(1.0 - 0.9);
            v0 = v0 * (float)0.999 + (float)(1.0 - 0.999) * g * g;
            const float denom = sqrtf(v0) / bc2_sqrt + SG_DISC_ADAM_EPS;
            p0 = p0 - step_size * (m0 / denom);
            __builtin_amdgcn_sched_barrier(0);
#if SG_STEP4_WSTORE == 1      // write-through (sc1) stores for what the next launch's chain blocks read
            __hip_atomic_store(c_params + idx, p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_nontemporal_store(m0, c_m + idx);
            __builtin_nontemporal_store(v0, c_v + idx);
            __hip_atomic_store(c_wT + img0, p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(c_wT + img1, p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#elif SG_STEP4_WSTORE == 2    // plain stores: written back by the end-of-kernel release
            c_params[idx] = p0; c_m[idx] = m0; c_v[idx] = v0; c_wT[img0] = p0; c_wT[img1] = p0;
#else
            __builtin_nontemporal_store(p0, c_params + idx);
            __builtin_nontemporal_store(m0, c_m + idx);
            __builtin_nontemporal_store(v0, c_v + idx);
            __builtin_nontemporal_store(p0, c_wT + img0);
            __builtin_nontemporal_store(p0, c_wT + img1);
#endif
        }
        return;
    }
    const int vid = (xcd - th) * (th + tf) + slot;
    if (vid > NV) return;
    if (vid == NV) {               // the next step's Adam scalars (double-precision pow), off everybody's path
        if (tid == 0) sg_opt_prepare(c_st, c_st->t0 + c_k1 + 1);
        return;
    }
    {
        // ------------------------------------------------------------------ biases, w3, loss sums: k_disc_wgrad's vector body
        const int nparts = n_chain, stride = 4 * Hp, NE = 3 * Hp + 4;
        const int i = 64 * vid + lane;
        const float4 sc = *reinterpret_cast<const float4*>(c_st->step_size2);
        const int t0 = c_st->t0;
        const bool odd = (t0 + c_k1) & 1;
        const float step_size = odd ? sc.y : sc.x, bc2_sqrt = odd ? sc.w : sc.z;
        const bool is_param = wave == 0 && i < 3 * Hp + 1;
        const int pidx = i < Hp ? d.b1 + i : i < 2 * Hp ? d.b2 + (i - Hp) : i < 3 * Hp ? d.w3 + (i - 2 * Hp) : d.b3;
        float pv = 0.f, pm = 0.f, pvv = 0.f;
        if (is_param) { pv = c_params[pidx]; pm = c_m[pidx]; pvv = c_v[pidx]; }
        const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(c_part, 0, nparts * stride * 4, 0x00020000);
        SG_STAMP(tv0); (void)tv0;
        if (tid == 0) *sh_ok = 1;
        __syncthreads();
        if (!(SG_ABL & 1) && wave == 7 && (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || !sg_step4_wait(flags, 0, n_chain, (unsigned)(t0 + c_k1), lane)) && lane == 0) *sh_ok = 0;
        __syncthreads();
        if (!*sh_ok) { if (tid == 0) { atomicOr(err, 1u); a.loss_acc[0] = __builtin_nanf(""); } return; }   // the host sees NaN losses, then the error word
        float g = 0.f;
        if (i < NE) {
            for (int s0 = wave; s0 < nparts; s0 += 16 * nw) {
                float tt[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int sidx = s0 + u * nw;
                    tt[u] = sidx < nparts ? sg_ld_sc1(rP, (sidx * stride + i) * 4) : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) g += tt[u];
            }
        }
        red[wave][lane] = g;
        __syncthreads();
        if (wave == 0 && i < NE) {
            g = 0.f;
            {
                float r8[8];
#pragma unroll
                for (int w = 0; w < 8; ++w) r8[w] = red[w][lane];
#pragma unroll
                for (int w = 0; w < 8; ++w) g += r8[w];
            }
            if (i < 3 * Hp + 1) {
                pm = pm + (g - pm) * (float)(1.0 - 0.9);
                pvv = pvv * (float)0.999 + (float)(1.0 - 0.999) * g * g;
                const float denom = sqrtf(pvv) / bc2_sqrt + SG_DISC_ADAM_EPS;
                c_params[pidx] = pv - step_size * (pm / denom);
                c_m[pidx] = pm;
                c_v[pidx] = pvv;
            }
            const int l0 = (3 * Hp) & 63;
            const float el_s = __shfl(g, l0 + 1), pl_s = __shfl(g, l0 + 2), gp_s = __shfl(g, l0 + 3);
            if (i == 3 * Hp) {
                // a2c/algo/gail.py:181-184: loss.item() etc. are float32, accumulated in Python doubles
                const float inv_B = 1.0f / (float)c_B;
                const float el = el_s * inv_B, pl = pl_s * inv_B, gp = 10.0f * (gp_s * inv_B);
                a.loss_acc[0] += (double)(el + pl + gp);
                a.loss_acc[1] += (double)el;
                a.loss_acc[2] += (double)pl;
            }
        }
        if (SG_STEP4_STAMPS && a.dbg) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (tid == 0) a.dbg[SG_STEP4_STAMP_SLOTS * blockIdx.x + 3] = wall_clock64();
        }
    }
}
