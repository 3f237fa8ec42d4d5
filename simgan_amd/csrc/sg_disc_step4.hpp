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
// Round 5: the hand-off WITHOUT flags -- SG_STEP4_POISON (default).  The round-4 hand-off above is three dependent fabric round
// trips (stores acknowledged 0.36-0.40 us, flag seen 0.96 us, operands in 0.68 us).  Now every operand word is its own
// notification: the words of a step's operand stacks (and of the per-workgroup partials) that chain workgroups write are
// POISONED one launch ahead -- filled with the bit pattern SG_POISON_BITS, a NaN no arithmetic on finite or default-NaN inputs
// produces -- by the weight-gradient workgroups of the previous launch while they are idle (the stacks are double-buffered by
// step parity, so nobody reads that buffer then; the epoch's first stacks by k_disc_pregather).  The chain workgroups publish with
// the same 4-byte write-through stores as before, do NOT drain them and raise nothing; a weight-gradient wave requests its
// operands as before, with L1-bypassing loads, and re-requests exactly the 256-byte pieces in which some lane still reads the
// poison pattern, until none does.  Each 4-byte value is validated on its own, so the form relies on no store ordering and on no
// multi-word atomicity; the layout, the bytes moved and the contraction order are unchanged: bit-identical to the two-launch step.
// A value that IS the poison pattern (an input NaN carrying that payload) ends in the time-out, like a lost flag did.
#pragma once
#ifndef SG_ABL
#define SG_ABL 0
#endif

struct Step4Args {          // not preloaded: read by the workgroups that copy the next step's rows and by one vector lane at its end
    PregatherArgs next;
    double* loss_acc;
    long long* dbg;         // SG_STEP4_STAMPS builds only (tools/step4_times.py): wall-clock stamps of a few workgroups
    int poll_a, poll_b;     // poisoned-word hand-off: when a weight-gradient wave requests the BCE half / the mixup half of its operands,
                            // in 10 ns ticks of the wall clock after the workgroup's start (sg_step4_poll_times)
};
#ifndef SG_STEP4_POISON
#define SG_STEP4_POISON 0     // 1: operand words validated against the poison pattern (round 5); 0: drained write-through stores + one flag per chain workgroup (round 4)
#endif
#ifndef SG_STEP4_CHAIN_DRAIN
#define SG_STEP4_CHAIN_DRAIN 0   // A/B: the chain workgroups of the poison form drain their stores before they end
#endif
#ifndef SG_STEP4_NO_CHECK
#define SG_STEP4_NO_CHECK 0   // 1: the poison form WITHOUT its validation (negative control: the one-launch tests must fail)
#endif
#define SG_POISON_BITS 0xFFFFFFFFu
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

// ---- round 5: poisoned operand words (see the head of this file)
__device__ __forceinline__ void sg_st_sc1_b128(__amdgpu_buffer_rsrc_t r, int byte_off, unsigned v) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{v, v, v, v}, r, byte_off, 0, 16 /* sc1: write-through */);
}
// The words of one parity's operand stacks that chain workgroups write -- L2 | R2 | L1 whole (they are adjacent), and rows
// [2nb, 3nb) of every R1t tile (gb; the other R1 / R1t rows are the pre-gathered inputs) -- set to the poison pattern by
// `n_workers` x 512 threads with 16-byte write-through stores.
// ... and the 12G x 4Hp per-workgroup partials of the same parity (`part`).
__device__ __forceinline__ void sg_disc_poison_stacks(float* ops, float* part, int G, int Hp, int Fp, int ldF, int worker, int n_workers) {
    const int Kt = 64 * G, nb = 16 * G;
    const int nA = (3 * Kt * Hp) >> 2, runB = nb * 4, nB = (Fp >> 4) * runB, nC = 12 * G * Hp;   // float4 counts
    const size_t r1t = (size_t)Kt * (3 * Hp + ldF);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(ops, 0, (int)((r1t + (size_t)Kt * Fp) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(part, 0, nC * 16, 0x00020000);
    for (int i = worker * 512 + (int)threadIdx.x; i < nA + nB + nC; i += n_workers * 512) {
        if (i >= nA + nB) { sg_st_sc1_b128(rp, (i - nA - nB) * 16, SG_POISON_BITS); continue; }
        int off4 = i;
        if (i >= nA) { const int j = i - nA, tn = j / runB, q = j - tn * runB; off4 = (int)(r1t >> 2) + (tn * Kt + 2 * nb) * 4 + q; }
        sg_st_sc1_b128(rs, off4 * 16, SG_POISON_BITS);
    }
}
// scratch of an epoch = stacks[0] | stacks[1] | partials[0] | partials[1] | ...: the partials of a step sit behind BOTH stacks, by parity
__device__ __forceinline__ float* sg_step4_part(float* c_ops, int par, int Kt, int Hp, int Fp, int ldF, int G) {
    return c_ops + (size_t)(2 - par) * ((size_t)Kt * (3 * Hp + ldF + Fp)) + (size_t)par * ((size_t)12 * G * 4 * Hp);
}
// true once no lane of the wave holds the poison pattern in v
__device__ __forceinline__ bool sg_is_poison_any(float v) { return __any(__float_as_uint(v) == SG_POISON_BITS) != 0; }

// After an epoch with an odd number of steps the current images are set 1: every epoch starts on set 0 (so do the two-launch forms
// and sg_disc_set_params), so they are copied back once.
__global__ __launch_bounds__(256) void k_disc_img_copy(float4* dst, const float4* src, int n4) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) dst[i] = src[i];
}
// Once per epoch, behind k_disc_pregather: the poison the first step of the epoch expects in ITS stacks and in the partials
// (every later step's stacks are poisoned by the launch before it, the partials by the vector workgroups that consumed them).
__global__ __launch_bounds__(512) void k_disc_poison(float* ops, float* part, int G, int Hp, int Fp, int ldF) {
    sg_disc_poison_stacks(ops, part, G, Hp, Fp, ldF, blockIdx.x, gridDim.x);
}
// A request that reaches memory before the word it asks for costs a whole further round trip (~0.75 us on this chip), one
// that is issued late costs only its lateness: the first request of each half is therefore TIMED -- the chain's schedule is the
// same in every launch (tools/step4_times.py: BCE bodies done 3.1-3.6 us, mixup bodies 3.9-4.3 us after the launch's first
// workgroup starts at the north-star shape) -- so that it arrives just behind the last write-through store of that half.
// Validation makes a request that arrives early harmless, only slower.
__device__ __forceinline__ void sg_sleep_until(long long t) {
    for (;;) {
        const long long left = t - wall_clock64();
        if (left <= 0) return;
        if (left > 24) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(1);
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
#if SG_STEP4_POISON
    float* c_part = sg_step4_part(c_ops, (c_k1 - 1) & 1, Kt, Hp, Fp, ldF, c_G);   // double-buffered by step parity, like the stacks
#else
    float* c_part = c_ops + (size_t)(2 - ((c_k1 - 1) & 1)) * ((size_t)Kt * (3 * Hp + ldF + Fp));
#endif

    if ((int)blockIdx.x < n_chain) {
        // ------------------------------------------------------------------ the chain: k_disc_chain4's body, published
        if (wave >= NWC) return;   // the launch has 8 waves per workgroup for the tile blocks; the chain uses one per column tile
        int t0 = 0;
        if (tid == 0) t0 = c_st->t0;
        SG_STAMP(ts0);
        // Weight images are double-buffered by the step's parity in the poison form: step k reads set (k - 1) & 1 and its tile
        // workgroups write set k & 1.  With word-level notification a tile workgroup is done as soon as ITS operands are in -- a W2
        // tile needs nothing of a BCE workgroup's last phase -- while a chain wave delayed by other work on the chip may still be
        // waiting for its W2^T / W1^T slices: updating the images in place let such a wave read the NEXT step's weights
        // (tools/handoff_stress.py under load: one 256-byte piece of one wave's slice, 2e-4 of the steps).
        Chain4Args ca{c_params, c_wT + (SG_STEP4_POISON ? (size_t)((c_k1 - 1) & 1) * (2 * Hp * (Fp + Hp)) : 0), c_ops, c_part, nullptr, c_B, c_G, 1.0f / (float)c_B, 10.0f};   // (time stamps: the two-launch path)
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
#if SG_STEP4_POISON
        // nothing to drain, nothing to raise: every operand word is its own notification
        (void)t0; (void)flags;
#if SG_STEP4_CHAIN_DRAIN
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#endif
        if (SG_STEP4_STAMPS && tid == 0 && a.dbg) { long long* o = a.dbg + SG_STEP4_STAMP_SLOTS * blockIdx.x; o[0] = ts0; o[1] = ts1; o[2] = ts1; o[3] = ts1; }
        return;
#else
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
#endif
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
#if SG_STEP4_POISON
        { const int io = (c_k1 & 1) * (2 * Hp * (Fp + Hp)); img0 += io; img1 += io; }   // the image set the NEXT step reads
#endif
        if (tid == 0) *sh_ok = 1;
        SG_STAMP(ts0);
#if SG_STEP4_POISON
        // While the chain runs: poison the OTHER parity's stacks for the next step's hand-off (their last reader was the previous
        // launch; the workgroups that copy the next step's rows write different words of them).
        {
            const size_t ops_f = (size_t)Kt * (3 * Hp + ldF + Fp);
            const int par = (c_k1 - 1) & 1;
            float* other = par ? c_ops - ops_f : c_ops + ops_f;
            sg_disc_poison_stacks(other, sg_step4_part(other, 1 - par, Kt, Hp, Fp, ldF, c_G), c_G, Hp, Fp, ldF, xcd * (th + tf) + slot, th * (th + tf));
        }
        const bool alive = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;   // a time-out is sticky (see below)
        __syncthreads();
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, alt = acc;
        const int half_chunks = n_chunks >> 1;
        bool ok = true;
        long long deadline = 0;
        long long dbg_first = 0;   // SG_STEP4_STAMPS: when the first check of the current settle() saw its loads back
        int dbg_rounds = 0;        // ... and how many re-request rounds the wave has gone through
        // Every wave waits for ITS operands: the eight loads of a 16-row chunk have been issued; re-request the ones in which a
        // lane still reads the poison pattern (wave-uniform decisions: one scalar branch per load) until none does.
        auto settle = [&](int c, int c_hi, float (&x)[4], float (&y)[4]) {
            if ((SG_STEP4_NO_CHECK | (SG_ABL & 1)) != 0) return;
            unsigned pend = 0xffu;
            for (int it = 0;; ++it) {
                unsigned np = 0;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if (((pend >> s) & 1u) && sg_is_poison_any(x[s])) np |= 1u << s;
                    if (((pend >> (4 + s)) & 1u) && sg_is_poison_any(y[s])) np |= 16u << s;
                }
                pend = np;
                if (SG_STEP4_STAMPS && it == 0 && !dbg_first) dbg_first = wall_clock64();
                if (!pend) return;
                if (SG_STEP4_STAMPS) ++dbg_rounds;
                if (!alive) { ok = false; return; }
                if ((it & 7) == 7) {
                    const long long now = wall_clock64();
                    if (!deadline) deadline = now + SG_STEP4_TIMEOUT_TICKS;
                    else if (now > deadline) { ok = false; return; }
                }
#if !SG_STEP4_NOSLEEP
                __builtin_amdgcn_s_sleep(2);
#endif
                const int r0 = 16 * c + lq;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if ((pend >> s) & 1u) x[s] = sg_ld_sc1(rL, ((r0 + 4 * s) * 16 + li) * 4);
                    if ((pend >> (4 + s)) & 1u) y[s] = sg_ld_sc1(rR, ((r0 + 4 * s) * 16 + li) * 4);
                }
            }
            (void)c_hi;
        };
        // This wave's share of the stacked rows: 16-row chunks dealt round-robin, two chunks (16 loads per lane) per batch, first
        // the half the BCE workgroups write (stacked rows [0, 32G): chain blocks [4G, 12G), done ~0.7 us before the mixup
        // workgroups: three dependent GEMMs against seven), then the mixup half.  At batch 128 this is the chunk order of
        // k_disc_wgrad, so the sums are bit-identical to it.  The mixup half is REQUESTED (at its own time) before the BCE half
        // is validated and contracted: the two round trips overlap.
        auto issue = [&](int c0, int c_hi, float (&x)[2][4], float (&y)[2][4]) {
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int c = c0 + cc * nw;
                    const int r = c < c_hi ? 16 * c + 4 * s + lq : Kt;   // past this half: the buffer's range check returns zero
                    x[cc][s] = sg_ld_first(rL, (r * 16 + li) * 4);
                    y[cc][s] = sg_ld_first(rR, (r * 16 + li) * 4);
                }
        };
        auto finish = [&](int c0, int c_hi, float (&x)[2][4], float (&y)[2][4]) {
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                if (c0 + cc * nw < c_hi) settle(c0 + cc * nw, c_hi, x[cc], y[cc]);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if (s & 1) alt = sg_mfma(x[cc][s], y[cc][s], alt);
                    else acc = sg_mfma(x[cc][s], y[cc][s], acc);
                }
            }
        };
        float xa[2][4], ya[2][4], xb[2][4], yb[2][4];
        const long long t_wg = wall_clock64();
        if (a.poll_a > 0) sg_sleep_until(t_wg + a.poll_a);
        int ca = wave;
        SG_STAMP(tia);
        issue(ca, half_chunks, xa, ya);
        while (ca + 2 * nw < half_chunks) {    // batches beyond 128 rows
            finish(ca, half_chunks, xa, ya);
            ca += 2 * nw;
            issue(ca, half_chunks, xa, ya);
        }
        if (a.poll_b > 0) sg_sleep_until(t_wg + a.poll_b);
        int cb = half_chunks + wave;
        SG_STAMP(tib);
        issue(cb, n_chunks, xb, yb);
        dbg_first = 0;
        finish(ca, half_chunks, xa, ya);
        SG_STAMP(ts1);
        const long long dbg_fa = dbg_first;
        const int dbg_ra = dbg_rounds;
        dbg_first = 0; dbg_rounds = 0;
        for (;;) {
            finish(cb, n_chunks, xb, yb);
            cb += 2 * nw;
            if (cb >= n_chunks) break;
            issue(cb, n_chunks, xb, yb);
        }
        if (SG_STEP4_STAMPS && tid == 0 && a.dbg) {
            long long* o = a.dbg + SG_STEP4_STAMP_SLOTS * blockIdx.x;
            o[1] = tia; o[2] = dbg_fa; o[3] = ts1; o[4] = tib; o[5] = dbg_first; o[6] = wall_clock64(); o[7] = (long long)(dbg_ra | (dbg_rounds << 16)) ;
        }
        if (!ok && lane == 0) *sh_ok = 0;
#else
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
#endif
        acc += alt;
#if SG_STEP4_VERIFY && SG_STEP4_POISON
        const long long t_contracted = wall_clock64();
#endif
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][(4 * lq + r) * 16 + li] = acc[r];
        SG_STAMP(ts2);
        __syncthreads();
#if SG_STEP4_POISON
        if (SG_STEP4_STAMPS && tid == 0 && a.dbg) { a.dbg[SG_STEP4_STAMP_SLOTS * blockIdx.x] = ts0; a.dbg[SG_STEP4_STAMP_SLOTS * blockIdx.x + 8] = wall_clock64(); (void)ts2; }
#else
        if (SG_STEP4_STAMPS && tid == 0 && a.dbg) { long long* o = a.dbg + SG_STEP4_STAMP_SLOTS * blockIdx.x; o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = wall_clock64(); }
#endif
#if SG_STEP4_POISON
        if (!*sh_ok) { if (tid == 0) { atomicOr(err, 1u); a.loss_acc[0] = __builtin_nanf(""); } return; }   // the host sees NaN losses, then the error word
#endif
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
#if SG_STEP4_STAMPS && SG_STEP4_POISON
        if (a.dbg) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (tid == 0) a.dbg[SG_STEP4_STAMP_SLOTS * blockIdx.x + 9] = wall_clock64();   // this wave's stores acknowledged
        }
#endif
#if SG_STEP4_VERIFY && SG_STEP4_POISON
        if (a.dbg) {
            if (tid == 0) {
                long long* vst = reinterpret_cast<long long*>(reinterpret_cast<unsigned*>(reinterpret_cast<float*>(a.dbg + 8 * 512) + (size_t)128 * 8 * 32 * 64) + 96 * 8 * 8 * 64) + 4 * blockIdx.x;
                vst[0] = t_wg; vst[1] = t_contracted; vst[2] = wall_clock64();
            }
            {   // the consumed operand words of this step, for the host to compare across epochs (tools/handoff_stress.py)
                float* lg = reinterpret_cast<float*>(a.dbg + 8 * 512) + (size_t)((xcd * (th + tf) + slot) * 8 + wave) * 32 * 64 + lane;
#pragma unroll
                for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        lg[(cc * 8 + s * 2 + 0) * 64] = xa[cc][s]; lg[(cc * 8 + s * 2 + 1) * 64] = ya[cc][s];
                        lg[(16 + cc * 8 + s * 2 + 0) * 64] = xb[cc][s]; lg[(16 + cc * 8 + s * 2 + 1) * 64] = yb[cc][s];
                    }
            }
            for (int z = 0; z < 10; ++z) __builtin_amdgcn_s_sleep(64);
            unsigned long long* cnt = reinterpret_cast<unsigned long long*>(a.dbg);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int side = 0; side < 2; ++side) {
                            const int c = (h ? half_chunks : 0) + wave + cc * nw;
                            if (c >= (h ? n_chunks : half_chunks)) continue;
                            const int off = ((16 * c + 4 * s + lq) * 16 + li) * 4;
                            const unsigned now = __float_as_uint(sg_ld_sc1(side ? rR : rL, off));
                            const unsigned was = __float_as_uint(h ? (side ? yb[cc][s] : xb[cc][s]) : (side ? ya[cc][s] : xa[cc][s]));
                            if (now != was) {
                                const unsigned long long n = atomicAdd(cnt, 1ull);
                                if (n < 100) {
                                    long long* o = a.dbg + 8 + 4 * n;
                                    o[0] = ((long long)blockIdx.x << 32) | (wave << 16) | (lane << 8) | (h << 4) | (cc << 3) | (s << 1) | side;
                                    o[1] = ((long long)was << 32) | now;
                                    o[2] = c_k1;
                                    o[3] = off;
                                }
                            }
                        }
        }
#endif
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
#if SG_STEP4_POISON
        // The partials are validated word by word like the operand stacks (and poisoned one launch ahead with them).
        const bool alive = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
        bool ok = true;
        float g = 0.f;
        if (a.poll_b > 0) sg_sleep_until(wall_clock64() + a.poll_b);
        for (int s0 = wave; s0 < nparts; s0 += 16 * nw) {
            float tt[16];
            const bool mine = i < NE;        // lanes past the last element request nothing (an out-of-range offset reads as zero)
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int sidx = s0 + u * nw;
                tt[u] = sidx < nparts ? sg_ld_sc1(rP, mine ? (sidx * stride + i) * 4 : 0x7ffffff0) : 0.f;
            }
            if ((SG_STEP4_NO_CHECK | (SG_ABL & 1)) == 0) {
                unsigned pend = 0xffffu;
                long long deadline = 0;
                for (int it = 0;; ++it) {
                    unsigned np = 0;
#pragma unroll
                    for (int u = 0; u < 16; ++u)
                        if (((pend >> u) & 1u) && sg_is_poison_any(tt[u])) np |= 1u << u;
                    pend = np;
                    if (!pend) break;
                    if (!alive) { ok = false; break; }
                    if ((it & 7) == 7) {
                        const long long now = wall_clock64();
                        if (!deadline) deadline = now + SG_STEP4_TIMEOUT_TICKS;
                        else if (now > deadline) { ok = false; break; }
                    }
                    __builtin_amdgcn_s_sleep(2);
#pragma unroll
                    for (int u = 0; u < 16; ++u)
                        if ((pend >> u) & 1u) tt[u] = sg_ld_sc1(rP, mine ? ((s0 + u * nw) * stride + i) * 4 : 0x7ffffff0);
                }
            }
            if (i < NE) {
#pragma unroll
                for (int u = 0; u < 16; ++u) g += tt[u];
            }
        }
        if (!ok && lane == 0) *sh_ok = 0;
        red[wave][lane] = g;
        SG_STAMP(tv1);
        __syncthreads();
        if (SG_STEP4_STAMPS && tid == 0 && a.dbg) { long long* o = a.dbg + SG_STEP4_STAMP_SLOTS * blockIdx.x; o[0] = tv0; o[1] = tv1; o[2] = wall_clock64(); }
        if (!*sh_ok) { if (tid == 0) { atomicOr(err, 1u); a.loss_acc[0] = __builtin_nanf(""); } return; }   // the host sees NaN losses, then the error word
#else
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
#endif
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
