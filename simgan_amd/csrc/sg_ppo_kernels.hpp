// sg_ppo_kernels.hpp -- device kernels of one PPO optimizer step (a2c/algo/ppo.py:82-145).
//
//   k_ppo_epoch_gather    once per epoch: the permuted copy of the rollout the epoch's minibatches
//                         read (obs rows padded to the LDS stride, actions, per-row scalars), so the
//                         per-step kernels load contiguous 16-byte-aligned rows instead of chasing
//                         permutation indices (a2c/storage.py:159-185 gathers per minibatch).
//   k_ppo_fwd<MT,KO,KH>   grid (row groups, trunks): one trunk's parameter block in LDS, forward on
//                         R = 16*MT rows; h1, h2 and the head outputs go to global row stacks.
//   k_ppo_bwd<MT,KO,KH>   grid (row groups, trunks): reload the rows' activations, evaluate the loss
//                         (needs the head outputs of every actor trunk: the log-prob sums over all
//                         action dims), back-propagate, form the row group's partial gradient of the
//                         trunk (three TN GEMMs on the LDS tiles, bias sums from the epilogues) and
//                         write it to the row group's slab.
//   k_ppo_reduce          slab sum per parameter (64 parameters per block, waves split the slabs), per-block
//                         sum of squares.
//   k_ppo_adam            clip coefficient max_norm/(||g||+1e-6) (<= 1), then Adam.
// Splitting forward / backward per trunk keeps every workgroup's LDS to one trunk's parameters
// (the split policy's trunks are 69-98 KB each) and lets all trunks of all row groups run
// concurrently: 2 x 128 workgroups for Policy, 3 x 128 for SplitPolicy at 4096-row minibatches.
#pragma once
#include "sg_common.h"
#include "sg_thin.hpp"

#define HALF_LOG_2PI 0.91893853320467274178f

struct PpoArgs {
    SgPolicyDesc d;
    const float* params;
    // this minibatch's rows in the epoch's permuted copy
    const float* X;       // [mb..][ldO]
    const float* ACT;     // [mb..][A]
    const float* SC;      // [4][sc_stride]: old_logp | adv | value_pred | return
    int sc_stride;
    int mb, mbp;          // local minibatch rows / rounded up to the row tile
    float inv_B;          // 1 / global minibatch rows
    float clip, vcoef, ecoef;
    int use_clipped;
    float* H1[3];         // [mbp][ldH] per trunk
    float* H2[3];
    float* OUT[3];        // [mbp][ldP] head outputs
    float* slabs;         // [G][total+8] per-row-group partial gradients (+3 loss sums)
    int slab_stride;
    int ldP;              // stride of the OUT stacks (max over trunks)
    int wbuf_floats;
    long long* dbg;       // optional phase timestamps [block][16] (test hook), NULL in production
    SgOptState* st;       // optimizer scalars (read by k_ppo_adam; prepared one step ahead by the previous k_ppo_adam)
    int k1, G;            // 1-based step index within the update; row groups
    unsigned* pair;       // k_ppo_pair: the error word an actor workgroup raises when its partner never shows up
};

// k_ppo_pair (SplitPolicy, one launch per step): the two actor workgroups of a row group exchange their head outputs
#define SG_PAIR_BYTES 64
#define SG_PAIR_TIMEOUT_TICKS 300000000ll   // 3 s of the 100 MHz wall clock



// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for vmcnt(0): in the backward phases that is
// the acknowledgement of the gradient-slab stores the phase has just issued -- a memory round trip per phase for data no
// later phase of the kernel reads.
#define SG_LDS_SYNC() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

#define SG_PPO_STAMP(n) do { if (a.dbg && threadIdx.x == 0 && blockIdx.y == 0) a.dbg[blockIdx.x * 16 + (n)] = clock64(); } while (0)
#define SG_PPO_WALL(n) do { if (a.dbg && threadIdx.x == 0 && blockIdx.y < 2) a.dbg[(blockIdx.x + 512 * blockIdx.y) * 16 + (n)] = wall_clock64(); } while (0)

struct EpochGatherArgs {
    const float *obs, *actions, *old_logp, *adv, *vpred, *ret;
    const int64_t* perm;
    int64_t TN;
    int O, Op, ldO, A, sc_stride;
    float *X, *ACT, *SC;
};

__global__ __launch_bounds__(256) void k_ppo_epoch_gather(EpochGatherArgs a) {
    __shared__ int IDX[64];
    const int tid = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * 64;
    if (tid < 64) {
        const int64_t rr = row0 + tid;
        const int idx = rr < a.TN ? (int)a.perm[rr] : -1;
        IDX[tid] = idx;
        if (idx >= 0) {
            a.SC[0 * (size_t)a.sc_stride + rr] = a.old_logp[idx];
            a.SC[1 * (size_t)a.sc_stride + rr] = a.adv[idx];
            a.SC[2 * (size_t)a.sc_stride + rr] = a.vpred[idx];
            a.SC[3 * (size_t)a.sc_stride + rr] = a.ret[idx];
        }
    }
    __syncthreads();
    for (int base = tid; base < 64 * a.Op; base += 8 * blockDim.x) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * blockDim.x;
            const int r = i / a.Op, c = i - r * a.Op;
            v[u] = (i < 64 * a.Op && IDX[r] >= 0 && c < a.O) ? a.obs[(size_t)IDX[r] * a.O + c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * blockDim.x;
            if (i < 64 * a.Op && IDX[i / a.Op] >= 0) a.X[(size_t)(row0 + i / a.Op) * a.ldO + (i % a.Op)] = v[u];
        }
    }
    for (int i = tid; i < 64 * a.A; i += blockDim.x) {
        const int r = i / a.A, c = i - r * a.A;
        if (IDX[r] >= 0) a.ACT[(size_t)(row0 + r) * a.A + c] = a.actions[(size_t)IDX[r] * a.A + c];
    }
}

// --------------------------------------------------------------------------------- forward
// GW (sg_gemm.hpp): the trunk's parameter block is read from global memory by the layer GEMMs instead of being staged
// into LDS -- the general-shape instances for trunks larger than a CU's LDS (run-time extents only: KO = KH = 0).
template <int MT, int KO, int KH, bool GW = false>
__device__ __forceinline__ void sg_ppo_fwd_body(const PpoArgs& a, const int t, const int bx) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int R = 16 * MT;
    // the descriptor is read in place (a modified local copy indexed by blockIdx.y would live in scratch memory);
    // with compile-time KO/KH the four extents below fold to constants
    const SgPolicyDesc& d = a.d;
    const int tid = threadIdx.x;
    const SgTrunk tr = d.trunk[t];
    // (run-time instances: the trunk's OWN hidden width -- a critic rebuilt by Policy.reset_critic differs from the actors)
    const int Op = (KO > 0 && KH > 0) ? 16 * KO : d.Op, Hp = (KO > 0 && KH > 0) ? 16 * KH : tr.Hp;
    const int ldO = Op + 4, ldH = Hp + 4, ldP = a.ldP;
    const float* W = GW ? a.params + tr.off : smem;
    float* X = smem + a.wbuf_floats;
    float* H1 = X + R * ldO;
    float* H2 = H1 + R * ldH;
    const int row0 = bx * R;

    SG_PPO_STAMP(0);
    float4 wv[12];
    if (!GW) sg_stage_issue<12>(wv, a.params + tr.off, tr.size / 4);
    {   // the row tile: contiguous in the epoch's permuted copy
        const float4* gx = reinterpret_cast<const float4*>(a.X + (size_t)row0 * ldO);
        float4* lx = reinterpret_cast<float4*>(X);
        for (int i = tid; i < R * ldO / 4; i += blockDim.x) lx[i] = gx[i];
    }
    SG_PPO_STAMP(1);
    if (!GW) sg_stage_commit<12>(smem, wv, a.params + tr.off, tr.size / 4);
    __syncthreads();
    SG_PPO_STAMP(2);
    const float* b1 = W + tr.b1;
    const float* b2 = W + tr.b2;
    const float* bh = W + tr.bh;
    float* gH1 = a.H1[t] + (size_t)row0 * ldH;
    float* gH2 = a.H2[t] + (size_t)row0 * ldH;
    float* gOUT = a.OUT[t] + (size_t)row0 * ldP;
    sg_layer_nt_u<MT, GW>(X, ldO, W + tr.w1, ldO, Op, Hp, [&](int r, int c, float v) {
        const float h = sg_tanh(v + b1[c]);
        H1[r * ldH + c] = h;
        gH1[r * ldH + c] = h;
    });
    __syncthreads();
    SG_PPO_STAMP(3);
    sg_layer_nt_u<MT, GW>(H1, ldH, W + tr.w2, ldH, Hp, Hp, [&](int r, int c, float v) {
        const float h = sg_tanh(v + b2[c]);
        H2[r * ldH + c] = h;
        gH2[r * ldH + c] = h;
    });
    __syncthreads();
    SG_PPO_STAMP(4);
    sg_layer_nt_u<MT, GW>(H2, ldH, W + tr.wh, ldH, Hp, tr.Pp, [&](int r, int c, float v) { gOUT[r * ldP + c] = v + bh[c]; });
    __syncthreads();
    SG_PPO_STAMP(5);
}

template <int MT, int KO, int KH, bool GW = false>
__global__ __launch_bounds__(512) void k_ppo_fwd(PpoArgs a) {
    sg_ppo_fwd_body<MT, KO, KH, GW>(a, blockIdx.y, blockIdx.x);
}

// -------------------------------------------------------------------------------- backward
// FUSED (Policy: the actor and critic trunks do not depend on each other's outputs): the workgroup stages the
// whole trunk, recomputes the forward on its own rows in LDS and goes straight on to the loss -- no k_ppo_fwd
// launch, no activation stacks written, flushed and read back (5 MB per step at the north-star shape).
// Launched with 8 waves (512 threads) for 32-row groups and up: the layer GEMMs are dealt out per 16x16 output tile
// (sg_layer_*_u), the bias gradients come from column sums of the finished dZ tiles, and two waves per SIMD hide each
// other's LDS / barrier latencies -- what two co-resident 16-row workgroups per CU did, with the weights staged once and
// one slab per 32 rows.
// PAIR (SplitPolicy, FUSED, every (row group, trunk) workgroup resident at once -- k_ppo_pair): a row's log-prob sums over the
// heads of BOTH actor trunks (a2c/model_split.py:201-238), which is why rounds 1-3 ran the forward as a launch of its own.  Here
// the two actor workgroups of a row group swap their [R x Pp] head outputs inside the launch, as write-through {value, step tag}
// words the partner polls (see the swap below).  The values are those the forward launch used to leave in the OUT
// stacks and the loss code below is unchanged: results are bit-identical to the two-launch step.  Both workgroups wait for each
// other, so they must be resident together: they are neighbours in dispatch order (trunk index fastest) and the launch is only
// used when all its workgroups fit the chip at once; the spin is bounded by the wall clock all the same.
template <int MT, int KO, int KH, bool FUSED, bool GW = false, bool PAIR = false>
__device__ __forceinline__ void sg_ppo_bwd_body(const PpoArgs& a, const int t, const int bx) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // (No spare workgroup for Adam's bias corrections any more: with G x trunks = 256 row-group blocks, two extra blocks
    // made 258 for 256 CUs, and whenever the dispatcher doubled two row groups up on one CU before the spare blocks had
    // exited, those pairs finished 4 us after everyone else.  k_ppo_adam prepares the NEXT step's scalars instead.)
    constexpr int R = 16 * MT;
    // the descriptor is read in place (a modified local copy indexed by blockIdx.y would live in scratch memory);
    // with compile-time KO/KH the four extents below fold to constants
    const SgPolicyDesc& d = a.d;
    const int tid = threadIdx.x;
    const SgTrunk tr = d.trunk[t];
    const int Op = (KO > 0 && KH > 0) ? 16 * KO : d.Op, Hp = (KO > 0 && KH > 0) ? 16 * KH : tr.Hp;   // the trunk's own width
    const bool critic = t == d.n_trunks - 1;
    const bool mlp = d.kind == SG_POLICY_MLP;
    const int ldO = Op + 4, ldH = Hp + 4, ldP = a.ldP, A = d.A;
    // LDS image of the trunk block from w2 on (w1/b1 are not needed going backward), or all of it when fused
    float* Wimg = smem;
    const int w_first = FUSED ? 0 : tr.w2;
    const float* W = GW ? a.params + tr.off : Wimg - w_first;   // so that W + tr.<off> addresses the block as usual
    const int wfl = tr.size - w_first;
    float* X = Wimg + a.wbuf_floats;                            // GW: wbuf_floats = 0, LDS holds the row tiles only
    float* H1 = X + R * ldO;
    float* H2 = H1 + R * ldH;
    float* O0 = H2 + R * ldH;                      // critic: value head; actors: trunk 0's head outputs
    float* O1 = O0 + R * ldP;                      // split: trunk 1's head outputs; MLP: per-row d/d logstd
    float* ACT = O1 + R * ldP;
    float* SC = ACT + ((R * A + 3) & ~3);          // [4][R]
    float* ROWL = SC + 4 * R;                      // [2][R]
    int* VALID = reinterpret_cast<int*>(ROWL + 2 * R);
    const int row0 = bx * R;
    float* slab = a.slabs + (size_t)bx * a.slab_stride;

    SG_PPO_STAMP(8);
    SG_PPO_WALL(6);
    unsigned pair_tag = 0;
    bool pair_failed = false;
    if (PAIR && !critic) pair_tag = (unsigned)(a.st->t0 + a.k1);
    // every global load of the block is issued before the first LDS store: one memory round trip, not six
    float4 wv[12];
    if (!GW) sg_stage_issue<12>(wv, a.params + tr.off + w_first, wfl / 4);
    constexpr int UX = MT <= 2 ? 4 : 8;          // float4 per thread for an [R][ld <= 116] tile at 256 threads
    constexpr int UO = MT <= 2 ? 2 : 4;          // ... for an [R][ldP <= 64] tile
    const int ta = critic ? t : 0;
    const bool two_heads = !critic && !mlp;
    const float* gX = a.X + (size_t)row0 * ldO;
    const float* gH1 = a.H1[t] + (size_t)row0 * ldH;
    const float* gH2 = a.H2[t] + (size_t)row0 * ldH;
    const float* gO0 = a.OUT[ta] + (size_t)row0 * ldP;
    const float* gO1 = a.OUT[two_heads ? 1 : ta] + (size_t)row0 * ldP;
    float4 xv[UX], h1v[UX], h2v[UX], o0v[UO], o1v[UO];
    sg_stage_issue<UX>(xv, gX, R * ldO / 4);
    if (!FUSED) {
        sg_stage_issue<UX>(h1v, gH1, R * ldH / 4);
        sg_stage_issue<UX>(h2v, gH2, R * ldH / 4);
        sg_stage_issue<UO>(o0v, gO0, R * ldP / 4);
        if (two_heads) sg_stage_issue<UO>(o1v, gO1, R * ldP / 4);
    }
    float actv[2] = {0.f, 0.f}, scv[4] = {0.f, 0.f, 0.f, 0.f};
    if (!critic) {
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int i = tid + u * 256; actv[u] = a.ACT[(size_t)row0 * A + (i < R * A ? i : 0)]; }
    }
    {
        const int r = tid < R ? tid : 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) scv[q] = a.SC[(size_t)q * a.sc_stride + row0 + r];
    }
    sg_stage_commit<UX>(X, xv, gX, R * ldO / 4);
    if (!FUSED) {
        sg_stage_commit<UX>(H1, h1v, gH1, R * ldH / 4);
        sg_stage_commit<UX>(H2, h2v, gH2, R * ldH / 4);
        sg_stage_commit<UO>(O0, o0v, gO0, R * ldP / 4);
        if (two_heads) sg_stage_commit<UO>(O1, o1v, gO1, R * ldP / 4);
    }
    if (!critic) {
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int i = tid + u * 256; if (i < R * A) ACT[i] = actv[u]; }
        for (int i = tid + 512; i < R * A; i += blockDim.x) ACT[i] = a.ACT[(size_t)row0 * A + i];
    }
    if (tid < R) {
        VALID[tid] = row0 + tid < a.mb;
#pragma unroll
        for (int q = 0; q < 4; ++q) SC[q * R + tid] = scv[q];
    }
    SG_PPO_STAMP(0);
    if (!GW) sg_stage_commit<12>(Wimg, wv, a.params + tr.off + w_first, wfl / 4);
    __syncthreads();
    SG_PPO_STAMP(1);
    if (FUSED) {   // forward on this row group (a2c/model.py:255-264, a2c/distributions.py:109-118), activations stay in LDS
        const float* b1 = W + tr.b1;
        const float* b2 = W + tr.b2;
        const float* bh = W + tr.bh;
        sg_layer_nt_u<MT, GW>(X, ldO, W + tr.w1, ldO, Op, Hp, [&](int r, int c, float v) { H1[r * ldH + c] = sg_tanh(v + b1[c]); });
        __syncthreads();
        SG_PPO_STAMP(2);
        sg_layer_nt_u<MT, GW>(H1, ldH, W + tr.w2, ldH, Hp, Hp, [&](int r, int c, float v) { H2[r * ldH + c] = sg_tanh(v + b2[c]); });
        __syncthreads();
        SG_PPO_STAMP(3);
        float* Oown = (PAIR && t == 1) ? O1 : O0;
        sg_layer_nt_u<MT, GW>(H2, ldH, W + tr.wh, ldH, Hp, tr.Pp, [&](int r, int c, float v) { Oown[r * ldP + c] = v + bh[c]; });
        __syncthreads();
        SG_PPO_STAMP(4);
        if (PAIR && !critic) {
            // The swap, low-latency form: every head output travels as an 8-byte {value, step tag} word (one aligned 8-byte
            // write-through store: the pair arrives together or not at all), and the partner polls the words it needs until they
            // carry this step's tag -- no store drain, no separate flag, no second round trip for the data behind a flag.  The
            // words live in the H1 row stacks, which the fused body does not use (2 Pp <= ldH: checked by the launch).
            float* Ooth = t == 1 ? O0 : O1;
            // (8-byte aligned: the stacks start at whatever float offset the epoch copy ends on, and a word that straddles an
            // 8-byte boundary is two stores -- a reader could see this step's tag beside the previous step's value)
            auto word_base = [](float* p) { return reinterpret_cast<unsigned long long*>((reinterpret_cast<uintptr_t>(p) + 7) & ~(uintptr_t)7); };
            unsigned long long* xown = word_base(a.H1[t] + (size_t)row0 * ldH);
            const unsigned long long* xoth = word_base(a.H1[1 - t] + (size_t)row0 * ldH);
            const int Pown = tr.Pp, Poth = d.trunk[1 - t].Pp;
            for (int i = tid; i < R * Pown; i += blockDim.x) {
                const int r = i / Pown, c = i - r * Pown;
                const unsigned long long w = ((unsigned long long)pair_tag << 32) | (unsigned long long)__float_as_uint(Oown[r * ldP + c]);
                __hip_atomic_store(xown + i, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // (a time-out is sticky: with the error word up, the actors of every later launch skip the wait)
            bool ok = __hip_atomic_load(a.pair + SG_PAIR_ERR_WORD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
            const long long deadline = wall_clock64() + SG_PAIR_TIMEOUT_TICKS;
            for (int i = tid; ok && i < R * Poth; i += blockDim.x) {
                const int r = i / Poth, c = i - r * Poth;
                unsigned long long w = 0;
                for (int it = 0;; ++it) {
                    w = __hip_atomic_load(xoth + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned)(w >> 32) == pair_tag) break;
                    if ((it & 31) == 31 && wall_clock64() > deadline) { ok = false; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                Ooth[r * ldP + c] = __uint_as_float((unsigned)w);
            }
            pair_failed = __syncthreads_or(!ok) != 0;
        }
    }
    SG_PPO_STAMP(9);

    // ---- loss and d(loss)/d(head outputs)  (a2c/algo/ppo.py:92-106)
    if (critic) {
        if (tid < R) {
            const int r = tid;
            const float v = O0[r * ldP];
            float dv = 0.f, lv = 0.f;
            if (VALID[r]) {
                const float Rt = SC[3 * R + r], vo = SC[2 * R + r];
                if (a.use_clipped) {
                    const float dvv = v - vo;
                    const float vc = vo + fminf(fmaxf(dvv, -a.clip), a.clip);
                    const float u = (v - Rt) * (v - Rt), w = (vc - Rt) * (vc - Rt);
                    const float m1 = u > w ? 1.f : (u < w ? 0.f : 0.5f);          // torch.max tie -> 1/2, 1/2
                    const float pass = (dvv >= -a.clip && dvv <= a.clip) ? 1.f : 0.f;
                    dv = 0.5f * a.inv_B * (m1 * 2.f * (v - Rt) + (1.f - m1) * 2.f * (vc - Rt) * pass);
                    lv = 0.5f * fmaxf(u, w);
                } else {
                    dv = 0.5f * a.inv_B * (-2.f) * (Rt - v);
                    lv = 0.5f * (Rt - v) * (Rt - v);
                }
                dv *= a.vcoef;
            }
            O0[r * ldP] = dv;
            ROWL[r] = lv;
        }
    } else {
        // Gaussian log-prob / entropy of every row and d(loss)/d(mean, log-std).  All R rows in ONE pass: a row gets
        // L = blockDim / R lanes (16, 8 or 4), a lane takes action dimensions sub, sub + L, ... (at most 8 of them: A <= 8 L
        // for every shipped policy); sigma and diff are computed once and kept in registers across the row reduction.
        // (Round 1 gave a row 32 lanes and walked the rows in R/8 passes: four dependent expf / logf / expf chains per
        // workgroup at 32-row groups, 6.4k of the block's 31k cycles.)
        const SgTrunk tra = d.trunk[0];
        const int L = blockDim.x / R;
        if (A <= 8 * L) {
            const int r = tid / L, sub = tid - r * L;
            float* o0 = O0 + r * ldP;
            float* o1 = O1 + r * ldP;
            float sig[8], dif[8];
            float* pmv[8];
            float* plv[8];
            float lp = 0.f, en = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = sub + j * L;
                sig[j] = 1.f; dif[j] = 0.f; pmv[j] = nullptr; plv[j] = nullptr;
                if (k < A) {
                    float mean, ls;
                    if (mlp) { pmv[j] = o0 + k; plv[j] = o1 + k; mean = *pmv[j]; ls = W[tra.ex + k]; }
                    else if (k < d.nc) { pmv[j] = o0 + k; plv[j] = o0 + d.nc + k; mean = *pmv[j]; ls = *plv[j]; }
                    else { pmv[j] = o1 + (k - d.nc); plv[j] = o1 + (d.na + k - d.nc); mean = *pmv[j]; ls = *plv[j]; }
                    // hardware exp2 / rcp (1 ulp) and log(exp(ls)) taken as ls: relative deviations of 1e-7 against the
                    // reference's libm chain, four orders below the tolerance of the per-step parity tests, and 0.9k
                    // cycles less on the one phase where every wave of the block waits on transcendental latency
                    sig[j] = __expf(ls);
                    dif[j] = ACT[r * A + k] - mean;
                    const float lsig = ls;
                    sig[j] = __builtin_amdgcn_rcpf(sig[j] * sig[j]);   // 1 / var from here on
                    lp += -0.5f * (dif[j] * dif[j]) * sig[j] - lsig - HALF_LOG_2PI;
                    en += 0.5f + HALF_LOG_2PI + lsig;
                }
            }
            if (L <= 16) {   // L = 4, 8 or 16 consecutive lanes inside one DPP row: VALU butterflies, no LDS crossbar
                lp = sg_dpp_add<0xB1>(lp); en = sg_dpp_add<0xB1>(en);
                lp = sg_dpp_add<0x4E>(lp); en = sg_dpp_add<0x4E>(en);
                if (L >= 8) { lp = sg_dpp_add<0x141>(lp); en = sg_dpp_add<0x141>(en); }
                if (L >= 16) { lp = sg_dpp_add<0x140>(lp); en = sg_dpp_add<0x140>(en); }
            } else {
                for (int o = 1; o < L; o <<= 1) { lp += __shfl_xor(lp, o); en += __shfl_xor(en, o); }
            }
            const float logp = lp, ent = en;
            const bool valid = VALID[r];
            float dlogp = 0.f, la = 0.f;
            if (valid) {
                const float adv = SC[1 * R + r];
                const float ratio = __expf(logp - SC[0 * R + r]);
                const float surr1 = ratio * adv;
                const float surr2 = fminf(fmaxf(ratio, 1.f - a.clip), 1.f + a.clip) * adv;
                const float w1 = surr1 < surr2 ? 1.f : (surr1 > surr2 ? 0.f : 0.5f);  // torch.min tie -> 1/2, 1/2
                const float inr = (ratio >= 1.f - a.clip && ratio <= 1.f + a.clip) ? 1.f : 0.f;
                dlogp = -a.inv_B * (w1 * adv + (1.f - w1) * adv * inr) * ratio;
                la = -fminf(surr1, surr2);
            }
            const float dent = valid ? a.ecoef * a.inv_B : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = sub + j * L;
                if (k < A) {
                    const float rvar = sig[j];
                    *pmv[j] = dlogp * dif[j] * rvar;
                    *plv[j] = dlogp * (dif[j] * dif[j] * rvar - 1.f) - dent;
                } else if (mlp && k < tra.Pp) {
                    o1[k] = 0.f;   // o1 doubles as the per-row d/d logstd tile: clear its padding columns
                }
            }
            if (sub == 0) { ROWL[r] = la; ROWL[R + r] = valid ? ent : 0.f; }
        } else {
        const int rows_per_pass = blockDim.x >> 5;
        for (int r = tid >> 5; r < R; r += rows_per_pass) {
            float* o0 = O0 + r * ldP;
            float* o1 = O1 + r * ldP;
            float logp = 0.f, ent = 0.f;
            for (int k0 = 0; k0 < A; k0 += 32) {
                const int k = k0 + (tid & 31);
                float lp = 0.f, en = 0.f;
                if (k < A) {
                    float mean, ls;
                    if (mlp) { mean = o0[k]; ls = W[tra.ex + k]; }
                    else if (k < d.nc) { mean = o0[k]; ls = o0[d.nc + k]; }
                    else { mean = o1[k - d.nc]; ls = o1[d.na + k - d.nc]; }
                    const float sigma = expf(ls), diff = ACT[r * A + k] - mean, lsig = logf(sigma);
                    lp = -(diff * diff) / (2.f * sigma * sigma) - lsig - HALF_LOG_2PI;
                    en = 0.5f + HALF_LOG_2PI + lsig;
                }
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) { lp += __shfl_xor(lp, o); en += __shfl_xor(en, o); }
                logp += lp;
                ent += en;
            }
            const bool valid = VALID[r];
            float dlogp = 0.f, la = 0.f;
            if (valid) {
                const float adv = SC[1 * R + r];
                const float ratio = expf(logp - SC[0 * R + r]);
                const float surr1 = ratio * adv;
                const float surr2 = fminf(fmaxf(ratio, 1.f - a.clip), 1.f + a.clip) * adv;
                const float w1 = surr1 < surr2 ? 1.f : (surr1 > surr2 ? 0.f : 0.5f);  // torch.min tie -> 1/2, 1/2
                const float inr = (ratio >= 1.f - a.clip && ratio <= 1.f + a.clip) ? 1.f : 0.f;
                dlogp = -a.inv_B * (w1 * adv + (1.f - w1) * adv * inr) * ratio;
                la = -fminf(surr1, surr2);
            }
            const float dent = valid ? a.ecoef * a.inv_B : 0.f;
            for (int k0 = 0; k0 < A; k0 += 32) {
                const int k = k0 + (tid & 31);
                if (k < A) {
                    float mean, ls, *pm, *pl;
                    if (mlp) { mean = o0[k]; ls = W[tra.ex + k]; pm = o0 + k; pl = o1 + k; }
                    else if (k < d.nc) { pm = o0 + k; pl = o0 + d.nc + k; mean = *pm; ls = *pl; }
                    else { pm = o1 + (k - d.nc); pl = o1 + (d.na + k - d.nc); mean = *pm; ls = *pl; }
                    const float sigma = expf(ls), var = sigma * sigma, diff = ACT[r * A + k] - mean;
                    *pm = dlogp * diff / var;
                    *pl = dlogp * (diff * diff / var - 1.f) - dent;
                } else if (mlp && k < tra.Pp) {
                    o1[k] = 0.f;
                }
            }
            if ((tid & 31) == 0) { ROWL[r] = la; ROWL[R + r] = valid ? ent : 0.f; }
        }
        }
    }
    __syncthreads();
    SG_PPO_STAMP(10);
    float* dout = (critic || t == 0) ? O0 : O1;     // this trunk's d loss / d head outputs
    // loss sums of this row group (recorded once: by the critic and by actor trunk 0)
    if (tid < 64 && (critic || t == 0)) {   // first wave: R <= 64 row losses, one per lane
        float s0 = tid < R ? ROWL[tid] : 0.f, s1 = (tid < R && !critic) ? ROWL[R + tid] : 0.f;
        s0 = sg_wave_sum(s0); s1 = sg_wave_sum(s1);
        if (tid == 0) {
            float* ls = slab + d.total;
            if (critic) ls[0] = s0;
            else { ls[1] = s0; ls[2] = s1; }
        }
    }
    float* g = slab + tr.off;
    // head weight / bias gradients (needs h2 before it is overwritten)
    sg_grad_tn<MT, (MT >= 2 ? 8 : 0), true>(dout, ldP, H2, ldH, tr.Pp, Hp, g + tr.wh, ldH, false);
    sg_colsum(dout, ldP, R, tr.Pp, g + tr.bh, false);
    if (tr.EX) sg_colsum(O1, ldP, R, SG_PAD16(tr.EX), g + tr.ex, false);
    SG_LDS_SYNC();
    SG_PPO_STAMP(11);
    {
        // per-tile tasks; dZ = (dY W) * (1 - h^2) in place over h, the bias gradient from column sums of the finished tile
        auto dz_u = [&](float* h) { return [=](int r, int c, float v) { float* ph = h + r * ldH + c; const float hv = *ph; *ph = v * (1.f - hv * hv); }; };
        sg_layer_nn_u<MT>(dout, ldP, W + tr.wh, ldH, tr.Pp, Hp, dz_u(H2));
        SG_LDS_SYNC();
        SG_PPO_STAMP(12);
        sg_grad_tn<MT, (MT >= 2 ? 8 : 0), true>(H2, ldH, H1, ldH, Hp, Hp, g + tr.w2, ldH, false);
        SG_PPO_STAMP(5);
        sg_colsum(H2, ldH, R, Hp, g + tr.b2, false);
        SG_LDS_SYNC();
        SG_PPO_STAMP(13);
        sg_layer_nn_u<MT>(H2, ldH, W + tr.w2, ldH, Hp, Hp, dz_u(H1));
        SG_LDS_SYNC();
        SG_PPO_STAMP(14);
        sg_grad_tn<MT, (MT >= 2 ? 8 : 0), true>(H1, ldH, X, ldO, Hp, Op, g + tr.w1, ldO, false);
        sg_colsum(H1, ldH, R, Hp, g + tr.b1, false);
        SG_LDS_SYNC();
    }
    SG_PPO_STAMP(15);
    SG_PPO_WALL(7);
    if (PAIR && pair_failed && tid == 0) {   // the partner never published: the host sees NaN losses, then the error word
        atomicOr(a.pair + SG_PAIR_ERR_WORD, 1u);
        slab[d.total + 1] = __builtin_nanf("");
    }
}

template <int MT, int KO, int KH, bool FUSED = false, bool GW = false>
__global__ __launch_bounds__(512) void k_ppo_bwd(PpoArgs a) {
    sg_ppo_bwd_body<MT, KO, KH, FUSED, GW>(a, blockIdx.y, blockIdx.x);
}

// SplitPolicy, ONE launch per step when its 3 G workgroups fit the chip together: every trunk's fused forward + loss + backward,
// the two actors of a row group joined by the hand-off described at sg_ppo_bwd_body.  Workgroup 3 g + y is trunk y of row group
// g (y = 2: the critic), so partners are dispatched back to back.
template <int MT, int KO, int KH>
__global__ __launch_bounds__(512) void k_ppo_pair(PpoArgs a) {
    const int bx = (int)blockIdx.x / 3, t = (int)blockIdx.x - 3 * bx;
    sg_ppo_bwd_body<MT, KO, KH, true, false, true>(a, t, bx);
}

// SplitPolicy at minibatches whose (row groups x 3 trunks) exceed the CU count: the critic trunk needs no actor output (its
// loss is the value loss alone), so it does not have to wait at the forward -> backward seam of the two actor trunks (a row's
// log-prob sums over both actors' heads).  This launch does the ACTORS' forward (blockIdx.y = 1, 2 -> trunks 0, 1) and the
// CRITIC's whole fused forward + loss + backward (blockIdx.y = 0: dispatched first, they are the long blocks); the backward
// launch that follows covers the two actor trunks only.  At 4096-row minibatches that is 384 + 256 workgroups on 256 CUs with
// the critic's 128 long blocks beside two half-rounds of forward blocks, instead of 384 + 384 in two rounds each.
template <int MT, int KO, int KH>
__global__ __launch_bounds__(512) void k_ppo_fwd_critic(PpoArgs a) {
    if (blockIdx.y == 0) sg_ppo_bwd_body<MT, KO, KH, true>(a, a.d.n_trunks - 1, blockIdx.x);
    else sg_ppo_fwd_body<MT, KO, KH>(a, blockIdx.y - 1, blockIdx.x);
}

// grad[i] = sum over slabs; part[block] = sum of squares of this block's grads.  A block owns 64 consecutive
// parameters; its 4 waves each sum a quarter of the slabs (16 independent loads in flight per lane, i.e. two
// round trips for 128 slabs instead of sixteen) and combine through LDS in a fixed order.
#define SG_PPO_REDUCE_PARAMS 64
__global__ __launch_bounds__(256) void k_ppo_reduce(const float* slabs, int n_slabs, int slab_stride, int total,
                                                    float* grad, float* part) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * SG_PPO_REDUCE_PARAMS + lane;
    float g = 0.f;
    if (i < total + 8) {
        for (int s0 = wave; s0 < n_slabs; s0 += 64) {
            float t[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int sidx = s0 + 4 * u;
                t[u] = slabs[(size_t)(sidx < n_slabs ? sidx : s0) * slab_stride + i];   // clamped, zeroed below: no branch
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) g += (s0 + 4 * u < n_slabs) ? t[u] : 0.f;
        }
    }
    red[wave][lane] = g;
    __syncthreads();
    if (wave == 0) {
        g = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        if (i < total + 8) grad[i] = g;
        float sq = (i < total) ? g * g : 0.f;
        sq = sg_wave_sum(sq);
        if (lane == 0) part[blockIdx.x] = sq;
    }
}

// Adam scalars of the update's first step (every later step's are prepared by the preceding k_ppo_adam)
__global__ void k_opt_prepare_first(SgOptState* st) { sg_opt_prepare(st, st->t0 + 1); }

// sum of squares only (data-parallel mode: recomputed after the all-reduce)
__global__ __launch_bounds__(256) void k_sumsq(const float* grad, int total, float* part) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float sq = (i < total) ? grad[i] * grad[i] : 0.f;
    sq = sg_wave_sum(sq);
    __shared__ float ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// clip_grad_norm_ + Adam (a2c/algo/ppo.py:143-145; torch.optim.Adam single-tensor math).
// (argument order: everything the first instructions need sits in the 16 dwords the command processor preloads)
__global__ __launch_bounds__(256) void k_ppo_adam(float* params, float* m, float* v, const float* grad,
                                                  const float* part, int n_part, int total,
                                                  const SgOptState* st, int k1, float eps, float max_norm,
                                                  float inv_mb, double* loss_acc) {
    __shared__ float s_coef;
    // Adam step t = st->t0 + k1; its bias-correction scalars were prepared in slot t & 1 by this step's k_ppo_bwd
    const int t = st->t0 + k1;
    const float s_step_size = st->step_size2[t & 1], s_bc2_sqrt = st->bc2_sqrt2[t & 1];
    __shared__ float s_ws[4];
    {   // ||g||^2 from the per-block partial sums: all 256 threads load (a few independent loads each)
        float s = 0.f;
        for (int j = threadIdx.x; j < n_part; j += 256) s += part[j];
        s = sg_wave_sum(s);
        if ((threadIdx.x & 63) == 0) s_ws[threadIdx.x >> 6] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float norm = sqrtf((s_ws[0] + s_ws[1]) + (s_ws[2] + s_ws[3]));
        float coef = max_norm / (norm + 1e-6f);
        s_coef = coef > 1.f ? 1.f : coef;
    }
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        const float g = grad[i] * s_coef;
        float mi = m[i], vi = v[i];
        mi = mi + (g - mi) * (float)(1.0 - 0.9);
        vi = vi * (float)0.999 + (float)(1.0 - 0.999) * g * g;
        const float denom = sqrtf(vi) / s_bc2_sqrt + eps;
        params[i] = params[i] - s_step_size * (mi / denom);
        m[i] = mi;
        v[i] = vi;
    }
    if (blockIdx.x == 0 && threadIdx.x < 3)  // value_loss.item() etc. are float32, summed in Python doubles
        loss_acc[threadIdx.x] += (double)(grad[total + threadIdx.x] * inv_mb);
    // the next step's bias corrections (double pow) into the other slot: nobody reads that slot before the next k_ppo_adam
    if (blockIdx.x == 0 && threadIdx.x == 64) sg_opt_prepare(const_cast<SgOptState*>(st), t + 1);
}

// adv = returns[:-1] - value_preds[:-1]; sums for mean / unbiased std (a2c/algo/ppo.py:66-68)
__global__ __launch_bounds__(1024) void k_adv_stats(const float* ret, const float* vpred, int64_t n, float* adv,
                                                    double* stats /* [0]=sum, [1]=sumsq-about-mean, [2]=n */,
                                                    int pass) {
    __shared__ double ws[16];
    __shared__ double s_mean;
    const int tid = threadIdx.x;
    if (pass == 0) {
        double s = 0.0;
        for (int64_t i = tid; i < n; i += blockDim.x) {
            const float a = ret[i] - vpred[i];
            adv[i] = a;
            s += (double)a;
        }
        s = sg_wave_sum(s);
        if ((tid & 63) == 0) ws[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += ws[w];
            stats[0] = t;
            stats[2] = (double)n;
        }
    } else if (pass == 1) {
        if (tid == 0) s_mean = (double)(float)(stats[0] / stats[2]);
        __syncthreads();
        const double mean = s_mean;
        double s = 0.0;
        for (int64_t i = tid; i < n; i += blockDim.x) {
            const double dd = (double)adv[i] - mean;
            s += dd * dd;
        }
        s = sg_wave_sum(s);
        if ((tid & 63) == 0) ws[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += ws[w];
            stats[1] = t;
        }
    } else {
        const float mean = (float)(stats[0] / stats[2]);
        const float sd = (float)sqrt(stats[1] / (stats[2] - 1.0));
        for (int64_t i = tid + (int64_t)blockIdx.x * blockDim.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
            adv[i] = (adv[i] - mean) / (sd + 1e-5f);
    }
}
