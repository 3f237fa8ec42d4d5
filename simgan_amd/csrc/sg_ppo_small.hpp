// sg_ppo_small.hpp -- k_ppo_small: one PPO EPOCH of a Policy (actor + critic trunk) as ONE launch when a minibatch is a single
// row group (a2c/algo/ppo.py:74-149 at the reference's own CPU-runnable geometry, BASELINE.json configs[0]: 8 processes x 128 steps,
// 32 minibatches -> 32-row optimizer steps, 320 per update).
//
// The three-launch step (k_ppo_bwd, k_ppo_reduce, k_ppo_adam) on such a grid is two workgroups of work and ~18 us of launch
// latency, 960 times per update.  Here one workgroup per trunk stays resident for the epoch's M steps: the trunk's parameter
// block, its Adam moments and its gradient live in LDS; a step is
//     the fused forward + loss + backward of k_ppo_bwd (the same body, weights not re-staged) -> gradient in LDS
//     -> the per-64-parameter sums of squares k_ppo_reduce forms, for the blocks inside this trunk
//     -> ONE exchange with the other trunk's workgroup: those sums, and the raw gradient values of the block that straddles
//        the trunk boundary, as 8-byte {value, step tag} words (write-through stores, polled; k_ppo_pair's carrier)
//     -> the clip coefficient from ALL blocks' sums in k_ppo_adam's order, Adam on this trunk's block in LDS.
// The global-norm clip is the only coupling between the trunks: a few hundred words per step.  Every sum is formed in the order
// the three kernels form it (one slab: the slab "sum" is the value itself), so parameters, moments and loss sums are
// bit-identical to the three-launch step (tests/test_gpu_fullsize.py::test_small_grid_epoch_*).  The words are double-buffered
// by step parity: a workgroup can overwrite the words of step k only after it has read the other's words of step k + 1, which
// that one publishes only after reading these.  The wait is bounded by the wall clock (sticky error word, NaN losses: the
// error path of k_ppo_pair); the tag is Adam's step number, as in k_ppo_pair; the launch needs its two workgroups resident together, i.e. the device to itself
// (sg_ctx_exclusive), like the other launches that wait inside themselves.
#pragma once
#include "sg_ppo_kernels.hpp"

#define SG_SMALL_WORDS 1024          // tagged words per trunk and parity: 64 head values | 64 tail values | block sums
#define SG_SMALL_MAX_TRUNKS 2
#ifndef SG_SMALL_STAMPS
#define SG_SMALL_STAMPS 0            // 1 (diagnostic builds): wall-clock stamps of the launch's last step behind the words, printed by sg_ppo_destroy
#endif
#define SG_SMALL_ST(n) do { if (SG_SMALL_STAMPS && tid == 0 && k == s.M - 1) (s.xbuf + 2 * SG_SMALL_MAX_TRUNKS * SG_SMALL_WORDS)[t * 16 + (n)] = (unsigned long long)wall_clock64(); } while (0)

struct SmallArgs {
    int M;                            // optimizer steps of this launch (one epoch)
    int rows_per_step;                // the epoch copy advances by this many rows per step
    unsigned long long* xbuf;         // [2][n_trunks][SG_SMALL_WORDS]
    float *params, *m, *v;            // the policy's parameter vector and the optimizer's moments (read at start, written at the end)
    double* loss_acc;                 // [3] value / action / entropy loss sums of the update
    float eps, max_norm;
    int extra_off;                    // floats: where this kernel's LDS buffers start (behind the body's)
    int gcap;                         // floats per LDS buffer (>= the largest trunk block + 8)
};

template <int MT, int KO, int KH>
__global__ __launch_bounds__(512) void k_ppo_small(PpoArgs a, SmallArgs s) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const SgPolicyDesc& d = a.d;
    const int t = blockIdx.x;                      // trunk
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const SgTrunk tr = d.trunk[t];
    const int lo = tr.off, hi = tr.off + tr.size, total = d.total;
    float* Wimg = smem;                            // the trunk's block (FUSED: the whole block from w1 on)
    float* Gb = smem + s.extra_off;                // gradient of the block (+ 8 unused)
    float* Mb = Gb + s.gcap;
    float* Vb = Mb + s.gcap;
    float* PART = Vb + s.gcap;                     // [n_part] per-64-parameter sums of squares of the WHOLE vector
    const int n_part = (total + 8 + SG_PPO_REDUCE_PARAMS - 1) / SG_PPO_REDUCE_PARAMS;
    float* EDGE = PART + ((n_part + 3) & ~3);      // [n_trunks][128]: head | tail raw values of every trunk
    float* LS = EDGE + 128 * SG_SMALL_MAX_TRUNKS;  // [0..2] loss sums of the step; [4..7] wave sums; [8] clip coefficient (16 floats), then Adam's scalars of the launch's steps [M][2]
    __shared__ int sh_ok;

    // ---- once: block, moments into LDS; gradient buffer cleared (padding columns are never written)
    for (int i = tid; i < tr.size; i += blockDim.x) { Wimg[i] = s.params[lo + i]; Mb[i] = s.m[lo + i]; Vb[i] = s.v[lo + i]; Gb[i] = 0.f; }
    if (tid < 8) Gb[tr.size + tid] = 0.f;
    const int t0 = a.st->t0;
    const float lr = a.st->lr;
    double acc = 0.0;
    if (tid < 3) acc = s.loss_acc[tid];             // thread j carries loss_acc[j] (written back by the owning trunk's workgroup)
    if (tid == 0) sh_ok = 1;
    // Adam's bias corrections of every step of the launch (sg_opt_prepare's arithmetic: two double pow each), one step per thread, once
    float* SCAL = LS + 16;                         // [M][2]
    for (int k = tid; k < s.M; k += blockDim.x) {
        const int tstep = t0 + a.k1 + k;
        const double bc1 = 1.0 - pow(0.9, (double)tstep), bc2 = 1.0 - pow(0.999, (double)tstep);
        SCAL[2 * k] = (float)((double)lr / bc1);
        SCAL[2 * k + 1] = (float)sqrt(bc2);
    }
    __syncthreads();

    // blocks of 64 parameters: [b_first, b_full_end) lie wholly inside [lo, hi); block b_edge = hi / 64 straddles hi (if hi % 64)
    const int b_first = (lo + 63) >> 6, b_full_end = hi >> 6;
    const int n_own = b_full_end - b_first;
    const bool has_edge = (hi & 63) != 0;
    unsigned long long* const xb = s.xbuf;
    const unsigned* err = a.pair + SG_PAIR_ERR_WORD;
    bool alive = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;

    PpoArgs ak = a;
    ak.gb_off = s.extra_off;                       // the body's gradient destination is the LDS buffer Gb, its loss sums go to LS
    ak.ls_off = (int)(LS - smem);
    ak.stage_mask = 0;                             // the block is resident: nothing is staged
    for (int k = 0; k < s.M; ++k) {
        const size_t rb = (size_t)k * s.rows_per_step;
        ak.X = a.X + rb * d.ldO; ak.ACT = a.ACT + rb * d.A; ak.SC = a.SC + rb;
        ak.k1 = a.k1 + k;
        SG_SMALL_ST(0);
        sg_ppo_bwd_body<MT, KO, KH, true, false, false, true>(ak, t, 0);
        __syncthreads();
        SG_SMALL_ST(1);
        const unsigned tag = (unsigned)(t0 + ak.k1);   // Adam's step number: never 0, never reused by this object (the host clears the words when it is set back)
        unsigned long long* mine = xb + ((size_t)(tag & 1) * d.n_trunks + t) * SG_SMALL_WORDS;
        // ---- this trunk's words: head values (parameters lo .. lo+63), tail values (the straddling block's lanes below hi),
        //      the sums of squares of its whole blocks (k_ppo_reduce: wave 0's lanes square the reduced value, sg_wave_sum)
        auto word = [&](float v) { return ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v); };
        if (tid < 64) __hip_atomic_store(mine + tid, word(0.f + Gb[tid]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (tid < 128) {
            const int i = (hi & ~63) + (tid - 64);          // parameter index
            __hip_atomic_store(mine + tid, word((has_edge && i < hi) ? 0.f + Gb[i - lo] : 0.f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        for (int j = wave; j < n_own; j += nw) {
            const int i = ((b_first + j) << 6) + lane;
            const float g = 0.f + Gb[i - lo];
            float sq = i < total ? g * g : 0.f;
            sq = sg_wave_sum(sq);
            if (lane == 0) {
                PART[b_first + j] = sq;
                __hip_atomic_store(mine + 128 + j, word(sq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (tid < 128) EDGE[t * 128 + tid] = tid < 64 ? 0.f + Gb[tid] : (((hi & ~63) + (tid - 64) < hi && has_edge) ? 0.f + Gb[(hi & ~63) + (tid - 64) - lo] : 0.f);
        SG_SMALL_ST(2);
        // ---- the other trunks' words
        bool ok = alive;
        const long long deadline = wall_clock64() + SG_PAIR_TIMEOUT_TICKS;
        for (int o = 0; o < d.n_trunks; ++o) {
            if (o == t) continue;
            const SgTrunk to = d.trunk[o];
            const int olo = to.off, ohi = to.off + to.size;
            const int ob_first = (olo + 63) >> 6, on_own = (ohi >> 6) - ob_first;
            const unsigned long long* theirs = xb + ((size_t)(tag & 1) * d.n_trunks + o) * SG_SMALL_WORDS;
            for (int i = tid; ok && i < 128 + on_own; i += blockDim.x) {
                unsigned long long w = 0;
                for (int it = 0;; ++it) {
                    w = __hip_atomic_load(theirs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned)(w >> 32) == tag) break;
                    if ((it & 31) == 31 && wall_clock64() > deadline) { ok = false; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                const float v = __uint_as_float((unsigned)w);
                if (i < 128) EDGE[o * 128 + i] = v;
                else PART[ob_first + (i - 128)] = v;
            }
        }
        if (!ok) sh_ok = 0;
        __syncthreads();
        SG_SMALL_ST(3);
        if (!sh_ok) {   // the other trunk's workgroup never published: NaN losses, sticky error word, parameters left as they are
            if (tid == 0) { atomicOr(a.pair + SG_PAIR_ERR_WORD, 1u); s.loss_acc[t == d.n_trunks - 1 ? 0 : 1] = __builtin_nan(""); }
            return;
        }
        // ---- the straddling blocks (and the tail block past `total`): everybody forms them from the raw values
        for (int q = wave; q < d.n_trunks; q += nw) {
            const SgTrunk tq = d.trunk[q];
            const int qhi = tq.off + tq.size;
            if ((qhi & 63) == 0) continue;
            const int b = qhi >> 6, i = (b << 6) + lane;
            float g = 0.f;
            if (i < qhi) g = EDGE[q * 128 + 64 + lane];                               // trunk q's tail
            else if (q + 1 < d.n_trunks) g = EDGE[(q + 1) * 128 + (i - qhi)];         // the next trunk's head
            float sq = i < total ? g * g : 0.f;
            sq = sg_wave_sum(sq);
            if (lane == 0) PART[b] = sq;
        }
        // (blocks past the last parameter hold only the 8 loss slots: zero)
        for (int b = ((total + 63) >> 6) + tid; b < n_part; b += blockDim.x) PART[b] = 0.f;
        __syncthreads();
        // ---- k_ppo_adam: ||g||^2 from the block sums (256 threads, each its strided share, wave sums, pairwise), clip, Adam
        if (tid < 256) {
            float sum = 0.f;
            for (int j = tid; j < n_part; j += 256) sum += PART[j];
            sum = sg_wave_sum(sum);
            if (lane == 0) LS[4 + wave] = sum;
        }
        __syncthreads();
        if (tid == 0) {
            const float norm = sqrtf((LS[4] + LS[5]) + (LS[6] + LS[7]));
            const float coef = s.max_norm / (norm + 1e-6f);
            LS[8] = coef > 1.f ? 1.f : coef;
        }
        __syncthreads();
        SG_SMALL_ST(4);
        const float coef = LS[8], step_size = SCAL[2 * k], bc2_sqrt = SCAL[2 * k + 1];
#pragma unroll 4
        for (int i = tid; i < tr.size; i += blockDim.x) {
            const float g = (0.f + Gb[i]) * coef;
            float mi = Mb[i], vi = Vb[i];
            mi = mi + (g - mi) * (float)(1.0 - 0.9);
            vi = vi * (float)0.999 + (float)(1.0 - 0.999) * g * g;
            const float denom = sqrtf(vi) / bc2_sqrt + s.eps;
            Wimg[i] = Wimg[i] - step_size * (mi / denom);
            Mb[i] = mi;
            Vb[i] = vi;
        }
        // loss sums: value loss from the critic's workgroup, action loss and entropy from the actor's (k_ppo_adam: float product, double sum)
        if (t == d.n_trunks - 1) { if (tid == 0) acc += (double)((0.f + LS[0]) * a.inv_B); }
        else if (t == 0 && (tid == 1 || tid == 2)) acc += (double)((0.f + LS[tid]) * a.inv_B);
        __syncthreads();
        SG_SMALL_ST(5);
    }
    // ---- the epoch's results: block and moments back to memory, loss sums, the next step's Adam scalars
    for (int i = tid; i < tr.size; i += blockDim.x) { s.params[lo + i] = Wimg[i]; s.m[lo + i] = Mb[i]; s.v[lo + i] = Vb[i]; }
    if (t == d.n_trunks - 1 && tid == 0) s.loss_acc[0] = acc;
    if (t == 0 && (tid == 1 || tid == 2)) s.loss_acc[tid] = acc;
    if (t == 0 && tid == 64) sg_opt_prepare(a.st, t0 + a.k1 + s.M);
}
