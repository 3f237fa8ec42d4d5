/* sg_test_api.h -- entry points of libsimgan_hip_test.so (sg_test.hip): test hooks and probes for tests/ and tools/.
 * Not part of the product ABI (include/simgan_hip.h); handles are the product library's. */
#pragma once
#include <stdint.h>

#include "../../include/simgan_hip.h"

#ifdef __cplusplus
extern "C" {
#endif
SG_API const char *sg_test_last_error(void);
/* C[M,N] = op(A) op(B) through the LDS/MFMA tile engine (mode 0 NT, 1 NN, 2 TN; 3 = the 4-row thin engine), host pointers. */
SG_API int sg_test_gemm(sg_ctx *ctx, int mode, int M, int N, int K, const float *A, const float *B, float *C);
SG_API int sg_test_gemm_bench(sg_ctx *ctx, int mode, int MT, int K, int Np, int threads, int iters, int epi, long long *cycles);
SG_API int sg_test_mfma_probe(sg_ctx *ctx, int abid, const float *a, const float *b, float *d);
SG_API int sg_test_flag_probe(sg_ctx *ctx, int mode, int np, int nc, int words, long long *stamps, float *sums);
SG_API int sg_test_fetch_probe(sg_ctx *ctx, int n_blocks, int waves, int mode, long long *out);
SG_API int sg_test_pstep_probe(sg_ctx *ctx, int mode, int S, int NW, int cyc_phase, int cyc_w, long long *stamps, int *err4);
SG_API int sg_test_pmc_calibrate(sg_ctx *ctx, int64_t mbytes);
/* out[0] / out[1]: how the last PPO update / discriminator epoch was issued: 0 direct, 1 replayed graph, 2 capture refused */
SG_API int sg_test_graph_state(sg_ppo *a, sg_disc *d, int out[2]);
SG_API int sg_test_disc_phase_times(sg_disc *d, int enable, long long *out, int n_blocks);
SG_API int sg_test_disc_step4_times(sg_disc *d, int enable, long long *out, int n_blocks);
SG_API int sg_test_ppo_phase_times(sg_ppo *a, int enable, long long *out, int n_blocks);
/* all-gathers the discriminator's replicated data-parallel mode has issued so far (it reuses the union across the epochs of an update) */
SG_API int sg_test_disc_gathers(sg_disc *d, long long *out);
/* kind 0: permutation of [0, n) -> int64 out; 1: uniform [0,1) -> float out; 2: standard normal -> float out */
/* one lane's write-through store of 16 / 8 bytes (aligned, or straddling its own size) read from another XCD: out4 = {torn words,
 * words read, largest value seen, reader lanes that saw the value change} (tools/tear_probe.py) */
SG_API int sg_test_tear_probe(sg_ctx *ctx, int mode, int pairs4, int iters, long long *out4);
SG_API int sg_test_rng(sg_ctx *ctx, int kind, int64_t n, uint64_t seed, void *out);
/* raises the sticky time-out word of k_disc_step4 (d) and / or k_ppo_pair (a) on the device, as a workgroup that gave up
 * waiting would: the next one-launch steps end at once with NaN losses (tests: how a queued update reports it) */
SG_API int sg_test_raise_handoff_error(sg_disc *d, sg_ppo *a);
#ifdef __cplusplus
}
#endif
