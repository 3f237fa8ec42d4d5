/* sg_oracle_f64.c -- the float64 ARBITER: sg_oracle.c compiled a second time with every `float` a `double`
 * (libsg_oracle64.so, same entry points, double arrays).  TEST INFRASTRUCTURE, like the oracle itself.
 *
 * What it is for: over a whole update (160 clipped-surrogate Adam steps) two float32 evaluations of the reference's
 * algorithm -- the oracle and the HIP path, or the oracle and itself under 1-ulp noise -- drift apart by more than the
 * 1e-4 the single steps hold, because rows on a clip / min / max boundary flip branch.  Neither float32 result is "the"
 * answer there; the same algorithm carried out in float64 is the reference point both are measured against
 * (tools/parity_f64.py -> profiles/r06_parity_f64.json; tests/test_gpu_benchpath.py derives its trajectory gates from it).
 * It is NOT a parity oracle: the reference computes in float32 (torch CPU), so a float64 result differs from the
 * reference's own by float32 round-off and must never be used as the expected value of a parity test.
 *
 * Mechanism: the system headers sg_oracle.c needs are included first (their include guards make its own #includes no-ops),
 * then `float` and the float math functions are renamed, then the oracle's source is included verbatim.  Float literals
 * (1e-7f, 0.9f ...) keep their float32 VALUE, promoted: the constants are the reference's. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#define float double
#define expf exp
#define logf log
#define log1pf log1p
#define tanhf tanh
#define sqrtf sqrt
#define fabsf fabs
#define fminf fmin
#define fmaxf fmax
#include "sg_oracle.c"
