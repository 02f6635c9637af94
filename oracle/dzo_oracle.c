/*
 * dzo_oracle.c -- CPU oracle for the BFGS / L-BFGS step!() hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (dzoptimization.jl_amd/, include/)
 * may link, load or call this file.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, as the checker / reported baseline.
 *
 * It is a plain-C restatement of the reference's algorithm (dzhang314/DZOptimization.jl @
 * 2025-09-05); each function cites the reference file:line it follows (see
 * dzo_oracle_impl.h).  PARITY UNPINNED BY THE REFERENCE: upstream ships no tests, fixtures
 * or golden vectors for this path and there is no Julia in the build container, so the
 * oracle is pinned by analytic identities, an mpmath twin and run_and_test!-style invariants
 * (tests/test_oracle_*.py), and the committed fixtures under tests/golden/ are self-pinned.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_PROBLEM_ROSENBROCK2D 0
#define ORC_PROBLEM_ROSENBROCK_CHAIN 1
#define ORC_PROBLEM_QUADRATIC 2
#define ORC_PROBLEM_LSE 3
#define ORC_PROBLEM_QUADRATIC_CHAIN 4

/* 0 sequential | 1 eight-lane | 2 wide accumulator; see orc_dot. */
static int orc_dot_mode = 0;
/* OpenMP threads for the timing baseline; 1 = the deterministic scalar port. */
static int orc_threads = 1;

void orc_set_dot_mode(int mode) { orc_dot_mode = mode; }
int orc_get_dot_mode(void) { return orc_dot_mode; }
void orc_set_threads(int t) { orc_threads = t < 1 ? 1 : t; }
int orc_get_threads(void) { return orc_threads; }

/* ------------------------------------------------------------------------------------------
 * PCG32 input generator -- legacy/PCG.jl:7-22.
 *   advance: state * 0x5851F42D4C957F2D + 0x14057B7EF767814F            (:7-8)
 *   extract: rotr32((state ^ (state >> 18)) >> 27, state >> 59)          (:11-12)
 *   seeding: state = advance(0x14057B7EF767814F + seed)                  (:16)
 *   value:   2^-32 * u32  (Float64 product, then converted to eltype)    (:18)
 * ---------------------------------------------------------------------------------------- */
static inline uint64_t pcg_advance(uint64_t s) {
    return 0x5851F42D4C957F2DULL * s + 0x14057B7EF767814FULL;
}
static inline uint32_t pcg_extract(uint64_t s) {
    uint32_t v = (uint32_t)(((s >> 18) ^ s) >> 27);
    uint32_t r = (uint32_t)(s >> 59);
    return (v >> r) | (v << ((32u - r) & 31u)); /* bitrotate(v, -r) = rotate right by r */
}
void orc_pcg_fill_f64(double *x, int64_t n, uint64_t seed) {
    uint64_t s = pcg_advance(0x14057B7EF767814FULL + seed);
    for (int64_t i = 0; i < n; ++i) {
        x[i] = 2.3283064365386962890625E-10 * (double)pcg_extract(s);
        s = pcg_advance(s);
    }
}
void orc_pcg_fill_f32(float *x, int64_t n, uint64_t seed) {
    uint64_t s = pcg_advance(0x14057B7EF767814FULL + seed);
    for (int64_t i = 0; i < n; ++i) {
        x[i] = (float)(2.3283064365386962890625E-10 * (double)pcg_extract(s));
        s = pcg_advance(s);
    }
}
/* raw 32-bit outputs, for the known-answer test against the PCG32 XSH-RR definition */
void orc_pcg_raw_u32(uint32_t *out, int64_t n, uint64_t seed) {
    uint64_t s = pcg_advance(0x14057B7EF767814FULL + seed);
    for (int64_t i = 0; i < n; ++i) {
        out[i] = pcg_extract(s);
        s = pcg_advance(s);
    }
}

#define T double
#define ACC long double
#define SUF _f64
#define ORC_IS_F64 1
#include "dzo_oracle_impl.h"
#undef T
#undef ACC
#undef SUF
#undef ORC_IS_F64

#define T float
#define ACC double
#define SUF _f32
#define ORC_IS_F64 0
#include "dzo_oracle_impl.h"
#undef T
#undef ACC
#undef SUF
#undef ORC_IS_F64
