/* bench_lbfgs.c -- the headline workload (BASELINE configs[2]: L-BFGS m = 20 on the N-D chained Rosenbrock function,
 * n = 10^7, fp64) driven through the plain C ABI of include/dzo.h alone: no Python, no PyTorch, no ctypes between the
 * host loop and the library.  Same inputs as bench.py (x0_i = -1.2 / 1.0 alternating + 0.01 (u_i - 1/2),
 * u = PCG32 XSH-RR with the reference's seeding, legacy/PCG.jl:7-22, seed 5), same protocol: m untimed steps fill the
 * history, W warm-up steps, K timed steps between two synchronisations.  What it shows: the step rate of the C ABI is the
 * step rate bench.py reports -- the Python binding adds nothing measurable to a 0.4-ms step.
 *
 *   gcc -O2 -Iinclude examples/bench_lbfgs.c -Ldzoptimization.jl_amd -ldzo_hip \
 *       -Wl,-rpath,$PWD/dzoptimization.jl_amd -o bench_lbfgs && ./bench_lbfgs [n [m [steps [warmup]]]]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "dzo.h"

#define CHECK(call)                                                                  \
    do {                                                                             \
        int32_t rc_ = (call);                                                        \
        if (rc_ != DZO_OK) {                                                         \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, dzo_last_error());   \
            return 1;                                                                \
        }                                                                            \
    } while (0)

static double now_s(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

/* legacy/PCG.jl:7-22: state = advance(increment + seed); u = 2^-32 xsh_rr(state); advance */
static void pcg32_fill(double *u, int64_t n, uint64_t seed) {
    const uint64_t mult = 0x5851F42D4C957F2Dull, inc = 0x14057B7EF767814Full;
    uint64_t state = (inc + seed) * mult + inc;
    for (int64_t i = 0; i < n; ++i) {
        const uint32_t xs = (uint32_t)(((state >> 18) ^ state) >> 27);
        const uint32_t rot = (uint32_t)(state >> 59);
        const uint32_t out = (xs >> rot) | (xs << ((32 - rot) & 31));
        u[i] = (double)out * 2.3283064365386962890625e-10;
        state = state * mult + inc;
    }
}

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 10000000;
    const int32_t m = argc > 2 ? atoi(argv[2]) : 20;
    const int steps = argc > 3 ? atoi(argv[3]) : 50;
    const int warmup = argc > 4 ? atoi(argv[4]) : 5;
    CHECK(dzo_init(0));
    char name[128];
    int32_t cus = 0;
    int64_t hbm = 0;
    CHECK(dzo_device_info(name, (int32_t)sizeof name, &cus, &hbm));

    double *h = (double *)malloc((size_t)n * sizeof(double));
    if (!h) return 1;
    pcg32_fill(h, n, 5);
    for (int64_t i = 0; i < n; ++i) h[i] = ((i % 2 == 0) ? -1.2 : 1.0) + 0.01 * (h[i] - 0.5);
    void *x_dev = NULL;
    CHECK(dzo_malloc(&x_dev, n * (int64_t)sizeof(double)));
    CHECK(dzo_memcpy_h2d(x_dev, h, n * (int64_t)sizeof(double)));
    free(h);

    dzo_problem_t prob = NULL;
    CHECK(dzo_problem_create(DZO_PROBLEM_ROSENBROCK_CHAIN, n, DZO_F64, NULL, NULL, 0.0, &prob));
    dzo_lbfgs_t opt = NULL;
    CHECK(dzo_lbfgs_create_problem(prob, m, x_dev, 1.0, &opt));     /* LBFGSOptimizer(nothing, f, g!, x0, 1.0, m) */

    for (int i = 0; i < m + warmup; ++i) CHECK(dzo_lbfgs_step(opt));
    double f_start = 0, f_end = 0;
    CHECK(dzo_lbfgs_get_s(opt, 0, &f_start));
    CHECK(dzo_synchronize());
    const double t0 = now_s();
    int64_t stuck = 0;
    int done = 0;
    for (; done < steps && !stuck; ++done) {
        CHECK(dzo_lbfgs_step(opt));                                 /* step!(opt) */
        CHECK(dzo_lbfgs_get_i(opt, 0, &stuck));                     /* opt.is_stuck[] */
    }
    CHECK(dzo_synchronize());
    const double el = now_s() - t0;
    CHECK(dzo_lbfgs_get_s(opt, 0, &f_end));
    int64_t iters = 0;
    CHECK(dzo_lbfgs_get_i(opt, 1, &iters));
    printf("{\"metric\": \"step!() calls/sec, L-BFGS n=%lld m=%d fp64, plain C host (examples/bench_lbfgs.c)\", \"value\": %.3f, "
           "\"unit\": \"step!() calls/s\", \"steps\": %d, \"warmup\": %d, \"ms_per_step\": %.4f, \"iteration_count\": %lld, "
           "\"f_start\": %.10e, \"f_end\": %.10e, \"stuck\": %lld, \"device\": \"%s\", \"compute_units\": %d}\n",
           (long long)n, m, (double)done / el, done, warmup, 1e3 * el / (done > 0 ? done : 1), (long long)iters, f_start, f_end,
           (long long)stuck, name, cus);
    CHECK(dzo_lbfgs_destroy(opt));
    CHECK(dzo_problem_destroy(prob));
    CHECK(dzo_free(x_dev));
    CHECK(dzo_shutdown());
    if (!(f_end < f_start) || done != steps) { printf("FAILED\n"); return 2; }
    printf("OK\n");
    return 0;
}
