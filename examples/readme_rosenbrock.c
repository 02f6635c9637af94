/* readme_rosenbrock.c -- the reference's README example (README.md:25-41) through the plain
 * C ABI of include/dzo.h, no Python and no PyTorch in the process:
 *
 *     opt = BFGSOptimizer(rosenbrock_objective, rosenbrock_gradient!, rand(2), 1.0)
 *     while !opt.has_converged[]; step!(opt); end
 *
 * The objective and the gradient are the CALLER's functions (host callbacks that read and write
 * device memory through dzo_memcpy_*), exactly as a Julia host would supply them; a second run
 * uses the library's built-in device objective and an L-BFGS optimizer on the chained problem, a
 * third the AdGD optimizer.
 *
 *   gcc -O2 -Iinclude examples/readme_rosenbrock.c -Ldzoptimization.jl_amd -ldzo_hip \
 *       -Wl,-rpath,$PWD/dzoptimization.jl_amd -o readme_rosenbrock && ./readme_rosenbrock
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "dzo.h"

#define CHECK(call)                                                                  \
    do {                                                                             \
        int32_t rc_ = (call);                                                        \
        if (rc_ != DZO_OK) {                                                         \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, dzo_last_error());   \
            return 1;                                                                \
        }                                                                            \
    } while (0)

/* README.md:25-31 */
static double rosenbrock_objective(void *ctx, const void *x_dev) {
    double x[2];
    (void)ctx;
    if (dzo_memcpy_d2h(x, x_dev, sizeof x) != DZO_OK) return 0.0 / 0.0;
    const double a = 1.0 - x[0], b = x[1] - x[0] * x[0];
    return a * a + 100.0 * b * b;
}

static void rosenbrock_gradient(void *ctx, void *g_dev, const void *x_dev) {
    double x[2], g[2];
    (void)ctx;
    if (dzo_memcpy_d2h(x, x_dev, sizeof x) != DZO_OK) return;
    const double b = x[1] - x[0] * x[0];
    g[0] = -2.0 * (1.0 - x[0]) - 400.0 * x[0] * b;
    g[1] = 200.0 * b;
    (void)dzo_memcpy_h2d(g_dev, g, sizeof g);
}

int main(void) {
    CHECK(dzo_init(0));

    /* ---- BFGSOptimizer(f, g!, x0, 1.0) with host callbacks ------------------------------ */
    double x0[2] = {0.25, 0.75};
    void *x0_dev = NULL;
    CHECK(dzo_malloc(&x0_dev, sizeof x0));
    CHECK(dzo_memcpy_h2d(x0_dev, x0, sizeof x0));
    dzo_bfgs_t opt = NULL;
    CHECK(dzo_bfgs_create_callbacks(rosenbrock_objective, rosenbrock_gradient, NULL, NULL, 2, DZO_F64, x0_dev, 1.0, &opt));
    int64_t done = 0, iters = 0;
    while (!done && iters < 10000) {                       /* while !opt.has_converged[] */
        CHECK(dzo_bfgs_step(opt));                         /* step!(opt) */
        CHECK(dzo_bfgs_get_i(opt, 0, &done));
        ++iters;
    }
    double f = 0, x[2];
    void *x_dev = NULL;
    CHECK(dzo_bfgs_get_s(opt, 0, &f));                     /* opt.current_objective_value[] */
    CHECK(dzo_bfgs_get_ptr(opt, 0, &x_dev));               /* opt.current_point */
    CHECK(dzo_memcpy_d2h(x, x_dev, sizeof x));
    printf("BFGS   2-D Rosenbrock (host callbacks): %lld steps, f = %.3e, x = (%.12f, %.12f)\n",
           (long long)iters, f, x[0], x[1]);
    const int ok1 = done && f < 1e-20 && x[0] > 0.999999 && x[0] < 1.000001 && x[1] > 0.999999 && x[1] < 1.000001;
    CHECK(dzo_bfgs_destroy(opt));
    CHECK(dzo_free(x0_dev));

    /* ---- LBFGSOptimizer(nothing, f, g!, x0, 1.0, 10) with the built-in device objective --- */
    const int64_t n = 1000;
    double *h = (double *)malloc((size_t)n * sizeof(double));
    for (int64_t i = 0; i < n; ++i) h[i] = (i % 2 == 0) ? -1.2 : 1.0;
    void *xd = NULL;
    CHECK(dzo_malloc(&xd, n * (int64_t)sizeof(double)));
    CHECK(dzo_memcpy_h2d(xd, h, n * (int64_t)sizeof(double)));
    dzo_problem_t prob = NULL;
    CHECK(dzo_problem_create(DZO_PROBLEM_ROSENBROCK_CHAIN, n, DZO_F64, NULL, NULL, 0.0, &prob));
    dzo_lbfgs_t lb = NULL;
    CHECK(dzo_lbfgs_create_problem(prob, 10, xd, 1.0, &lb)); /* aliases xd as current_point (:393) */
    int64_t stuck = 0;
    iters = 0;
    while (!stuck && iters < 100000) {
        CHECK(dzo_lbfgs_step(lb));
        CHECK(dzo_lbfgs_get_i(lb, 0, &stuck));             /* opt.is_stuck[] */
        ++iters;
    }
    CHECK(dzo_lbfgs_get_s(lb, 0, &f));
    CHECK(dzo_memcpy_d2h(h, xd, n * (int64_t)sizeof(double)));
    double worst = 0;
    for (int64_t i = 0; i < n; ++i) { const double e = h[i] > 1 ? h[i] - 1 : 1 - h[i]; if (e > worst) worst = e; }
    printf("L-BFGS chained Rosenbrock n=%lld (device objective): %lld steps, f = %.3e, max|x-1| = %.2e\n",
           (long long)n, (long long)iters, f, worst);
    const int ok2 = stuck && f < 1e-18 && worst < 1e-8;
    CHECK(dzo_lbfgs_destroy(lb));

    /* ---- AdGDOptimizer(nothing, f, g!, x0, 0.1) (src/DZOptimization.jl:245-251), same objective ---- */
    for (int64_t i = 0; i < n; ++i) h[i] = (i % 2 == 0) ? -1.2 : 1.0;
    CHECK(dzo_memcpy_h2d(xd, h, n * (int64_t)sizeof(double)));
    dzo_adgd_t ad = NULL;
    CHECK(dzo_adgd_create_problem(prob, xd, 0.1, &ad));
    double f_start = 0, f_prev = 0;
    CHECK(dzo_adgd_get_s(ad, 0, &f_start));
    f_prev = f_start;
    int monotone = 1;
    stuck = 0;
    for (iters = 0; iters < 2000 && !stuck; ++iters) {
        CHECK(dzo_adgd_step(ad));                          /* step!(opt) (:270-312) */
        CHECK(dzo_adgd_get_i(ad, 0, &stuck));
        CHECK(dzo_adgd_get_s(ad, 0, &f));
        if (!stuck && !(f < f_prev)) monotone = 0;         /* :139: accepted steps strictly decrease f */
        f_prev = f;
    }
    printf("AdGD   chained Rosenbrock n=%lld (device objective): %lld steps, f = %.6e -> %.6e\n",
           (long long)n, (long long)iters, f_start, f);
    const int ok3 = monotone && f < 0.5 * f_start;
    CHECK(dzo_adgd_destroy(ad));
    CHECK(dzo_problem_destroy(prob));
    CHECK(dzo_free(xd));
    free(h);
    CHECK(dzo_shutdown());
    if (!ok3) { fprintf(stderr, "FAILED: AdGD did not decrease the objective monotonically\n"); return 3; }
    if (!ok1 || !ok2) { fprintf(stderr, "FAILED: did not converge to (1, ..., 1)\n"); return 2; }
    printf("OK\n");
    return 0;
}
