/*
 * dzo_oracle_impl.h -- type-generic body of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 *
 * Included twice by dzo_oracle.c with
 *     #define T double / float      (element type)
 *     #define ACC long double / double   (wide accumulator for "truth-like" reductions)
 *     #define SUF _f64 / _f32
 *
 * Every function cites the reference lines it restates.  Paths are relative to the
 * upstream snapshot (dzhang314/DZOptimization.jl @ 2025-09-05):
 *     src/DZOptimization.jl      -- live v0.6.0 code (normative for L-BFGS)
 *     legacy/DZOptimization.jl   -- commented-out spec (normative for dense BFGS)
 *     legacy/Kernels.jl          -- sequential-order primitive definitions
 *
 * PARITY UNPINNED BY THE REFERENCE: upstream has no tests, no golden vectors, and Julia is
 * not installed in the build container, so this restatement is pinned by analytic identities,
 * an mpmath twin (oracle/mp_twoloop.py) and the legacy run_and_test! invariants instead.
 *
 * Compile with -ffp-contract=off: every fused multiply-add below is written explicitly so
 * the elementwise arithmetic is the same on CPU and GPU.
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

#if ORC_IS_F64
#define T_FMA(a, b, c) fma((a), (b), (c))
#define T_SQRT(a) sqrt(a)
#define T_MAXVAL DBL_MAX
#define T_ISFINITE(a) isfinite(a)
#else
#define T_FMA(a, b, c) fmaf((a), (b), (c))
#define T_SQRT(a) sqrtf(a)
#define T_MAXVAL FLT_MAX
#define T_ISFINITE(a) isfinite(a)
#endif

/* ------------------------------------------------------------------------------------------
 * L1 primitives.  legacy/Kernels.jl:12-20 (dot), :49-55 (norm2), :76-135 (negate/scale/
 * delta/axpy); live call sites src/DZOptimization.jl:118,124,128,145,151,381-387,438-449,
 * 478-480,505 go through LinearAlgebra (BLAS order unknowable).
 * ---------------------------------------------------------------------------------------- */

/* Reduction order is selectable because the live reference's order (BLAS ddot) is not
 * written down anywhere: 0 = sequential (legacy/Kernels.jl:12-20, the only order the
 * reference states), 1 = 8-lane strided partial sums (what a SIMD BLAS kernel does),
 * 2 = wide accumulator (closest to the exact value; arbiter). */
T FN(orc_dot)(const T *v, const T *w, int64_t n) {
    if (orc_dot_mode == 2) {
        ACC acc = 0;
        if (orc_threads > 1) {
#pragma omp parallel for reduction(+ : acc) num_threads(orc_threads) schedule(static)
            for (int64_t i = 0; i < n; ++i) acc += (ACC)v[i] * (ACC)w[i];
        } else {
            for (int64_t i = 0; i < n; ++i) acc += (ACC)v[i] * (ACC)w[i];
        }
        return (T)acc;
    }
    if (orc_dot_mode == 1) {
        T lane[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int64_t i = 0;
        for (; i + 8 <= n; i += 8)
            for (int l = 0; l < 8; ++l) lane[l] = T_FMA(v[i + l], w[i + l], lane[l]);
        T tail = 0;
        for (; i < n; ++i) tail = T_FMA(v[i], w[i], tail);
        T s01 = lane[0] + lane[1], s23 = lane[2] + lane[3];
        T s45 = lane[4] + lane[5], s67 = lane[6] + lane[7];
        return ((s01 + s23) + (s45 + s67)) + tail;
    }
    if (orc_threads > 1) {
        /* timing-only variant: chunked sequential sums, combined in thread order */
        T result = 0;
#pragma omp parallel for reduction(+ : result) num_threads(orc_threads) schedule(static)
        for (int64_t i = 0; i < n; ++i) result += v[i] * w[i];
        return result;
    }
    T result = 0;
    for (int64_t i = 0; i < n; ++i) result += v[i] * w[i]; /* Kernels.jl:17 */
    return result;
}

/* legacy/Kernels.jl:49-55 -- sum of squares, NOT its square root. */
T FN(orc_norm2)(const T *x, int64_t n) { return FN(orc_dot)(x, x, n); }

/* LinearAlgebra.norm as used at src/DZOptimization.jl:230,381 and legacy :921,928. */
T FN(orc_norm)(const T *x, int64_t n) { return T_SQRT(FN(orc_norm2)(x, n)); }

/* y += a*x   (LinearAlgebra.axpy!, legacy/Kernels.jl:118-125); one fused rounding. */
void FN(orc_axpy)(T a, const T *x, T *y, int64_t n) {
#pragma omp parallel for num_threads(orc_threads) schedule(static) if (orc_threads > 1)
    for (int64_t i = 0; i < n; ++i) y[i] = T_FMA(a, x[i], y[i]);
}

/* dst = a*x + y   (out-of-place trial point, legacy/Kernels.jl:127-135). */
void FN(orc_axpy_oop)(T *dst, T a, const T *x, const T *y, int64_t n) {
#pragma omp parallel for num_threads(orc_threads) schedule(static) if (orc_threads > 1)
    for (int64_t i = 0; i < n; ++i) dst[i] = T_FMA(a, x[i], y[i]);
}

/* y = a*x + b*y   (LinearAlgebra.axpby!; the reference only ever calls it with a=1,b=-1,
 * src/DZOptimization.jl:145,308,480, where the result is exactly x - y). */
void FN(orc_axpby)(T a, const T *x, T b, T *y, int64_t n) {
#pragma omp parallel for num_threads(orc_threads) schedule(static) if (orc_threads > 1)
    for (int64_t i = 0; i < n; ++i) y[i] = T_FMA(a, x[i], b * y[i]);
}

/* x *= a   (LinearAlgebra.rmul!, legacy/Kernels.jl:87-94). */
void FN(orc_scal)(T *x, T a, int64_t n) {
#pragma omp parallel for num_threads(orc_threads) schedule(static) if (orc_threads > 1)
    for (int64_t i = 0; i < n; ++i) x[i] *= a;
}

void FN(orc_copy)(T *dst, const T *src, int64_t n) { memcpy(dst, src, (size_t)n * sizeof(T)); }

void FN(orc_fill)(T *x, T a, int64_t n) {
    for (int64_t i = 0; i < n; ++i) x[i] = a;
}

/* Base.isequal on arrays (src/DZOptimization.jl:128): NaN equals NaN, -0.0 differs from +0.0. */
int FN(orc_isequal)(const T *a, const T *b, int64_t n) {
    for (int64_t i = 0; i < n; ++i) {
        if (isnan(a[i]) && isnan(b[i])) continue;
        if (memcmp(&a[i], &b[i], sizeof(T)) != 0) return 0;
    }
    return 1;
}

/* ------------------------------------------------------------------------------------------
 * Synthetic problems (user callbacks in the reference; SURVEY.md section 8(d) defines them).
 * ---------------------------------------------------------------------------------------- */

typedef struct {
    int32_t kind;    /* ORC_PROBLEM_* */
    int64_t n;
    const T *A;      /* QUADRATIC: dense n x n, column-major */
    const T *c;      /* LSE: centre of the quadratic term */
    T lambda;        /* LSE: weight of the quadratic term */
    /* decorators of legacy/DZOptimization.jl:219-296 ("next" row 8(f).3) */
    T l2;            /* L2RegularizationWrapper / L2GradientWrapper lambda (:225-249); 0 = off */
    int32_t bg_on;   /* UniformBoxGradientWrapper (:275-296) */
    T bg_lo, bg_hi;
    int32_t cons_on; /* UniformBoxConstraint (:258-272) used as constraint_function! */
    T cons_lo, cons_hi;
} FN(orc_problem);

/* Elementwise term formulas are written once, with explicit fma, and the HIP kernels in
 * dzoptimization.jl_amd/csrc/dzo_problems.hip use the same expressions. */
static inline ACC FN(rosen_term)(T xi, T xn) {
    T t1 = (T)1 - xi;
    T t2 = T_FMA(-xi, xi, xn);                 /* x_{i+1} - x_i^2 */
    return (ACC)T_FMA((T)100 * t2, t2, t1 * t1);
}

/* Chained quadratic (build-defined synthetic objective, the large-n member of north_star's "synthetic quadratic"
 * problems; dzoptimization.jl_amd/csrc/dzo_rosen.h uses the same expressions):
 *   f = sum_{i<n-1} 1/2 (x[i+1]-x[i])^2 + sum_{i<n} lambda/2 (x[i]-1)^2
 * index tests as 0 / 1 coefficients, so that every element runs the same operations in the same order */
static inline ACC FN(qchain_term)(int64_t i, int64_t n, T lambda, T xi, T xn) {
    const T hk = i + 1 < n ? (T)0.5 : (T)0, hm = (T)0.5 * lambda;
    const T d = xi - (T)1;
    const T pr = xn - xi;
    return (ACC)T_FMA(hk * pr, pr, (hm * d) * d);
}
static inline T FN(qchain_grad)(int64_t i, int64_t n, T lambda, T xp, T xi, T xn) {
    const T cR = i + 1 < n ? (T)1 : (T)0, cL = i > 0 ? (T)1 : (T)0;
    return T_FMA(lambda, xi - (T)1, T_FMA(cL, xi - xp, cR * (xi - xn)));
}

static T FN(problem_eval_base)(const FN(orc_problem) *p, const T *x) {
    const int64_t n = p->n;
    switch (p->kind) {
    case ORC_PROBLEM_ROSENBROCK2D: {
        /* legacy/ExampleFunctions.jl:10-15 */
        T t1 = (T)1 - x[0];
        T t2 = x[1] - x[0] * x[0];
        return t1 * t1 + (T)100 * (t2 * t2);
    }
    case ORC_PROBLEM_ROSENBROCK_CHAIN: {
        ACC acc = 0;
#pragma omp parallel for reduction(+ : acc) num_threads(orc_threads) schedule(static) if (orc_threads > 1)
        for (int64_t i = 0; i < n - 1; ++i) acc += FN(rosen_term)(x[i], x[i + 1]);
        return (T)acc;
    }
    case ORC_PROBLEM_QUADRATIC: {
        /* f = 1/2 x' A x, A column-major and symmetric: column j contributes x_j * (A[:,j].x) */
        ACC acc = 0;
#pragma omp parallel for reduction(+ : acc) num_threads(orc_threads) schedule(static) if (orc_threads > 1)
        for (int64_t j = 0; j < n; ++j) {
            ACC col = 0;
            const T *a = p->A + j * n;
            for (int64_t i = 0; i < n; ++i) col += (ACC)a[i] * (ACC)x[i];
            acc += col * (ACC)x[j];
        }
        return (T)((ACC)0.5 * acc);
    }
    case ORC_PROBLEM_QUADRATIC_CHAIN: {
        ACC acc = 0;
#pragma omp parallel for reduction(+ : acc) num_threads(orc_threads) schedule(static) if (orc_threads > 1)
        for (int64_t i = 0; i < n; ++i) acc += FN(qchain_term)(i, n, p->lambda, x[i], i + 1 < n ? x[i + 1] : (T)0);
        return (T)acc;
    }
    case ORC_PROBLEM_LSE: {
        /* f = log sum exp(x_i) + lambda/2 |x - c|^2, max-subtracted */
        T mx = x[0];
        for (int64_t i = 1; i < n; ++i) mx = x[i] > mx ? x[i] : mx;
        ACC se = 0, sq = 0;
        for (int64_t i = 0; i < n; ++i) {
            se += (ACC)exp((double)(x[i] - mx));
            T dlt = x[i] - p->c[i];
            sq += (ACC)dlt * (ACC)dlt;
        }
        return (T)((ACC)mx + (ACC)log((double)se) + (ACC)0.5 * (ACC)p->lambda * sq);
    }
    }
    return (T)NAN;
}

static void FN(problem_grad_base)(const FN(orc_problem) *p, T *g, const T *x) {
    const int64_t n = p->n;
    switch (p->kind) {
    case ORC_PROBLEM_ROSENBROCK2D: {
        /* legacy/ExampleFunctions.jl:17-24 */
        T t1 = (T)1 - x[0];
        T t2 = x[1] - x[0] * x[0];
        g[0] = (T)-2 * t1 - (T)400 * x[0] * t2;
        g[1] = (T)200 * t2;
        return;
    }
    case ORC_PROBLEM_ROSENBROCK_CHAIN: {
#pragma omp parallel for num_threads(orc_threads) schedule(static) if (orc_threads > 1)
        for (int64_t i = 0; i < n; ++i) {
            T gi = 0;
            if (i + 1 < n) {
                T t2 = T_FMA(-x[i], x[i], x[i + 1]);
                T t1 = (T)1 - x[i];
                gi = T_FMA((T)-400 * x[i], t2, (T)-2 * t1);
            }
            if (i > 0) {
                T t2p = T_FMA(-x[i - 1], x[i - 1], x[i]);
                gi = T_FMA((T)200, t2p, gi);
            }
            g[i] = gi;
        }
        return;
    }
    case ORC_PROBLEM_QUADRATIC: {
        /* g = A x via symmetric column dots: g_j = A[:,j] . x */
#pragma omp parallel for num_threads(orc_threads) schedule(static) if (orc_threads > 1)
        for (int64_t j = 0; j < n; ++j) {
            ACC col = 0;
            const T *a = p->A + j * n;
            for (int64_t i = 0; i < n; ++i) col += (ACC)a[i] * (ACC)x[i];
            g[j] = (T)col;
        }
        return;
    }
    case ORC_PROBLEM_QUADRATIC_CHAIN: {
#pragma omp parallel for num_threads(orc_threads) schedule(static) if (orc_threads > 1)
        for (int64_t i = 0; i < n; ++i)
            g[i] = FN(qchain_grad)(i, n, p->lambda, i > 0 ? x[i - 1] : (T)0, x[i], i + 1 < n ? x[i + 1] : (T)0);
        return;
    }
    case ORC_PROBLEM_LSE: {
        T mx = x[0];
        for (int64_t i = 1; i < n; ++i) mx = x[i] > mx ? x[i] : mx;
        ACC se = 0;
        for (int64_t i = 0; i < n; ++i) se += (ACC)exp((double)(x[i] - mx));
        for (int64_t i = 0; i < n; ++i) {
            double sm = exp((double)(x[i] - mx)) / (double)se;
            g[i] = (T)(sm + (double)p->lambda * (double)(x[i] - p->c[i]));
        }
        return;
    }
    }
}

/* L2RegularizationWrapper call: objective(x) + lambda * norm2(x)  (legacy/DZOptimization.jl:231-232) */
T FN(orc_problem_eval)(const FN(orc_problem) *p, const T *x) {
    T f = FN(problem_eval_base)(p, x);
    if (p->l2 != (T)0) f = f + p->l2 * FN(orc_norm2)(x, p->n);
    return f;
}

/* L2GradientWrapper (:241-249): g += (lambda + lambda) * x, then UniformBoxGradientWrapper
 * (:282-296): zero the components that push against an active bound. */
void FN(orc_problem_grad)(const FN(orc_problem) *p, T *g, const T *x) {
    FN(problem_grad_base)(p, g, x);
    if (p->l2 != (T)0) FN(orc_axpy)(p->l2 + p->l2, x, g, p->n);          /* :247 */
    if (p->bg_on) {
        for (int64_t i = 0; i < p->n; ++i) {                               /* :289-294 */
            if ((x[i] <= p->bg_lo && g[i] >= (T)0) || (x[i] >= p->bg_hi && g[i] <= (T)0)) g[i] = (T)0;
        }
    }
}

/* UniformBoxConstraint call (:264-272): clamp in place, always feasible. */
void FN(orc_box_clamp)(T *x, T lo, T hi, int64_t n) {
    for (int64_t i = 0; i < n; ++i) x[i] = x[i] < lo ? lo : (x[i] > hi ? hi : x[i]);
}

/* Callback triple in the reference's order (src/DZOptimization.jl:323-325). A NULL
 * constraint is the reference's `nothing` (:71,134,412). */
typedef T (*FN(orc_objective_fn))(void *ctx, const T *x, int64_t n);
typedef void (*FN(orc_gradient_fn))(void *ctx, T *g, const T *x, int64_t n);
typedef int (*FN(orc_constraint_fn))(void *ctx, T *x, int64_t n);

static T FN(problem_obj_cb)(void *ctx, const T *x, int64_t n) {
    (void)n;
    return FN(orc_problem_eval)((const FN(orc_problem) *)ctx, x);
}
static void FN(problem_grad_cb)(void *ctx, T *g, const T *x, int64_t n) {
    (void)n;
    FN(orc_problem_grad)((const FN(orc_problem) *)ctx, g, x);
}
static int FN(problem_constraint_cb)(void *ctx, T *x, int64_t n) {
    const FN(orc_problem) *p = (const FN(orc_problem) *)ctx;
    FN(orc_box_clamp)(x, p->cons_lo, p->cons_hi, n);
    return 1;
}

/* ------------------------------------------------------------------------------------------
 * L-BFGS  (src/DZOptimization.jl:321-509)
 * ---------------------------------------------------------------------------------------- */

typedef struct {
    FN(orc_constraint_fn) constraint; /* :323 */
    FN(orc_objective_fn) objective;   /* :324 */
    FN(orc_gradient_fn) gradient;     /* :325 */
    void *ctx;
    int32_t is_stuck;                 /* :327 */
    int64_t iteration_count;          /* :328 */
    int64_t n;
    T *x;                             /* :330 current_point (ALIASES the caller's array, :393) */
    T *dx;                            /* :331 delta_point */
    T f;                              /* :332 */
    T df;                             /* :333 */
    T *g;                             /* :334 current_gradient (aliased, :395) */
    T *dg;                            /* :335 delta_gradient */
    T *d;                             /* :337 step_direction */
    int32_t m;                        /* :338 history_length */
    int32_t k;                        /* current length of the histories */
    T **S;                            /* :339 delta_point_history, index 0 = newest */
    T **Y;                            /* :340 delta_gradient_history */
    T *alpha;                         /* :341 */
    T *rho;                           /* :342  stores s.y itself, not its inverse (:505) */
    int32_t n_alpha;
    int32_t n_rho;
    int64_t max_halvings;             /* build-added escape from the NaN loop */
    int64_t last_trials;              /* number of objective evaluations in the last step */
    /* Optional safeguards (all off by default = the live reference).  SURVEY.md 8(f) rows 2, 4:
     * the legacy optimizer's descent check and steepest-descent fallback with history reset
     * (legacy/DZOptimization.jl:588-610, :682-692) and a strong-Wolfe search driven by the
     * LineSearchEvaluator quotients (src/DZOptimization.jl:65-92). */
    int32_t descent_check;
    int32_t sd_fallback;
    int32_t line_search;              /* 0 = take_backtracking_step!, 1 = strong Wolfe */
    T wolfe_c1, wolfe_c2;
    int32_t wolfe_max_evals;
    T last_step_length;               /* legacy :625-627 sqrt(norm2(delta_point)) */
    int64_t history_resets;
    int64_t descent_resets;
    int32_t last_step_kind;           /* 0 quasi-Newton, 1 descent-check replacement, 2 fallback */
    T *xt, *gt;                       /* LineSearchEvaluator trial_point / trial_gradient (:26-27) */
} FN(orc_lbfgs);

/* compute_lbfgs_step_direction!  src/DZOptimization.jl:430-451 */
void FN(orc_lbfgs_direction)(T *d, const T *g, T *const *S, T *const *Y, T *alpha, const T *rho,
                             int32_t k, int64_t n) {
    FN(orc_copy)(d, g, n);                                   /* :438 */
    for (int32_t i = 0; i < k; ++i) {                        /* :439 newest -> oldest */
        alpha[i] = FN(orc_dot)(S[i], d, n) / rho[i];         /* :440 */
        FN(orc_axpy)(-alpha[i], Y[i], d, n);                 /* :441 */
    }
    if (k > 0) {                                             /* :443 */
        FN(orc_scal)(d, -rho[0] / FN(orc_dot)(Y[0], Y[0], n), n); /* :444 */
    }
    for (int32_t i = k - 1; i >= 0; --i) {                   /* :446 oldest -> newest */
        T beta = FN(orc_dot)(Y[i], d, n) / rho[i];           /* :447 */
        FN(orc_axpy)(-(alpha[i] + beta), S[i], d, n);        /* :448 */
    }
}

/* Full constructor  src/DZOptimization.jl:347-397 (x0 and g0 are aliased, not copied). */
FN(orc_lbfgs) *FN(orc_lbfgs_create_full)(FN(orc_constraint_fn) cf, FN(orc_objective_fn) of,
                                         FN(orc_gradient_fn) gf, void *ctx, T *x0, T f0, T *g0,
                                         T initial_step_length, int32_t m, int64_t n) {
    if (!(initial_step_length > (T)0)) return NULL;         /* :380 @assert */
    FN(orc_lbfgs) *o = (FN(orc_lbfgs) *)calloc(1, sizeof(*o));
    o->constraint = cf; o->objective = of; o->gradient = gf; o->ctx = ctx;
    o->n = n; o->m = m;
    o->x = x0; o->g = g0; o->f = f0; o->df = 0;
    o->dx = (T *)calloc((size_t)n, sizeof(T));               /* :366-369 */
    o->dg = (T *)calloc((size_t)n, sizeof(T));               /* :371-374 */
    o->d = (T *)malloc((size_t)n * sizeof(T));               /* :376 */
    T gnorm = FN(orc_norm)(g0, n);                           /* :381 */
    o->is_stuck = (gnorm == (T)0);                           /* :382 */
    if (o->is_stuck) {
        FN(orc_fill)(o->d, (T)0, n);                         /* :384 */
    } else {
        FN(orc_copy)(o->d, g0, n);                           /* :386 */
        FN(orc_scal)(o->d, -initial_step_length / gnorm, n); /* :387 */
    }
    o->S = (T **)calloc((size_t)(m > 0 ? m : 1), sizeof(T *)); /* :396 empty histories */
    o->Y = (T **)calloc((size_t)(m > 0 ? m : 1), sizeof(T *));
    o->alpha = (T *)calloc((size_t)(m > 0 ? m : 1), sizeof(T));
    o->rho = (T *)calloc((size_t)(m > 0 ? m : 1), sizeof(T));
    o->k = 0; o->n_alpha = 0; o->n_rho = 0;
    o->max_halvings = 4096;
    o->wolfe_c1 = (T)1e-4; o->wolfe_c2 = (T)0.9; o->wolfe_max_evals = 40;
    o->last_step_length = initial_step_length;               /* as legacy BFGS :779 */
    return o;
}

/* Convenience constructor  src/DZOptimization.jl:400-427 */
FN(orc_lbfgs) *FN(orc_lbfgs_create)(FN(orc_constraint_fn) cf, FN(orc_objective_fn) of,
                                    FN(orc_gradient_fn) gf, void *ctx, T *x0, T *g0_storage,
                                    T initial_step_length, int32_t m, int64_t n) {
    if (cf && !cf(ctx, x0, n)) return NULL;                  /* :412-414 */
    T f0 = of(ctx, x0, n);                                   /* :416 */
    gf(ctx, g0_storage, x0, n);                              /* :418-421 */
    return FN(orc_lbfgs_create_full)(cf, of, gf, ctx, x0, f0, g0_storage, initial_step_length, m, n);
}

FN(orc_lbfgs) *FN(orc_lbfgs_create_problem)(const FN(orc_problem) *p, T *x0, T *g0_storage,
                                            T initial_step_length, int32_t m) {
    return FN(orc_lbfgs_create)(p->cons_on ? FN(problem_constraint_cb) : NULL, FN(problem_obj_cb),
                                FN(problem_grad_cb), (void *)p, x0, g0_storage, initial_step_length, m, p->n);
}

void FN(orc_lbfgs_destroy)(FN(orc_lbfgs) *o) {
    if (!o) return;
    for (int32_t i = 0; i < o->k; ++i) { free(o->S[i]); free(o->Y[i]); }
    free(o->S); free(o->Y); free(o->alpha); free(o->rho);
    free(o->dx); free(o->dg); free(o->d); free(o->xt); free(o->gt);
    free(o);
}

/* take_backtracking_step!  src/DZOptimization.jl:107-154 (shared with AdGD). Operates on
 * the five fields it touches so both optimizers can call it. */
static void FN(backtracking_step)(FN(orc_constraint_fn) cf, FN(orc_objective_fn) of, void *ctx,
                                  int64_t n, T *x, T *dx, T *f, T *df, int32_t *is_stuck,
                                  T step_size, const T *dir, int64_t max_halvings,
                                  int64_t *trials) {
    const T half = (T)1 / ((T)1 + (T)1);                     /* :113-115 */
    FN(orc_copy)(dx, x, n);                                  /* :118 */
    int64_t halvings = 0;
    *trials = 0;
    for (;;) {                                               /* :121 */
        FN(orc_axpy)(step_size, dir, x, n);                  /* :124 */
        if (FN(orc_isequal)(x, dx, n)) {                     /* :128 */
            *is_stuck = 1;                                   /* :129 */
            return;
        }
        if (!cf || cf(ctx, x, n)) {                          /* :134-135 */
            T f_new = of(ctx, x, n);                         /* :138 */
            ++*trials;
            if (f_new < *f) {                                /* :139 strict decrease only */
                *df = f_new - *f;                            /* :142-143 */
                *f = f_new;                                  /* :144 */
                FN(orc_axpby)((T)1, x, (T)-1, dx, n);        /* :145 dx = x_new - x_old */
                return;
            }
        }
        FN(orc_copy)(x, dx, n);                              /* :151 exact restore */
        step_size *= half;                                   /* :152 */
        if (max_halvings > 0 && ++halvings >= max_halvings) {
            /* Not in the reference: a NaN direction never satisfies :128 or :139 and would
             * loop forever. Declare the optimizer stuck instead. */
            *is_stuck = 1;
            return;
        }
    }
}

/* LineSearchEvaluator call (src/DZOptimization.jl:65-92) on the optimizer's own buffers. */
static void FN(lbfgs_ls_eval)(FN(orc_lbfgs) *o, const T *dir, T overlap, T t, T *f_t, T *ir, T *sr) {
    const int64_t n = o->n;
    FN(orc_copy)(o->xt, o->x, n);                            /* :69 */
    FN(orc_axpy)(t, dir, o->xt, n);                          /* :70 */
    if (o->constraint && !o->constraint(o->ctx, o->xt, n)) { /* :71-79 */
        *f_t = T_MAXVAL; *ir = -T_MAXVAL; *sr = T_MAXVAL;
        return;
    }
    *f_t = o->objective(o->ctx, o->xt, n);                   /* :80-81 */
    *ir = (*f_t - o->f) / (t * overlap);                     /* :84 */
    o->gradient(o->ctx, o->gt, o->xt, n);                    /* :87 */
    *sr = FN(orc_dot)(o->gt, dir, n) / overlap;              /* :88-89 */
}

/* Strong-Wolfe search by bisection / doubling on the evaluator's quotients:
 *   improvement_ratio >= c1  (Armijo),  |slope_ratio| <= c2  (curvature).
 * On success x, f, df, dx = x_new - x_old, g, dg = g_new - g_old are all updated (the trial
 * gradient is reused) and 1 is returned; otherwise nothing moves and is_stuck is set. */
static int FN(lbfgs_wolfe_search)(FN(orc_lbfgs) *o, const T *dir) {
    const int64_t n = o->n;
    if (!o->xt) { o->xt = (T *)malloc((size_t)n * sizeof(T)); o->gt = (T *)malloc((size_t)n * sizeof(T)); }
    o->last_trials = 0;
    const T overlap = FN(orc_dot)(o->g, dir, n);             /* g.d, the evaluator's `overlap` */
    if (!(overlap < (T)0)) { o->is_stuck = 1; return 0; }    /* not a descent direction (or NaN) */
    const T half = (T)1 / ((T)1 + (T)1);
    T t = (T)1, lo = (T)0, hi = (T)-1;                       /* hi < 0: no upper bound yet */
    for (int32_t it = 0; it < o->wolfe_max_evals; ++it) {
        T f_t, ir, sr;
        FN(lbfgs_ls_eval)(o, dir, overlap, t, &f_t, &ir, &sr);
        o->last_trials += 1;
        if (!(ir >= o->wolfe_c1) || !(f_t < o->f)) hi = t;   /* Armijo fails (NaN counts as failure) */
        else if (sr > o->wolfe_c2) lo = t;                   /* still descending steeply: move right */
        else if (sr < -o->wolfe_c2) hi = t;                  /* overshot the minimiser */
        else {
            FN(orc_copy)(o->dx, o->xt, n);
            FN(orc_axpby)((T)-1, o->x, (T)1, o->dx, n);      /* dx = x_new - x_old */
            FN(orc_copy)(o->x, o->xt, n);
            o->df = f_t - o->f;
            o->f = f_t;
            FN(orc_copy)(o->dg, o->gt, n);
            FN(orc_axpby)((T)-1, o->g, (T)1, o->dg, n);      /* dg = g_new - g_old */
            FN(orc_copy)(o->g, o->gt, n);
            return 1;
        }
        const T t_next = hi < (T)0 ? t + t : (lo + hi) * half;
        if (t_next == lo || t_next == hi || !(t_next > (T)0)) break;   /* interval exhausted */
        t = t_next;
    }
    o->is_stuck = 1;
    return 0;
}

/* one line search along `dir`; returns 1 when the gradient (and dg) is already up to date */
static int FN(lbfgs_search)(FN(orc_lbfgs) *o, const T *dir) {
    if (o->line_search == 1) return FN(lbfgs_wolfe_search)(o, dir);
    FN(backtracking_step)(o->constraint, o->objective, o->ctx, o->n, o->x, o->dx, &o->f, &o->df,
                          &o->is_stuck, (T)1, dir, o->max_halvings, &o->last_trials); /* :473 */
    return 0;
}

/* d = -(last_step_length / ||g||) g   (legacy :594-596, :688-690) */
static void FN(lbfgs_steepest)(FN(orc_lbfgs) *o) {
    const T inv = (T)1 / T_SQRT(FN(orc_norm2)(o->g, o->n));  /* Kernels.jl:141 */
    FN(orc_copy)(o->d, o->g, o->n);
    FN(orc_scal)(o->d, -o->last_step_length * inv, o->n);
}

/* step!(::LBFGSOptimizer)  src/DZOptimization.jl:454-509 */
void FN(orc_lbfgs_step)(FN(orc_lbfgs) *o) {
    if (o->is_stuck) return;                                 /* :456-458 */
    const int64_t n = o->n;
    int quasi = 0;
    o->last_step_kind = 0;
    if (o->iteration_count > 0) {                            /* :463 */
        FN(orc_lbfgs_direction)(o->d, o->g, o->S, o->Y, o->alpha, o->rho, o->k, n);
        quasi = o->k > 0;
        if (o->descent_check) {                              /* legacy :682-692 */
            const T gd = FN(orc_dot)(o->d, o->g, n);
            if (!T_ISFINITE(gd)) { o->is_stuck = 1; return; }
            if (gd >= (T)0) {
                FN(lbfgs_steepest)(o);
                o->descent_resets += 1;
                o->last_step_kind = 1;
                quasi = 0;
            }
        }
    }
    int have_gradient = FN(lbfgs_search)(o, o->d);
    if (o->is_stuck && o->sd_fallback && quasi) {            /* legacy :588-610 */
        o->is_stuck = 0;
        FN(lbfgs_steepest)(o);
        o->last_step_kind = 2;
        have_gradient = FN(lbfgs_search)(o, o->d);
        if (o->is_stuck) return;
        for (int32_t i = 0; i < o->k; ++i) { free(o->S[i]); free(o->Y[i]); o->S[i] = NULL; o->Y[i] = NULL; }
        o->k = 0; o->n_rho = 0;                              /* legacy :609 _history_count[] = 0 */
        o->history_resets += 1;
    }
    if (o->is_stuck) return;                                 /* :474-476 */
    if (o->descent_check || o->sd_fallback)
        o->last_step_length = T_SQRT(FN(orc_norm2)(o->dx, n));   /* legacy :625-627 */

    if (!have_gradient) {
    FN(orc_copy)(o->dg, o->g, n);                            /* :478 */
    o->gradient(o->ctx, o->g, o->x, n);                      /* :479 */
    FN(orc_axpby)((T)1, o->g, (T)-1, o->dg, n);              /* :480 */
    }

    if (o->m > 0) {
        /* :482-496  pushfirst! a copy, recycling the oldest buffer once full */
        T *s_buf, *y_buf;
        if (o->k < o->m) {
            s_buf = (T *)malloc((size_t)n * sizeof(T));
            y_buf = (T *)malloc((size_t)n * sizeof(T));
            o->k += 1;
        } else {
            s_buf = o->S[o->k - 1];
            y_buf = o->Y[o->k - 1];
        }
        memmove(o->S + 1, o->S, (size_t)(o->k - 1) * sizeof(T *));
        memmove(o->Y + 1, o->Y, (size_t)(o->k - 1) * sizeof(T *));
        FN(orc_copy)(s_buf, o->dx, n);
        FN(orc_copy)(y_buf, o->dg, n);
        o->S[0] = s_buf;
        o->Y[0] = y_buf;
        if (o->n_alpha < o->m) o->n_alpha += 1;              /* :498-500 */
        if (o->n_rho >= o->m) o->n_rho -= 1;                 /* :502-504 pop! */
        memmove(o->rho + 1, o->rho, (size_t)o->n_rho * sizeof(T));
        o->rho[0] = FN(orc_dot)(o->dx, o->dg, n);            /* :505 */
        o->n_rho += 1;
    }
    /* m == 0: the reference would grow unbounded-length-0 histories incorrectly (it
     * pushes then never pops S/Y when history_length == 0); the build requires m >= 1 on
     * the GPU path and keeps m == 0 here as plain steepest descent with the initial d. */
    o->iteration_count += 1;                                 /* :507 */
}

/* ------------------------------------------------------------------------------------------
 * AdGD  (src/DZOptimization.jl:179-312) -- "next" row 8(f).1; shares every primitive.
 * ---------------------------------------------------------------------------------------- */

typedef struct {
    FN(orc_constraint_fn) constraint;
    FN(orc_objective_fn) objective;
    FN(orc_gradient_fn) gradient;
    void *ctx;
    int32_t is_stuck;
    int64_t iteration_count;
    int64_t n;
    T *x, *dx, *g, *dg;
    T f, df;
    T current_step_size, previous_step_size;                 /* :195-196 */
    int64_t max_halvings;
    int64_t last_trials;
} FN(orc_adgd);

/* src/DZOptimization.jl:201-242 */
FN(orc_adgd) *FN(orc_adgd_create_full)(FN(orc_constraint_fn) cf, FN(orc_objective_fn) of,
                                       FN(orc_gradient_fn) gf, void *ctx, T *x0, T f0, T *g0,
                                       T initial_step_length, int64_t n) {
    if (!(initial_step_length > (T)0)) return NULL;         /* :229 */
    FN(orc_adgd) *o = (FN(orc_adgd) *)calloc(1, sizeof(*o));
    o->constraint = cf; o->objective = of; o->gradient = gf; o->ctx = ctx;
    o->n = n; o->x = x0; o->g = g0; o->f = f0; o->df = 0;
    o->dx = (T *)calloc((size_t)n, sizeof(T));
    o->dg = (T *)calloc((size_t)n, sizeof(T));
    T gnorm = FN(orc_norm)(g0, n);                           /* :230 */
    o->is_stuck = (gnorm == (T)0);                           /* :231 */
    T s0 = o->is_stuck ? (T)0 : initial_step_length / gnorm; /* :232-233 */
    o->current_step_size = s0;
    o->previous_step_size = s0;                              /* :241 */
    o->max_halvings = 4096;
    return o;
}

FN(orc_adgd) *FN(orc_adgd_create_problem)(const FN(orc_problem) *p, T *x0, T *g0_storage,
                                          T initial_step_length) {
    if (p->cons_on) FN(orc_box_clamp)(x0, p->cons_lo, p->cons_hi, p->n);   /* :256-258 */
    T f0 = FN(orc_problem_eval)(p, x0);                      /* :260 */
    FN(orc_problem_grad)(p, g0_storage, x0);                 /* :262-265 */
    return FN(orc_adgd_create_full)(p->cons_on ? FN(problem_constraint_cb) : NULL, FN(problem_obj_cb), FN(problem_grad_cb), (void *)p, x0,
                                    f0, g0_storage, initial_step_length, p->n);
}

void FN(orc_adgd_destroy)(FN(orc_adgd) *o) {
    if (!o) return;
    free(o->dx); free(o->dg); free(o);
}

/* step!(::AdGDOptimizer)  src/DZOptimization.jl:274-312 */
void FN(orc_adgd_step)(FN(orc_adgd) *o) {
    if (o->is_stuck) return;                                 /* :276-278 */
    const T one = (T)1, half = one / (one + one);
    const T inv_sqrt_two = T_SQRT(half);                     /* :283 */
    T previous = o->previous_step_size;                      /* :285 */
    T current = o->current_step_size;                        /* :286 */
    T next = current;                                        /* :287 */
    if (o->iteration_count > 0) {                            /* :288 */
        T theta = current / previous;                        /* :290 */
        next *= T_SQRT(one + theta);                         /* :291 */
        T dgn = FN(orc_norm)(o->dg, o->n);                   /* :292 */
        if (dgn != (T)0) {                                   /* :293 */
            T inv_L = FN(orc_norm)(o->dx, o->n) / dgn;       /* :294 */
            T cap = inv_sqrt_two * inv_L;
            next = next < cap ? next : cap;                  /* :295 */
        }
    }
    o->previous_step_size = current;                         /* :298 */
    o->current_step_size = next;                             /* :299 */
    FN(backtracking_step)(o->constraint, o->objective, o->ctx, o->n, o->x, o->dx, &o->f, &o->df,
                          &o->is_stuck, -next, o->g, o->max_halvings, &o->last_trials); /* :301 */
    if (o->is_stuck) return;                                 /* :302-304 */
    FN(orc_copy)(o->dg, o->g, o->n);                         /* :306 */
    o->gradient(o->ctx, o->g, o->x, o->n);                   /* :307 */
    FN(orc_axpby)(one, o->g, -one, o->dg, o->n);             /* :308 */
    o->iteration_count += 1;                                 /* :310 */
}

/* ------------------------------------------------------------------------------------------
 * LineSearchEvaluator call  (src/DZOptimization.jl:65-92)
 * Returns f_new and writes improvement_ratio / slope_ratio.
 * ---------------------------------------------------------------------------------------- */
T FN(orc_line_search_eval)(FN(orc_constraint_fn) cf, FN(orc_objective_fn) of,
                           FN(orc_gradient_fn) gf, void *ctx, int64_t n, const T *x, T f_old,
                           const T *dir, T overlap, T step_size, int compute_gradient,
                           T *trial_point, T *trial_gradient, T *improvement_ratio,
                           T *slope_ratio) {
    FN(orc_copy)(trial_point, x, n);                         /* :69 */
    FN(orc_axpy)(step_size, dir, trial_point, n);            /* :70 */
    if (cf && !cf(ctx, trial_point, n)) {                    /* :71-79 */
        *improvement_ratio = -T_MAXVAL;
        *slope_ratio = T_MAXVAL;
        return T_MAXVAL;
    }
    T f_new = of(ctx, trial_point, n);                       /* :80 */
    *improvement_ratio = (f_new - f_old) / (step_size * overlap); /* :84 */
    if (compute_gradient) {                                  /* :85-90 */
        gf(ctx, trial_gradient, trial_point, n);
        *slope_ratio = FN(orc_dot)(trial_gradient, dir, n) / overlap;
    }
    return f_new;
}

T FN(orc_line_search_eval_problem)(const FN(orc_problem) *p, const T *x, T f_old, const T *dir, T overlap,
                                   T step_size, int compute_gradient, T *trial_point, T *trial_gradient,
                                   T *improvement_ratio, T *slope_ratio) {
    return FN(orc_line_search_eval)(p->cons_on ? FN(problem_constraint_cb) : NULL, FN(problem_obj_cb),
                                    FN(problem_grad_cb), (void *)p, p->n, x, f_old, dir, overlap, step_size,
                                    compute_gradient, trial_point, trial_gradient, improvement_ratio, slope_ratio);
}

/* ------------------------------------------------------------------------------------------
 * Dense BFGS -- legacy/DZOptimization.jl:733-994 (specification by reading; the legacy code
 * cannot run even under Julia because LineSearchFunctor / quadratic_line_search / add! /
 * scalar_mul! are undefined).  The search below is the build's definition,
 * derived from find_three_point_bracket (:49-172) + QuadraticLineSearch (:191-216), with the
 * trial point x - t*dir implied by the call sites (:945,:973).  It is written once here and
 * mirrored verbatim by the HIP host logic.
 * ---------------------------------------------------------------------------------------- */

typedef struct {
    FN(orc_objective_fn) objective;     /* :734 */
    FN(orc_gradient_fn) gradient;       /* :735 */
    FN(orc_constraint_fn) constraint;   /* :736 (NULL = NULL_CONSTRAINT) */
    void *ctx;
    int64_t iteration_count;            /* :737 */
    int32_t has_terminated;             /* :738 */
    int64_t n;
    T *x;                               /* :739 (copied from the caller, :769) */
    T f;                                /* :740 */
    T *g;                               /* :741 */
    T *dx;                              /* :742 */
    T *dg;                              /* :743 */
    T last_step_length;                 /* :744 */
    int32_t last_step_type;             /* :745  0 Null, 1 GradientDescent, 2 BFGS (:727-731) */
    T *H;                               /* :746 n x n column-major */
    T *d;                               /* :747 next_step_direction = H*g (NOT negated) */
    T *scratch;                         /* :748 */
    T *ref_point;                       /* LineSearchEvaluator.reference_point (:17) */
    int32_t max_increases;              /* QuadraticLineSearch.max_increases (:181-188), 0 = off */
    int64_t evals;                      /* objective evaluations so far */
    T sign;                             /* trial point x + sign*t*dir: -1 for BFGS (:945), +1 for the
                                           legacy evaluator (:33) used by GradientDescentOptimizer */
} FN(orc_bfgs);

/* phi(t) = f(P(x - t*dir)), evaluated out of place into `scratch`
 * (legacy/DZOptimization.jl:25-46 with legacy/Kernels.jl:127-135; sign from :945). */
static T FN(bfgs_phi)(FN(orc_bfgs) *o, const T *dir, T t) {
    FN(orc_axpy_oop)(o->scratch, o->sign * t, dir, o->x, o->n);
    if (o->constraint && !o->constraint(o->ctx, o->scratch, o->n)) return T_MAXVAL; /* :40-42 */
    o->evals += 1;
    return o->objective(o->ctx, o->scratch, o->n);
}

/* find_three_point_bracket  legacy/DZOptimization.jl:49-172, started at step size t0. */
static void FN(bfgs_bracket)(FN(orc_bfgs) *o, const T *dir, T f0, T t0, T *x1, T *f1, T *x2,
                             T *f2) {
    const int64_t n = o->n;
    *x1 = 0; *f1 = f0; *x2 = 0; *f2 = f0;
    if (!T_ISFINITE(f0)) return;                             /* :64-66 */
    if (!(t0 > (T)0) || !T_ISFINITE(t0)) return;             /* zero/NaN direction norm */
    int step_is_zero = 1, point_changed = 0;                 /* :71-80 */
    for (int64_t i = 0; i < n; ++i) {
        step_is_zero &= (dir[i] == (T)0);
        T nw = T_FMA(o->sign * t0, dir[i], o->x[i]);
        point_changed |= (o->x[i] != nw);
    }
    if (step_is_zero) return;                                /* :83-85 */
    T step = t0;
    int step_is_small = 0;
    while (!point_changed) {                                 /* :91-101 */
        step += step;
        step_is_small = 1;
        if (!T_ISFINITE(step)) return;
        for (int64_t i = 0; i < n; ++i) {
            T nw = T_FMA(o->sign * step, dir[i], o->x[i]);
            point_changed |= (o->x[i] != nw);
        }
    }
    T fa = FN(bfgs_phi)(o, dir, step);                       /* :104,:126 */
    if (step_is_small) {                                     /* :107-123 */
        if (fa == T_MAXVAL && o->constraint) return;
        if (FN(orc_isequal)(o->x, o->scratch, n)) return;
    }
    if (fa <= f0) {                                          /* :130 grow */
        int32_t increases = 0;
        FN(orc_copy)(o->ref_point, o->scratch, n);           /* :136 */
        for (;;) {                                           /* :143-156 */
            T dbl = step + step;
            increases += 1;
            T fb = FN(bfgs_phi)(o, dir, dbl);
            if ((o->max_increases > 0 && increases >= o->max_increases) || !T_ISFINITE(fb) ||
                fb > fa || FN(orc_isequal)(o->scratch, o->ref_point, n)) { /* :147-150 */
                *x1 = step; *f1 = fa; *x2 = dbl; *f2 = fb;   /* :151 */
                return;
            }
            step = dbl;
            fa = fb;
            FN(orc_copy)(o->ref_point, o->scratch, n);       /* :155 */
        }
    } else {                                                 /* :157-171 shrink */
        const T half = (T)1 / ((T)1 + (T)1);
        for (;;) {
            T hs = half * step;
            T fb = FN(bfgs_phi)(o, dir, hs);
            if (fb <= f0) {
                *x1 = hs; *f1 = fb; *x2 = step; *f2 = fa;    /* :166 */
                return;
            }
            if (hs == (T)0) return;                          /* underflow guard (build-added) */
            step = hs;
            fa = fb;
        }
    }
}

/* QuadraticLineSearch  legacy/DZOptimization.jl:191-216 */
static void FN(bfgs_quadratic_search)(FN(orc_bfgs) *o, const T *dir, T f0, T t0, T *t_best,
                                      T *f_best) {
    T x1, f1, x2, f2;
    FN(bfgs_bracket)(o, dir, f0, t0, &x1, &f1, &x2, &f2);    /* :195 */
    T xb = 0, fb = f0;                                       /* :196 */
    if (f1 < fb) { xb = x1; fb = f1; }                       /* :197-199 */
    if (f2 < fb) { xb = x2; fb = f2; }                       /* :200-202 */
    T d1 = f0 - f1, d2 = f2 - f1, sum = d1 + d2;             /* :203-205 */
    if (d1 >= (T)0 && d2 >= (T)0 && sum > (T)0) {            /* :206 */
        T ratio = ((d1 + d1) + sum) / (sum + sum);           /* :207-208 */
        T xq = ratio * x1;                                   /* :209 */
        T fq = FN(bfgs_phi)(o, dir, xq);                     /* :210 */
        if (fq < fb) { xb = xq; fb = fq; }                   /* :211-213 */
    }
    *t_best = xb; *f_best = fb;
}

static void FN(identity)(T *H, int64_t n) {                  /* :712-720 */
    for (int64_t j = 0; j < n; ++j)
        for (int64_t i = 0; i < n; ++i) H[i + j * n] = (i == j) ? (T)1 : (T)0;
}

/* t = H*v, column-major; uses the exact symmetry of H (t_j = H[:,j] . v), which the update
 * below preserves bit-for-bit from H0 = I.  Stand-in for mul! at :875,:958-960. */
void FN(orc_symv)(T *t, const T *H, const T *v, int64_t n) {
#pragma omp parallel for num_threads(orc_threads) schedule(static) if (orc_threads > 1)
    for (int64_t j = 0; j < n; ++j) t[j] = FN(orc_dot)(H + j * n, v, n);
}

/* update_inverse_hessian!  legacy/DZOptimization.jl:864-889.  Rescales `dir` in place (:874). */
void FN(orc_bfgs_update)(T *H, T step_length, T *dir, const T *dg, T *scratch, int64_t n) {
    T overlap = FN(orc_dot)(dir, dg, n);                     /* :873 */
    FN(orc_scal)(dir, (T)1 / overlap, n);                    /* :874 scalar_mul!(d, inv(overlap)) */
    FN(orc_symv)(scratch, H, dg, n);                         /* :875 */
    T delta_norm = step_length * overlap + FN(orc_dot)(dg, scratch, n); /* :876 */
#pragma omp parallel for num_threads(orc_threads) schedule(static) if (orc_threads > 1)
    for (int64_t j = 0; j < n; ++j) {                        /* :878 */
        T sj = dir[j], tj = scratch[j];                      /* :879-880 */
        for (int64_t i = 0; i < n; ++i) {                    /* :881 */
            H[i + j * n] += (delta_norm * (dir[i] * sj) - (scratch[i] * sj + dir[i] * tj)); /* :882-884 */
        }
    }
}

/* Constructor  legacy/DZOptimization.jl:762-810 */
FN(orc_bfgs) *FN(orc_bfgs_create)(FN(orc_objective_fn) of, FN(orc_gradient_fn) gf,
                                  FN(orc_constraint_fn) cf, void *ctx, const T *x0,
                                  T initial_step_length, int64_t n) {
    FN(orc_bfgs) *o = (FN(orc_bfgs) *)calloc(1, sizeof(*o));
    o->objective = of; o->gradient = gf; o->constraint = cf; o->ctx = ctx; o->n = n;
    o->x = (T *)malloc((size_t)n * sizeof(T));
    FN(orc_copy)(o->x, x0, n);                               /* :769 */
    if (cf && !cf(ctx, o->x, n)) { free(o->x); free(o); return NULL; } /* :770-771 */
    o->f = of(ctx, o->x, n);                                 /* :772 */
    o->g = (T *)malloc((size_t)n * sizeof(T));
    gf(ctx, o->g, o->x, n);                                  /* :775-776 */
    o->dx = (T *)calloc((size_t)n, sizeof(T));               /* :777 */
    o->dg = (T *)calloc((size_t)n, sizeof(T));               /* :778 */
    o->last_step_length = initial_step_length;               /* :779 */
    o->last_step_type = 0;                                   /* :780 */
    o->H = (T *)malloc((size_t)n * (size_t)n * sizeof(T));
    FN(identity)(o->H, n);                                   /* :781-783 */
    o->d = (T *)malloc((size_t)n * sizeof(T));
    FN(orc_copy)(o->d, o->g, n);                             /* :784 */
    o->scratch = (T *)malloc((size_t)n * sizeof(T));         /* :785 */
    o->ref_point = (T *)malloc((size_t)n * sizeof(T));
    o->has_terminated = isnan(o->f) ? 1 : 0;                 /* :773 @assert -> terminated */
    o->sign = (T)-1;
    return o;
}

FN(orc_bfgs) *FN(orc_bfgs_create_problem)(const FN(orc_problem) *p, const T *x0,
                                          T initial_step_length) {
    return FN(orc_bfgs_create)(FN(problem_obj_cb), FN(problem_grad_cb),
                               p->cons_on ? FN(problem_constraint_cb) : NULL, (void *)p, x0,
                               initial_step_length, p->n);
}

void FN(orc_bfgs_destroy)(FN(orc_bfgs) *o) {
    if (!o) return;
    free(o->x); free(o->g); free(o->dx); free(o->dg); free(o->H); free(o->d); free(o->scratch);
    free(o->ref_point); free(o);
}

/* Move along -t*dir and refresh g, dx, dg  (:943-950 / :971-978). */
static void FN(bfgs_move)(FN(orc_bfgs) *o, T t, const T *dir) {
    const int64_t n = o->n;
    for (int64_t i = 0; i < n; ++i) { o->dx[i] = -o->x[i]; o->dg[i] = -o->g[i]; } /* :943-944 */
    FN(orc_axpy)(-t, dir, o->x, n);                          /* :945 add!(point, -t, dir) */
    if (o->constraint) o->constraint(o->ctx, o->x, n);       /* :946-947 */
    o->gradient(o->ctx, o->g, o->x, n);                      /* :948 */
    for (int64_t i = 0; i < n; ++i) { o->dx[i] += o->x[i]; o->dg[i] += o->g[i]; } /* :949-950 */
}

/* step!(::BFGSOptimizer)  legacy/DZOptimization.jl:891-994 */
void FN(orc_bfgs_step)(FN(orc_bfgs) *o) {
    if (o->has_terminated) return;                           /* :893 */
    const int64_t n = o->n;
    T step_length = o->last_step_length;                     /* :918 */
    T grad_norm = FN(orc_norm)(o->g, n);                     /* :921 */
    T t_g, f_g;
    FN(bfgs_quadratic_search)(o, o->g, o->f, step_length / grad_norm, &t_g, &f_g); /* :922-925 */
    T bfgs_norm = FN(orc_norm)(o->d, n);                     /* :928 */
    T t_b, f_b;
    FN(bfgs_quadratic_search)(o, o->d, o->f, step_length / bfgs_norm, &t_b, &f_b); /* :929-932 */

    if (f_b < o->f && !(f_b > f_g)) {                        /* :934 */
        o->f = f_b;                                          /* :937 */
        o->last_step_length = t_b * bfgs_norm;               /* :938 */
        o->last_step_type = 2;                               /* :939 */
        o->iteration_count += 1;                             /* :940 */
        FN(bfgs_move)(o, t_b, o->d);                         /* :943-950 */
        FN(orc_bfgs_update)(o->H, -t_b, o->d, o->dg, o->scratch, n); /* :953-955 */
        FN(orc_symv)(o->d, o->H, o->g, n);                   /* :958-960 */
    } else if (f_g < o->f) {                                 /* :962 */
        o->f = f_g;                                          /* :965 */
        o->last_step_length = t_g * grad_norm;               /* :966 */
        o->last_step_type = 1;                               /* :967 */
        o->iteration_count += 1;                             /* :968 */
        FN(bfgs_move)(o, t_g, o->g);                         /* :971-978 (x moves along the old
                                                                g before g is refreshed) */
        FN(identity)(o->H, n);                               /* :981 */
        FN(orc_copy)(o->d, o->g, n);                         /* :984-986 */
    } else {
        o->has_terminated = 1;                               /* :989 */
    }
}

/* ------------------------------------------------------------------------------------------
 * Legacy GradientDescentOptimizer -- legacy/DZOptimization.jl:305-449,
 * with line_search_function! = QuadraticLineSearch() (:181-216).  It reuses the search state
 * of the BFGS restatement with sign = +1 (trial point x + t*d, :33) and d = next_step_direction.
 * Fields: f / last_step_length / iteration_count / has_terminated / x / g / dx / dg / d as above;
 * `df` is delta_objective_value (:313).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    FN(orc_bfgs) b;     /* H stays NULL */
    T df;
} FN(orc_gd);

static T FN(inv_norm)(const T *x, int64_t n) { return (T)1 / T_SQRT(FN(orc_norm2)(x, n)); }   /* Kernels.jl:141 rsqrt(norm2) */

/* constructor :330-374 */
FN(orc_gd) *FN(orc_gd_create_problem)(const FN(orc_problem) *p, const T *x0, T initial_step_length) {
    FN(orc_gd) *o = (FN(orc_gd) *)calloc(1, sizeof(*o));
    FN(orc_bfgs) *b = &o->b;
    const int64_t n = p->n;
    b->objective = FN(problem_obj_cb); b->gradient = FN(problem_grad_cb);
    b->constraint = p->cons_on ? FN(problem_constraint_cb) : NULL;
    b->ctx = (void *)p; b->n = n; b->sign = (T)1;
    b->x = (T *)malloc((size_t)n * sizeof(T));
    FN(orc_copy)(b->x, x0, n);                                    /* :339 collect */
    if (b->constraint) b->constraint(b->ctx, b->x, n);            /* :340 @assert */
    b->dx = (T *)calloc((size_t)n, sizeof(T));                    /* :341 */
    b->f = b->objective(b->ctx, b->x, n);                         /* :343 */
    b->g = (T *)malloc((size_t)n * sizeof(T));
    b->gradient(b->ctx, b->g, b->x, n);                           /* :347-348 */
    b->dg = (T *)calloc((size_t)n, sizeof(T));                    /* :349 */
    b->last_step_length = 0;                                      /* :351 */
    T ign = FN(inv_norm)(b->g, n);                                /* :352 */
    b->d = (T *)calloc((size_t)n, sizeof(T));                     /* :353 */
    if (T_ISFINITE(ign)) {                                        /* :354-357 */
        FN(orc_copy)(b->d, b->g, n);
        FN(orc_scal)(b->d, -initial_step_length * ign, n);
    }
    b->scratch = (T *)malloc((size_t)n * sizeof(T));
    b->ref_point = (T *)malloc((size_t)n * sizeof(T));
    b->has_terminated = (!T_ISFINITE(b->f)) || (!T_ISFINITE(ign));   /* :364-366 */
    return o;
}

void FN(orc_gd_destroy)(FN(orc_gd) *o) {
    if (!o) return;
    free(o->b.x); free(o->b.g); free(o->b.dx); free(o->b.dg); free(o->b.d); free(o->b.scratch); free(o->b.ref_point);
    free(o);
}

/* step! :393-449 */
void FN(orc_gd_step)(FN(orc_gd) *o) {
    FN(orc_bfgs) *b = &o->b;
    const int64_t n = b->n;
    if (b->has_terminated) return;                                /* :402 */
    T t, fv;
    FN(bfgs_quadratic_search)(b, b->d, b->f, (T)1, &t, &fv);      /* :405-407 bracket starts at step size 1 (:89) */
    if (t == (T)0 || !(fv < b->f)) { b->has_terminated = 1; return; }   /* :410-414 */
    b->iteration_count += 1;                                      /* :415 */
    FN(orc_copy)(b->dx, b->x, n);                                 /* :418 */
    FN(orc_axpy)(t, b->d, b->x, n);                               /* :419 */
    if (b->constraint) b->constraint(b->ctx, b->x, n);            /* :420 */
    for (int64_t i = 0; i < n; ++i) b->dx[i] = b->x[i] - b->dx[i];   /* :423 delta! */
    T step_length = T_SQRT(FN(orc_norm2)(b->dx, n));              /* :424 */
    b->last_step_length = step_length;                            /* :425 */
    o->df = fv - b->f;                                            /* :428-429 */
    b->f = fv;                                                    /* :430 */
    FN(orc_copy)(b->dg, b->g, n);                                 /* :433 */
    b->gradient(b->ctx, b->g, b->x, n);                           /* :434 */
    for (int64_t i = 0; i < n; ++i) b->dg[i] = b->g[i] - b->dg[i];   /* :435 */
    T ign = FN(inv_norm)(b->g, n);                                /* :438 */
    if (!T_ISFINITE(ign)) { b->has_terminated = 1; return; }      /* :439-442 */
    const T sc = -step_length * ign;
    for (int64_t i = 0; i < n; ++i) b->d[i] = sc * b->g[i];       /* :445-446 scale!(dst, alpha, x) */
}

FN(orc_bfgs) *FN(orc_gd_base)(FN(orc_gd) *o) { return &o->b; }
T FN(orc_gd_delta_f)(const FN(orc_gd) *o) { return o->df; }

/* ------------------------------------------------------------------------------------------
 * Field accessors for the ctypes loader (oracle/oracle.py); no reference counterpart -- in
 * the reference every field is a public struct member (README.md:11).
 * ---------------------------------------------------------------------------------------- */
int64_t FN(orc_lbfgs_get_i)(const FN(orc_lbfgs) *o, int what) {
    switch (what) {
    case 0: return o->is_stuck;
    case 1: return o->iteration_count;
    case 2: return o->n;
    case 3: return o->m;
    case 4: return o->k;
    case 5: return o->n_alpha;
    case 6: return o->n_rho;
    case 7: return o->last_trials;
    case 8: return o->history_resets;
    case 9: return o->descent_resets;
    case 10: return o->last_step_kind;
    }
    return -1;
}
T FN(orc_lbfgs_get_s)(const FN(orc_lbfgs) *o, int what) {
    return what == 0 ? o->f : what == 1 ? o->df : o->last_step_length;
}
/* test infrastructure: install f (and clear the stuck flag) when a test makes the oracle follow another optimizer's state */
void FN(orc_lbfgs_set_f)(FN(orc_lbfgs) *o, T f, int32_t is_stuck) { o->f = f; o->is_stuck = is_stuck; }
void FN(orc_lbfgs_set_safeguards)(FN(orc_lbfgs) *o, int32_t descent_check, int32_t sd_fallback) {
    o->descent_check = descent_check; o->sd_fallback = sd_fallback;
}
void FN(orc_lbfgs_set_line_search)(FN(orc_lbfgs) *o, int32_t kind, T c1, T c2, int32_t max_evals) {
    o->line_search = kind;
    if (c1 > (T)0) o->wolfe_c1 = c1;
    if (c2 > (T)0) o->wolfe_c2 = c2;
    if (max_evals > 0) o->wolfe_max_evals = max_evals;
}
T *FN(orc_lbfgs_get_v)(const FN(orc_lbfgs) *o, int what, int idx) {
    switch (what) {
    case 0: return o->x;
    case 1: return o->dx;
    case 2: return o->g;
    case 3: return o->dg;
    case 4: return o->d;
    case 5: return idx < o->k ? o->S[idx] : NULL;
    case 6: return idx < o->k ? o->Y[idx] : NULL;
    case 7: return o->alpha;
    case 8: return o->rho;
    }
    return NULL;
}
void FN(orc_lbfgs_set_max_halvings)(FN(orc_lbfgs) *o, int64_t v) { o->max_halvings = v; }
/* Install a history directly (frozen-state parity tests): S/Y are k x n row-major, newest
 * first; rho[i] = s_i . y_i is recomputed with the current dot mode unless given. */
void FN(orc_lbfgs_set_history)(FN(orc_lbfgs) *o, int32_t k, const T *S, const T *Y,
                               const T *rho_or_null, int64_t iteration_count) {
    for (int32_t i = 0; i < o->k; ++i) { free(o->S[i]); free(o->Y[i]); }
    o->k = k; o->n_alpha = k; o->n_rho = k;
    for (int32_t i = 0; i < k; ++i) {
        o->S[i] = (T *)malloc((size_t)o->n * sizeof(T));
        o->Y[i] = (T *)malloc((size_t)o->n * sizeof(T));
        FN(orc_copy)(o->S[i], S + (int64_t)i * o->n, o->n);
        FN(orc_copy)(o->Y[i], Y + (int64_t)i * o->n, o->n);
        o->rho[i] = rho_or_null ? rho_or_null[i] : FN(orc_dot)(o->S[i], o->Y[i], o->n);
    }
    o->iteration_count = iteration_count;
}

int64_t FN(orc_adgd_get_i)(const FN(orc_adgd) *o, int what) {
    switch (what) {
    case 0: return o->is_stuck;
    case 1: return o->iteration_count;
    case 2: return o->n;
    case 7: return o->last_trials;
    }
    return -1;
}
T FN(orc_adgd_get_s)(const FN(orc_adgd) *o, int what) {
    switch (what) {
    case 0: return o->f;
    case 1: return o->df;
    case 2: return o->current_step_size;
    case 3: return o->previous_step_size;
    }
    return (T)NAN;
}
T *FN(orc_adgd_get_v)(const FN(orc_adgd) *o, int what) {
    switch (what) {
    case 0: return o->x;
    case 1: return o->dx;
    case 2: return o->g;
    case 3: return o->dg;
    }
    return NULL;
}

int64_t FN(orc_bfgs_get_i)(const FN(orc_bfgs) *o, int what) {
    switch (what) {
    case 0: return o->has_terminated;
    case 1: return o->iteration_count;
    case 2: return o->n;
    case 3: return o->last_step_type;
    case 4: return o->evals;
    }
    return -1;
}
T FN(orc_bfgs_get_s)(const FN(orc_bfgs) *o, int what) {
    return what == 0 ? o->f : o->last_step_length;
}
T *FN(orc_bfgs_get_v)(const FN(orc_bfgs) *o, int what) {
    switch (what) {
    case 0: return o->x;
    case 1: return o->dx;
    case 2: return o->g;
    case 3: return o->dg;
    case 4: return o->d;
    case 5: return o->H;
    case 6: return o->scratch;
    }
    return NULL;
}
void FN(orc_bfgs_set_max_increases)(FN(orc_bfgs) *o, int32_t v) { o->max_increases = v; }
/* k step! calls in one call (the timing baseline runs one optimizer per host thread; a Python-level loop
 * would hand the interpreter lock around between the threads at every step) */
void FN(orc_bfgs_steps)(FN(orc_bfgs) *o, int32_t k) {
    for (int32_t i = 0; i < k; ++i) FN(orc_bfgs_step)(o);
}
/* state installation for the per-step parity tests (every field of the reference's struct is public,
 * legacy/DZOptimization.jl:733-751; the vectors and H are written through orc_bfgs_get_v) */
void FN(orc_bfgs_set_s)(FN(orc_bfgs) *o, int what, T v) {
    if (what == 0) o->f = v; else o->last_step_length = v;
}
void FN(orc_bfgs_set_i)(FN(orc_bfgs) *o, int what, int64_t v) {
    switch (what) {
    case 0: o->has_terminated = v != 0; break;
    case 1: o->iteration_count = v; break;
    case 3: o->last_step_type = (int32_t)v; break;
    case 4: o->evals = v; break;
    }
}
/* exposed for unit tests of the search itself */
void FN(orc_bfgs_line_search)(FN(orc_bfgs) *o, int use_gradient_dir, T t0, T *t_best, T *f_best) {
    FN(bfgs_quadratic_search)(o, use_gradient_dir ? o->g : o->d, o->f, t0, t_best, f_best);
}

#undef T_FMA
#undef T_SQRT
#undef T_MAXVAL
#undef T_ISFINITE
#undef FN
#undef CAT
#undef CAT_
