// dzo_adgd.hip -- AdGDOptimizer + step! (src/DZOptimization.jl:179-312), SURVEY.md 8(f).1.
// Reuses the backtracking kernels of dzo_optcore.hip; adds only two nrm2 reductions.
#include "dzo_optcore.h"

struct dzo_adgd_s {
    dzo::OptCore core;
    double current_step_size = 0;    // :195
    double previous_step_size = 0;   // :196
    void *dx_buf = nullptr, *dg_buf = nullptr;
};

namespace dzo {

static int32_t norm_blocking(OptCore &c, const void *v, double *out) {
    double ss = 0;
    DZO_TRY(dot_blocking(c.stream, c.n, c.dtype, v, v, c.partials(), c.host, &ss));
    *out = c.dtype == DZO_F32 ? (double)sqrtf((float)ss) : sqrt(ss);
    return DZO_OK;
}

static int32_t adgd_step(dzo_adgd_s *o) {
    OptCore &c = o->core;
    if (c.is_stuck) return DZO_OK;                                   // :276-278
    DZO_REQUIRE(c.has_objective() && c.has_gradient(), DZO_ERR_STATE,
                "step! needs objective and gradient (callbacks or a built-in problem)");
    const int32_t dt = c.dtype;
    const double half = 0.5;
    const double inv_sqrt_two = dt == DZO_F32 ? (double)sqrtf(0.5f) : sqrt(0.5);   // :283
    const double previous = o->previous_step_size;                   // :285
    const double current = o->current_step_size;                     // :286
    double next = current;                                           // :287
    (void)half;
    if (c.iteration_count > 0) {                                     // :288
        DZO_REQUIRE(previous != 0.0, DZO_ERR_ASSERT, "@assert !iszero(previous_step_size) (src/DZOptimization.jl:289)");
        const double theta = round_to_dtype(dt, current / previous); // :290
        const double root = dt == DZO_F32 ? (double)sqrtf((float)(1.0 + theta)) : sqrt(1.0 + theta);
        next = round_to_dtype(dt, next * root);                      // :291
        double dgn = 0;
        DZO_TRY(norm_blocking(c, c.dg, &dgn));                       // :292
        if (dgn != 0.0) {                                            // :293
            double dxn = 0;
            DZO_TRY(norm_blocking(c, c.dx, &dxn));
            const double inv_L = round_to_dtype(dt, dxn / dgn);      // :294
            const double cap = round_to_dtype(dt, inv_sqrt_two * inv_L);
            next = next < cap ? next : cap;                          // :295
        }
    }
    o->previous_step_size = current;                                 // :298
    o->current_step_size = next;                                     // :299
    DZO_TRY(core_backtracking_step(c, -next, c.g));                  // :301
    if (c.is_stuck) return DZO_OK;                                   // :302-304
    DZO_HIP(hipMemcpyAsync(c.dg, c.g, (size_t)c.n * dtype_size(dt), hipMemcpyDeviceToDevice, c.stream));  // :306
    DZO_TRY(core_gradient(c));                                       // :307
    DZO_DISPATCH(dt, launch_axpby<T>(c.stream, c.n, (T)1, (const T *)c.g, (T)-1, (T *)c.dg));          // :308
    DZO_HIP(hipGetLastError());
    c.iteration_count += 1;                                          // :310
    return DZO_OK;
}

}  // namespace dzo

using namespace dzo;

extern "C" {

int32_t dzo_adgd_create(int64_t n, int32_t dtype, void *x_dev, void *g_dev, double initial_objective_value,
                        double initial_step_length, dzo_adgd_t *out) {
    DZO_TRY(require_init());
    DZO_REQUIRE(out && x_dev && g_dev && n >= 1, DZO_ERR_INVALID, "bad argument");
    DZO_REQUIRE(dtype == DZO_F32 || dtype == DZO_F64, DZO_ERR_INVALID, "bad dtype %d", dtype);
    DZO_REQUIRE(initial_step_length > 0, DZO_ERR_ASSERT, "@assert initial_step_length > 0 (src/DZOptimization.jl:229)");
    dzo_adgd_s *o = new dzo_adgd_s();
    OptCore &c = o->core;
    c.n = n; c.dtype = dtype; c.x = x_dev; c.g = g_dev;
    c.f = round_to_dtype(dtype, initial_objective_value);
    int32_t rc = core_alloc(c);
    if (rc != DZO_OK) { delete o; return rc; }
    const size_t bytes = (size_t)((n + 63) / 64 * 64) * dtype_size(dtype);
    if (hipMalloc(&o->dx_buf, bytes) != hipSuccess || hipMalloc(&o->dg_buf, bytes) != hipSuccess) {
        dzo_adgd_destroy(o);
        set_error("out of device memory allocating AdGD state");
        return DZO_ERR_NOMEM;
    }
    DZO_HIP(hipMemsetAsync(o->dx_buf, 0, bytes, c.stream));        // :219-222
    DZO_HIP(hipMemsetAsync(o->dg_buf, 0, bytes, c.stream));        // :224-227
    c.dx = o->dx_buf; c.dg = o->dg_buf;
    double gnorm = 0;
    rc = norm_blocking(c, g_dev, &gnorm);                          // :230
    if (rc != DZO_OK) { dzo_adgd_destroy(o); return rc; }
    c.is_stuck = (gnorm == 0.0);                                   // :231
    const double s0 = c.is_stuck ? 0.0 : round_to_dtype(dtype, initial_step_length / gnorm);  // :232-233
    o->current_step_size = s0; o->previous_step_size = s0;         // :241
    *out = o;
    return DZO_OK;
}

int32_t dzo_adgd_create_problem(dzo_problem_t problem, void *x_dev, double initial_step_length, dzo_adgd_t *out) {
    DZO_TRY(require_init());
    DZO_REQUIRE(problem && x_dev && out, DZO_ERR_INVALID, "null argument");
    if (problem->cons_on)                                          // :256-258
        DZO_TRY(dzo_box_clamp(problem->n, problem->dtype, x_dev, problem->cons_lo, problem->cons_hi));
    double f0 = 0;
    DZO_TRY(dzo_problem_eval(problem, x_dev, &f0));                // :260
    void *g = nullptr;
    DZO_HIP(hipMalloc(&g, (size_t)((problem->n + 63) / 64 * 64) * dtype_size(problem->dtype)));   // :262
    int32_t rc = dzo_problem_grad(problem, g, x_dev);              // :265
    if (rc == DZO_OK) rc = dzo_adgd_create(problem->n, problem->dtype, x_dev, g, f0, initial_step_length, out);
    if (rc != DZO_OK) { (void)hipFree(g); return rc; }
    (*out)->core.owns_g = true;
    (*out)->core.problem = problem;
    (*out)->core.box_on = problem->cons_on; (*out)->core.box_lo = problem->cons_lo; (*out)->core.box_hi = problem->cons_hi;
    return DZO_OK;
}

int32_t dzo_adgd_destroy(dzo_adgd_t o) {
    if (!o) return DZO_OK;
    if (o->core.stream) (void)hipStreamSynchronize(o->core.stream);
    if (o->dx_buf) (void)hipFree(o->dx_buf);
    if (o->dg_buf) (void)hipFree(o->dg_buf);
    core_free(o->core);
    delete o;
    return DZO_OK;
}

int32_t dzo_adgd_set_callbacks(dzo_adgd_t o, dzo_constraint_fn constraint, dzo_objective_fn objective,
                               dzo_gradient_fn gradient, void *cb_ctx) {
    DZO_REQUIRE(o, DZO_ERR_INVALID, "null optimizer");
    o->core.constraint = constraint; o->core.objective = objective; o->core.gradient = gradient;
    o->core.cb_ctx = cb_ctx;
    return DZO_OK;
}

int32_t dzo_adgd_step(dzo_adgd_t o) {
    DZO_REQUIRE(o, DZO_ERR_INVALID, "null optimizer");
    return adgd_step(o);
}

int32_t dzo_adgd_get_i(dzo_adgd_t o, int32_t what, int64_t *value) {
    DZO_REQUIRE(o && value, DZO_ERR_INVALID, "null argument");
    switch (what) {
    case 0: *value = o->core.is_stuck ? 1 : 0; break;
    case 1: *value = o->core.iteration_count; break;
    case 2: *value = o->core.n; break;
    default: set_error("dzo_adgd_get_i: unknown field %d", what); return DZO_ERR_INVALID;
    }
    return DZO_OK;
}

int32_t dzo_adgd_get_s(dzo_adgd_t o, int32_t what, double *value) {
    DZO_REQUIRE(o && value, DZO_ERR_INVALID, "null argument");
    switch (what) {
    case 0: *value = o->core.f; break;
    case 1: *value = o->core.df; break;
    case 2: *value = o->current_step_size; break;
    case 3: *value = o->previous_step_size; break;
    default: set_error("dzo_adgd_get_s: unknown field %d", what); return DZO_ERR_INVALID;
    }
    return DZO_OK;
}

int32_t dzo_adgd_get_ptr(dzo_adgd_t o, int32_t what, void **ptr_dev) {
    DZO_REQUIRE(o && ptr_dev, DZO_ERR_INVALID, "null argument");
    DZO_HIP(hipStreamSynchronize(o->core.stream));
    switch (what) {
    case 0: *ptr_dev = o->core.x; break;
    case 1: *ptr_dev = o->core.dx; break;
    case 2: *ptr_dev = o->core.g; break;
    case 3: *ptr_dev = o->core.dg; break;
    default: set_error("dzo_adgd_get_ptr: unknown field %d", what); return DZO_ERR_INVALID;
    }
    return DZO_OK;
}

}  // extern "C"
