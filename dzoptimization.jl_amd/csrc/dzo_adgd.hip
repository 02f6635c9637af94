// dzo_adgd.hip -- AdGDOptimizer + step! (src/DZOptimization.jl:179-312), SURVEY.md 8(f).1.
// Reuses the backtracking kernels of dzo_optcore.hip; adds only two nrm2 reductions.
#include "dzo_optcore.h"

#include "dzo_problems.h"
#include "dzo_rosen.h"

struct dzo_adgd_s {
    dzo::OptCore core;
    double current_step_size = 0;    // :195
    double previous_step_size = 0;   // :196
    void *dx_buf = nullptr, *dg_buf = nullptr, *dg_alt = nullptr;
    // fused step (built-in chained Rosenbrock): one pass does :301 (first trial), :306-308 and the
    // two sums of squares that :292 / :294 of the NEXT step need.  The pass never writes x or g: the
    // trial point and its gradient go to TWIN buffers, which swap roles with x / g when the trial is
    // accepted (no backups, nothing to restore after a rejected trial, no boundary snapshots).  x_user /
    // g_user are the arrays the optimizer aliases (:253-254 of the reference constructor); they are
    // settled whenever the host looks (get_ptr, dzo_synchronize, dzo_memcpy_*, destroy).
    bool fused = true;               // DZO_TUNE_ADGD_FUSED=0 forces the generic kernel sequence
    void *twin = nullptr, *x_twin = nullptr, *g_twin = nullptr;
    void *x_user = nullptr, *g_user = nullptr;
    bool unsettled = false;
    int device = 0;
    std::recursive_mutex mu;
    bool norms_ready = false;        // |delta_point|^2, |delta_gradient|^2 of the last step are in norm2[]
    double norm2[2] = {0, 0};
    int64_t fused_steps = 0, fused_rejections = 0;
};

namespace dzo {

// ---------------------------------------------------------------------------------------------
// Fused AdGD step for the built-in chained Rosenbrock objective.  Wave-rows of 62 owned 16-B vectors
// plus one halo vector on each side (the 3-point stencil of the gradient needs x_new of both
// neighbours; the halo lanes recompute it from x_old / g_old, which this pass only reads).  Per element:
//   x_new = fma(-step, g_old, x_old)                 take_backtracking_step! :124 (first trial)
//   objective terms of x_new, changed flag            :128, :138
//   g_new = grad f(x_new)                             :307
//   delta_point = x_new - x_old                       :145
//   delta_gradient = g_new - g_old                    :306, :308
//   partial sums of |delta_point|^2, |delta_gradient|^2   (:292, :294 of the next step)
// 2 reads + 4 writes per element instead of the 16 element passes of the separate kernels.
constexpr int kAdgdOwn = 62;

template <typename T> struct AdgdFusedParams {
    int64_t n;
    T t;                                       // -step size
    const T *x, *g;                            // read only
    T *x_out, *g_out, *dx, *dg;
    double *partials;                          // [3][gridDim.x]: objective, |dx|^2, |dg|^2
    int32_t *changed;
};

// One kernel for the first trial and for every later trial of the same step (:151-152): x and g still hold
// x_old / g_old, whatever happened before.
template <typename T>
__global__ __launch_bounds__(kBlock) void adgd_fused_rosen_kernel(AdgdFusedParams<T> p) {
    constexpr int N = Vec16<T>::N;
    __shared__ double lds[kWaves];
    __shared__ int lds_flag;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t nvec = p.n / N;
    const int64_t rows = (nvec + kAdgdOwn - 1) / kAdgdOwn;
    const bool halo_lane = lane == 0 || lane == 63;
    double fobj = 0, sdx = 0, sdg = 0;
    bool diff = false;
    for (int64_t row = (int64_t)blockIdx.x * kWaves + wave; row < rows; row += (int64_t)gridDim.x * kWaves) {
        const int64_t v = row * kAdgdOwn - 1 + lane;
        const bool valid = v >= 0 && v < nvec;
        const bool owner = valid && !halo_lane;
        const int64_t vc = v < 0 ? 0 : (v >= nvec ? nvec - 1 : v);          // clamped: the value is never used
        const int64_t e0 = v * N;
        T xo[N], go[N];
        load16(p.x + vc * N, xo);                                            // halo lanes read their neighbours directly
        load16(p.g + vc * N, go);
        T xn[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            xn[j] = dfma(p.t, go[j], xo[j]);                               // :124
            diff |= owner && !is_equal(xn[j], xo[j]);                      // :128
        }
        const T xprev = __shfl_up(xn[N - 1], 1, 64);
        const T xnext = __shfl_down(xn[0], 1, 64);
        T gn[N], sn[N], yn[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const T xp = j > 0 ? xn[(j + N - 1) % N] : xprev;
            const T xq = j + 1 < N ? xn[(j + 1) % N] : xnext;
            gn[j] = rosen_grad_elem<T>(e0 + j, p.n, xp, xn[j], xq);        // :307
            sn[j] = xn[j] - xo[j];                                         // :145
            yn[j] = gn[j] - go[j];                                         // :308
            if (owner) {
                if (e0 + j + 1 < p.n) fobj += rosen_term<T>(xn[j], xq);    // :138
                sdx = __builtin_fma((double)sn[j], (double)sn[j], sdx);
                sdg = __builtin_fma((double)yn[j], (double)yn[j], sdg);
            }
        }
        if (owner) {
            store16_nt(p.x_out + v * N, xn);
            store16_nt(p.g_out + v * N, gn);
            store16_nt(p.dx + v * N, sn);
            store16_nt(p.dg + v * N, yn);
        }
    }
    block_raise_flag(diff, p.changed, &lds_flag);
    const double f = block_sum(fobj, lds);
    const double a = block_sum(sdx, lds);
    const double b = block_sum(sdg, lds);
    if (threadIdx.x == 0) {
        p.partials[blockIdx.x] = f;
        p.partials[(int64_t)gridDim.x + blockIdx.x] = a;
        p.partials[2 * (int64_t)gridDim.x + blockIdx.x] = b;
    }
}

// fixed-order sums of the three partial arrays, the :128 / :139 decision, everything the host needs
// straight into its pinned mirror: {f_new, -, -, status, changed, |dx|^2, |dg|^2}
__global__ __launch_bounds__(kBlock) void adgd_decide_kernel(const double *__restrict__ partials, int grid,
                                                             int32_t *__restrict__ changed, double f_cur, int to_f32,
                                                             double *__restrict__ host_out) {
    __shared__ double lds[kWaves];
    double v[3];
    for (int c = 0; c < 3; ++c) {
        double a = 0;
        for (int i = threadIdx.x; i < grid; i += kBlock) a += partials[(int64_t)c * grid + i];
        v[c] = block_sum(a, lds);
    }
    if (threadIdx.x == 0) {
        double f_new = v[0];
        if (to_f32) f_new = (double)(float)f_new;
        const int32_t ch = *changed;
        int32_t st = 0;
        if (ch == 0) st = 2;
        else if (f_new < f_cur) st = 1;
        *changed = 0;
        host_out[0] = v[0];
        host_out[5] = v[1];
        host_out[6] = v[2];
        reinterpret_cast<int32_t *>(host_out + 3)[0] = st;
        reinterpret_cast<int32_t *>(host_out + 4)[0] = ch;
        __threadfence_system();
    }
}

static int32_t adgd_settle_entry(void *h);

static void adgd_mark_unsettled(dzo_adgd_s *o) {
    const bool dirty = o->core.x != o->x_user || o->core.g != o->g_user;
    if (dirty && !o->unsettled) { unsettled_add(o, adgd_settle_entry); o->unsettled = true; }
    if (!dirty && o->unsettled) { unsettled_remove(o); o->unsettled = false; }
}

// current_point / current_gradient back into the arrays the optimizer aliases
static int32_t adgd_settle(dzo_adgd_s *o) {
    std::lock_guard<std::recursive_mutex> lk(o->mu);
    OptCore &c = o->core;
    const size_t bytes = (size_t)c.n * dtype_size(c.dtype);
    if (c.x != o->x_user) {
        DZO_HIP(hipMemcpyAsync(o->x_user, c.x, bytes, hipMemcpyDeviceToDevice, c.stream));
        o->x_twin = c.x; c.x = o->x_user;
    }
    if (c.g != o->g_user) {
        DZO_HIP(hipMemcpyAsync(o->g_user, c.g, bytes, hipMemcpyDeviceToDevice, c.stream));
        o->g_twin = c.g; c.g = o->g_user;
    }
    adgd_mark_unsettled(o);
    return DZO_OK;
}

static int32_t adgd_settle_entry(void *h) {
    dzo_adgd_s *o = static_cast<dzo_adgd_s *>(h);
    DeviceScope scope(o->device);
    DZO_TRY(adgd_settle(o));
    DZO_HIP(hipStreamSynchronize(o->core.stream));
    return DZO_OK;
}

// Eligibility of the one-pass step, INCLUDING its buffers (the twins of x and g, the second
// delta_gradient buffer): they are allocated here, before step! changes anything, and a failed
// allocation only switches the optimizer to the generic kernel sequence (which needs no extra memory)
// instead of failing the step.
static bool adgd_fused_ok(dzo_adgd_s *o) {
    OptCore &c = o->core;
    if (!o->fused || c.objective || c.gradient || c.constraint || c.box_on || !c.problem) return false;
    const int vecn = 16 / (int)dtype_size(c.dtype);
    if (c.n % vecn != 0 || c.n < 4 * vecn) return false;
    if (!problem_has_fused_post(c.problem, c.x, c.dx, c.g, c.dg)) return false;
    const size_t es = dtype_size(c.dtype);
    if (!o->twin) {
        const size_t padded = (size_t)((c.n + 63) / 64 * 64) * es;
        const size_t slot = ((padded + 1023) / 1024 | 1) * 1024;             // an odd number of KiB apart
        if (hipMalloc(&o->twin, 2 * slot + 16 * 1024) != hipSuccess) { (void)hipGetLastError(); o->twin = nullptr; o->fused = false; return false; }
        o->x_twin = (char *)o->twin + 5 * 1024;
        o->g_twin = (char *)o->x_twin + slot;
    }
    if (!o->dg_alt) {
        const size_t bytes = (size_t)((c.n + 63) / 64 * 64) * es;
        if (hipMalloc(&o->dg_alt, bytes) != hipSuccess) { (void)hipGetLastError(); o->dg_alt = nullptr; o->fused = false; return false; }
    }
    return true;
}

// The whole step on fused passes: the first trial and, after a rejection, the halving loop of
// take_backtracking_step! (:121-152) re-run the same pass with half the step (x and g are intact).
// *done = false: not eligible or nothing touched, the caller runs the generic sequence.
template <typename T> static int32_t adgd_fused_step(dzo_adgd_s *o, double step, bool *done) {
    OptCore &c = o->core;
    hipStream_t s = c.stream;
    constexpr int N = Vec16<T>::N;
    *done = false;
    const int64_t nvec = c.n / N;
    const int64_t rows = (nvec + kAdgdOwn - 1) / kAdgdOwn;
    AdgdFusedParams<T> fp;
    fp.n = c.n;
    fp.x = (const T *)c.x; fp.g = (const T *)c.g; fp.x_out = (T *)o->x_twin; fp.g_out = (T *)o->g_twin;
    // delta_gradient is written into the OTHER of two buffers and the pointers swap when the step is
    // accepted: a step that ends stuck leaves the previous step's delta_gradient untouched, as
    // take_backtracking_step! does (:128-130 returns before anything but delta_point was written)
    void *dg_new = (c.dg == o->dg_buf) ? o->dg_alt : o->dg_buf;
    fp.dx = (T *)c.dx; fp.dg = (T *)dg_new;
    fp.partials = c.partials();
    fp.changed = c.flag();
    int64_t blocks = (rows + kWaves - 1) / kWaves;
    if (blocks > 1024) blocks = 1024;                                        // 3 x 1024 partials fit the workspace
    const int grid = (int)(blocks < 1 ? 1 : blocks);
    if (!c.flag_armed) DZO_HIP(hipMemsetAsync(c.flag(), 0, sizeof(int32_t), s));
    c.flag_armed = false;
    auto give_up = [&]() -> int32_t {                                        // :118 delta_point holds x_old when the search gives up
        DZO_HIP(hipMemcpyAsync(c.dx, c.x, (size_t)c.n * sizeof(T), hipMemcpyDeviceToDevice, s));
        return DZO_OK;
    };
    *done = true;
    c.last_trials = 0;
    int64_t halvings = 0;
    double t = step;
    for (bool first = true;; first = false) {                                // :121
        fp.t = (T)(-t);
        {
            DZO_TIMED(first ? "adgd_fused_step" : "adgd_fused_retry", s);
            hipLaunchKernelGGL(adgd_fused_rosen_kernel<T>, dim3(grid), dim3(kBlock), 0, s, fp);
        }
        hipLaunchKernelGGL(adgd_decide_kernel, dim3(1), dim3(kBlock), 0, s, (const double *)c.partials(), grid, c.flag(), c.f,
                           c.dtype == DZO_F32 ? 1 : 0, c.host_dev);
        c.flag_armed = true;
        DZO_HIP(hipGetLastError());
        DZO_HIP(hipStreamSynchronize(s));
        const int32_t st = reinterpret_cast<const int32_t *>(c.host + 3)[0];
        if (st == 2) {                                                       // :128-130 (x_new == x_old bit for bit)
            DZO_TRY(give_up());
            c.is_stuck = true;
            return DZO_OK;
        }
        c.last_trials += 1;
        if (st == 1) {                                                       // :139
            const double f_new = round_to_dtype(c.dtype, c.host[0]);
            c.df = round_to_dtype(c.dtype, f_new - c.f);                     // :142-143
            c.f = f_new;                                                     // :144
            o->norm2[0] = c.host[5]; o->norm2[1] = c.host[6];
            o->norms_ready = true;
            c.dg = dg_new;                                                   // (the previous delta_gradient buffer is the next step's target)
            std::swap(c.x, o->x_twin);                                       // the trial point and its gradient are the current ones now
            std::swap(c.g, o->g_twin);
            adgd_mark_unsettled(o);
            o->fused_steps += 1;
            if (!first) o->fused_rejections += 1;
            return DZO_OK;
        }
        t = round_to_dtype(c.dtype, t * 0.5);                                // :152
        if (c.max_halvings > 0 && ++halvings >= c.max_halvings) {
            DZO_TRY(give_up());
            c.is_stuck = true;
            return DZO_OK;
        }
    }
}

static int32_t norm_blocking(OptCore &c, const void *v, double *out) {
    double ss = 0;
    DZO_TRY(dot_blocking(c.stream, c.n, c.dtype, v, v, c.partials(), c.host, &ss));
    *out = c.dtype == DZO_F32 ? (double)sqrtf((float)ss) : sqrt(ss);
    return DZO_OK;
}

static int32_t adgd_step(dzo_adgd_s *o) {
    OptCore &c = o->core;
    if (c.is_stuck) return DZO_OK;                                   // :276-278
    std::lock_guard<std::recursive_mutex> step_lock(o->mu);
    DZO_REQUIRE(c.has_objective() && c.has_gradient(), DZO_ERR_STATE,
                "step! needs objective and gradient (callbacks or a built-in problem)");
    if (c.problem && c.problem->parent) {
        problem_view_sync(c.problem);
        c.box_on = c.problem->cons_on; c.box_lo = c.problem->cons_lo; c.box_hi = c.problem->cons_hi;
    }
    const int32_t dt = c.dtype;
    const bool fused_ok = adgd_fused_ok(o);                          // (may allocate; before any state changes)
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
        if (o->norms_ready) dgn = dt == DZO_F32 ? (double)sqrtf((float)o->norm2[1]) : sqrt(o->norm2[1]);
        else DZO_TRY(norm_blocking(c, c.dg, &dgn));                  // :292
        if (dgn != 0.0) {                                            // :293
            double dxn = 0;
            if (o->norms_ready) dxn = dt == DZO_F32 ? (double)sqrtf((float)o->norm2[0]) : sqrt(o->norm2[0]);
            else DZO_TRY(norm_blocking(c, c.dx, &dxn));
            const double inv_L = round_to_dtype(dt, dxn / dgn);      // :294
            const double cap = round_to_dtype(dt, inv_sqrt_two * inv_L);
            next = next < cap ? next : cap;                          // :295
        }
    }
    o->previous_step_size = current;                                 // :298
    o->current_step_size = next;                                     // :299
    o->norms_ready = false;
    if (fused_ok) {
        bool done = false;
        DZO_DISPATCH(dt, DZO_TRY(adgd_fused_step<T>(o, next, &done)));
        if (done) {
            if (!c.is_stuck) c.iteration_count += 1;                 // :302-304, :310
            return DZO_OK;
        }
    }
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
    o->x_user = x_dev; o->g_user = g_dev; o->device = ctx().device;
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
    o->fused = getenv("DZO_TUNE_ADGD_FUSED") ? atoi(getenv("DZO_TUNE_ADGD_FUSED")) != 0 : true;
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
    rc = problem_view_create(problem, &(*out)->core.problem);    // private partial-sum workspace per optimizer
    if (rc != DZO_OK) { dzo_adgd_destroy(*out); *out = nullptr; return rc; }
    (*out)->core.box_on = problem->cons_on; (*out)->core.box_lo = problem->cons_lo; (*out)->core.box_hi = problem->cons_hi;
    return DZO_OK;
}

int32_t dzo_adgd_destroy(dzo_adgd_t o) {
    if (!o) return DZO_OK;
    if (o->core.stream && o->x_user) (void)adgd_settle(o);           // the caller's arrays end up holding the final point / gradient
    unsettled_remove(o);
    if (o->core.stream) (void)hipStreamSynchronize(o->core.stream);
    if (o->dx_buf) (void)hipFree(o->dx_buf);
    if (o->dg_buf) (void)hipFree(o->dg_buf);
    if (o->dg_alt) (void)hipFree(o->dg_alt);
    if (o->twin) (void)hipFree(o->twin);
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
    case 3: *value = o->fused_steps; break;
    case 4: *value = o->fused_rejections; break;
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
    DZO_TRY(adgd_settle(o));                                         // current_point / current_gradient ARE the caller's arrays again
    DZO_HIP(hipStreamSynchronize(o->core.stream));
    // the caller may write through the pointer: do not trust what the fused step cached about it
    if (what == 1 || what == 3) o->norms_ready = false;
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
