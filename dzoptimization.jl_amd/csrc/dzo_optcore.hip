// dzo_optcore.hip -- take_backtracking_step! (src/DZOptimization.jl:107-154) on the device.
//
// Kernels K2 (trial), K3 (accept), K4 (reject).  The reference restores the point by copy
// (:151) and re-applies axpy! (:124) on every halving; here each trial recomputes
// x = fma(t, d, x_old) from the saved point, which is the same arithmetic bit for bit and
// removes the restore pass from the loop.
#include <cstdlib>

#include "dzo_optcore.h"

#include <atomic>

namespace dzo {

template <typename T, bool VEC, bool FIRST>
__global__ __launch_bounds__(kBlock) void trial_kernel(int64_t n, T *x, T *__restrict__ backup, const T *__restrict__ d,
                                                       T t, int32_t *__restrict__ changed) {
    constexpr int N = VEC ? Vec16<T>::N : 1;
    constexpr int U = 4;
    const int64_t nthreads = (int64_t)gridDim.x * kBlock;
    bool diff = false;
    const int64_t nvec = n / N;
    const T *src = FIRST ? x : backup;
    for (int64_t base = (int64_t)blockIdx.x * kBlock * U; base < nvec; base += nthreads * U) {
        T xo[U][N], dv[U][N];
        // all loads first: x is read and written through the same pointer, so the compiler
        // cannot hoist the later loads above the earlier stores by itself
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t v = base + (int64_t)u * kBlock + threadIdx.x;
            if (v < nvec) {
                if constexpr (VEC) { load16(src + v * N, xo[u]); load16(d + v * N, dv[u]); }
                else { xo[u][0] = src[v]; dv[u][0] = d[v]; }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t v = base + (int64_t)u * kBlock + threadIdx.x;
            if (v < nvec) {
                T xn[N];
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    xn[j] = dfma(t, dv[u][j], xo[u][j]);             // :124
                    diff |= !is_equal(xn[j], xo[u][j]);              // :128
                }
                if constexpr (VEC) { store16(x + v * N, xn); if (FIRST) store16(backup + v * N, xo[u]); }   // :118
                else { x[v] = xn[0]; if (FIRST) backup[v] = xo[u][0]; }
            }
        }
    }
    if constexpr (VEC) {
        const int64_t i = nvec * N + (int64_t)blockIdx.x * kBlock + threadIdx.x;
        if (i < n) {
            const T xo = src[i];
            const T xn = dfma(t, d[i], xo);
            diff |= !is_equal(xn, xo);
            x[i] = xn;
            if (FIRST) backup[i] = xo;
        }
    }
    __shared__ int lds_flag;
    block_raise_flag(diff, changed, &lds_flag);
}

// Device-side acceptance test of take_backtracking_step! (:128, :139): lets the host enqueue the
// accepted-step tail speculatively instead of idling the GPU across a host round trip.  When
// `partials` is given the kernel also performs the objective's final fixed-order sum (one launch
// fewer per trial) and publishes f_new in result[0].
__global__ __launch_bounds__(kBlock) void decide_kernel(DecideArgs a) {
    __shared__ double lds[kWaves];
    decide_body(a, lds);
}

DecideArgs decide_args(OptCore &c, const double *partials, int64_t count, double scale) {
    c.ticket += 1.0;
    DecideArgs a;
    a.result = c.result(); a.partials = partials; a.partials2 = nullptr; a.count = count; a.scale = scale;
    a.changed = c.flag(); a.f_cur = c.f; a.to_f32 = c.dtype == DZO_F32 ? 1 : 0;
    a.status = c.status(); a.host_out = c.host_dev; a.ticket = c.ticket;
    return a;
}

// device-side :128 / :139 decision; with `partials` it also finishes the objective's sum
void launch_decide(OptCore &c, const double *partials, int64_t count, double scale) {
    hipLaunchKernelGGL(decide_kernel, dim3(1), dim3(kBlock), 0, c.stream, decide_args(c, partials, count, scale));
    c.flag_armed = true;
}

// Wait for the outcome of the last launch_decide.  The decision kernel writes straight into pinned host memory
// and stamps it with a ticket, so the host spins on that word: no event record between the decision and the gated
// kernels behind it (a barrier packet, ~5 us on the stream) and no wake-up through the runtime's signal wait.
// Every few thousand spins the stream is queried so that a failed launch cannot hang the caller.
int32_t core_wait_decision(OptCore &c) {
    DZO_TRY(wait_ticket(c.stream, c.host + 7, c.ticket));
    return wait_sealed(c.stream, c.host, 6, c.host + 6, c.ticket);   // (words 0, 1, 3, 4 carry the outcome; 2 and 5 stay zero)
}

static inline bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <typename T>
static void launch_trial(hipStream_t s, int64_t n, T *x, T *backup, const T *d, T t, bool first, int32_t *changed) {
    DZO_TIMED("lbfgs_trial", s);
    const bool vec = al16(x) && al16(backup) && al16(d);
    const int grid = stream_grid(n, (vec ? Vec16<T>::N : 1) * 4);
#define L(V, F) hipLaunchKernelGGL((trial_kernel<T, V, F>), dim3(grid), dim3(kBlock), 0, s, n, x, backup, d, t, changed)
    if (vec) { if (first) L(true, true); else L(true, false); }
    else { if (first) L(false, true); else L(false, false); }
#undef L
}

int32_t core_alloc(OptCore &c) {
    DZO_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    DZO_HIP(hipMalloc((void **)&c.ws, sizeof(double) * (2 * kMaxPartialBlocks + 16)));
    DZO_HIP(hipMemset(c.ws, 0, sizeof(double) * (2 * kMaxPartialBlocks + 16)));
    DZO_HIP(hipDeviceSynchronize());   // null-stream memset/D2D copies are asynchronous to the host and to our non-blocking streams
    DZO_HIP(hipHostMalloc((void **)&c.host, sizeof(double) * 8, hipHostMallocMapped | hipHostMallocCoherent));
    DZO_HIP(hipHostGetDevicePointer((void **)&c.host_dev, c.host, 0));
    for (int i = 0; i < 8; ++i) c.host[i] = 0;
    DZO_HIP(hipEventCreateWithFlags(&c.decided, hipEventDisableTiming));
    return DZO_OK;
}

void core_free(OptCore &c) {
    if (c.stream) (void)hipStreamSynchronize(c.stream);
    if (c.ws) (void)hipFree(c.ws);
    if (c.host) (void)hipHostFree(c.host);
    if (c.decided) (void)hipEventDestroy(c.decided);
    if (c.owns_g && c.g) (void)hipFree(c.g);
    problem_view_destroy(c.problem);                         // (no-op unless the optimizer owns a view)
    c.problem = nullptr;
    if (c.stream) (void)hipStreamDestroy(c.stream);
    c.ws = nullptr; c.host = nullptr; c.stream = nullptr;
}

int32_t core_begin_search(OptCore &c) {
    c.search_open = true;
    c.last_trials = 0;
    return DZO_OK;
}

int32_t core_trial(OptCore &c, double t, const void *dir, bool fuse_objective, int32_t *changed, double *f_new,
                   bool *f_valid) {
    hipStream_t s = c.stream;
    if (!c.flag_armed) DZO_HIP(hipMemsetAsync(c.flag(), 0, sizeof(int32_t), s));   // decide_kernel re-arms it itself
    c.flag_armed = false;
    const bool first = c.search_open;
    c.search_open = false;
    const bool fused = fuse_objective && c.problem && !c.objective && !c.constraint;
    const bool speculate = fused && c.speculative_tail != nullptr;
    const double *partials = nullptr;
    int64_t count = 0;
    double scale = 1.0;
    // Trial point and objective in one pass (dev knob, off by default).  It removes the objective
    // kernel's pass over x, but that pass hits the Infinity Cache (x was written one kernel
    // earlier), so no HBM traffic disappears: interleaved A/B at n = 1e7 gave -13 us in the small
    // kernels and +11 us in the two-loop kernels, which then absorb the write-back of the dirty
    // lines that the objective kernel used to overlap (792 vs 791 step!()/s, noise +-4).
    const bool fuse_trial = getenv("DZO_TUNE_FUSED_TRIAL") ? atoi(getenv("DZO_TUNE_FUSED_TRIAL")) != 0 : false;
    const bool trial_fused = speculate && fuse_trial && !c.box_on &&
                             problem_trial_eval_async(c.problem, s, c.x, c.dx, dir, t, first, c.flag(), &partials, &count, &scale);
    if (!trial_fused) {
        DZO_DISPATCH(c.dtype, launch_trial<T>(s, c.n, (T *)c.x, (T *)c.dx, (const T *)dir, (T)t, first, c.flag()));
        DZO_HIP(hipGetLastError());
    }
    // built-in box constraint: projection is always feasible, so it can ride in the stream (:134-135)
    if (fused && c.box_on) DZO_TRY(box_clamp_async(s, c.n, c.dtype, c.x, c.box_lo, c.box_hi));
    if (speculate) {
        if (!trial_fused && !problem_eval_partials_async(c.problem, s, c.x, &partials, &count, &scale)) {
            partials = nullptr;
            DZO_TRY(problem_eval_async(c.problem, s, c.x, c.result()));
        }
        launch_decide(c, partials, count, scale);
        DZO_HIP(hipGetLastError());
        DZO_TRY(c.speculative_tail(c.speculative_self, c.status()));   // gated kernels, enqueued blind
        DZO_TRY(core_wait_decision(c));
    } else {
        if (fused) DZO_TRY(problem_eval_async(c.problem, s, c.x, c.result()));
        // one D->H copy brings back {f_new, misc, status, flag}
        DZO_HIP(hipMemcpyAsync(c.host, c.result(), sizeof(double) * 5, hipMemcpyDeviceToHost, s));
        DZO_HIP(hipStreamSynchronize(s));
    }
    *changed = reinterpret_cast<int32_t *>(c.host + 4)[0] != 0;
    if (f_valid) *f_valid = fused;
    if (fused && f_new) *f_new = round_to_dtype(c.dtype, c.host[0]);
    return DZO_OK;
}

int32_t core_accept(OptCore &c, double f_new) {
    c.df = round_to_dtype(c.dtype, f_new - c.f);          // :142-143
    c.f = f_new;                                          // :144
    if (c.defer_delta) return DZO_OK;                     // the fused post-step kernel computes :145
    // :145  delta_point = 1*x + (-1)*delta_point  (exactly x_new - x_old)
    DZO_DISPATCH(c.dtype, launch_axpby<T>(c.stream, c.n, (T)1, (const T *)c.x, (T)-1, (T *)c.dx));
    DZO_HIP(hipGetLastError());
    return DZO_OK;
}

int32_t core_reject(OptCore &c) {
    DZO_HIP(hipMemcpyAsync(c.x, c.dx, (size_t)c.n * dtype_size(c.dtype), hipMemcpyDeviceToDevice, c.stream)); // :151
    return DZO_OK;
}

int32_t core_constraint(OptCore &c, bool *feasible) {
    if (!c.constraint) { *feasible = true; return DZO_OK; }      // isnothing(...) :134 (a built-in box was
                                                                 // already applied inside core_trial)
    DZO_HIP(hipStreamSynchronize(c.stream));
    *feasible = c.constraint(c.cb_ctx, c.x) != 0;                // :135
    return DZO_OK;
}

int32_t core_objective(OptCore &c, double *f_new) {
    if (c.objective) {
        DZO_HIP(hipStreamSynchronize(c.stream));
        *f_new = round_to_dtype(c.dtype, c.objective(c.cb_ctx, c.x));   // :138
        return DZO_OK;
    }
    DZO_REQUIRE(c.problem, DZO_ERR_STATE, "no objective callback and no built-in problem set");
    DZO_TRY(problem_eval_async(c.problem, c.stream, c.x, c.result()));
    DZO_HIP(hipMemcpyAsync(c.host, c.result(), sizeof(double), hipMemcpyDeviceToHost, c.stream));
    DZO_HIP(hipStreamSynchronize(c.stream));
    *f_new = round_to_dtype(c.dtype, c.host[0]);
    return DZO_OK;
}

int32_t core_gradient(OptCore &c) {
    if (c.gradient) {
        DZO_HIP(hipStreamSynchronize(c.stream));
        c.gradient(c.cb_ctx, c.g, c.x);                          // :479
        return DZO_OK;
    }
    DZO_REQUIRE(c.problem, DZO_ERR_STATE, "no gradient callback and no built-in problem set");
    return problem_grad_async(c.problem, c.stream, c.g, c.x);
}

int32_t core_backtracking_step(OptCore &c, double step_size, const void *dir, int trials_rejected) {
    DZO_REQUIRE(c.has_objective(), DZO_ERR_STATE, "step! needs an objective (callbacks or built-in problem)");
    int64_t halvings = 0;
    if (trials_rejected <= 0) {
        DZO_TRY(core_begin_search(c));                           // :118
    } else {
        // the caller already evaluated x_old + step_size*dir (and step_size/2, ...) OUT OF PLACE (x still holds
        // x_old) and found no decrease: continue the loop after those halvings (:151-152); the next trial is the
        // first one that writes x, so it is the one that saves x_old in delta_point (:118)
        c.search_open = true;
        c.last_trials = trials_rejected;
        for (int r = 0; r < trials_rejected; ++r) {
            step_size = round_to_dtype(c.dtype, step_size * 0.5);
            if (c.max_halvings > 0 && ++halvings >= c.max_halvings) {
                DZO_HIP(hipMemcpyAsync(c.dx, c.x, (size_t)c.n * dtype_size(c.dtype), hipMemcpyDeviceToDevice, c.stream));   // :118
                c.is_stuck = true;
                return DZO_OK;
            }
        }
    }
    for (;;) {                                                   // :121
        int32_t changed = 0;
        double f_new = 0;
        bool f_valid = false;
        DZO_TRY(core_trial(c, step_size, dir, true, &changed, &f_new, &f_valid));  // :124
        if (!changed) {                                          // :128
            c.is_stuck = true;                                   // :129
            return DZO_OK;
        }
        bool feasible = true;
        DZO_TRY(core_constraint(c, &feasible));                  // :134-135
        if (feasible) {
            if (!f_valid) DZO_TRY(core_objective(c, &f_new));    // :138
            c.last_trials += 1;
            if (f_new < c.f) {                                   // :139
                return core_accept(c, f_new);                    // :142-146
            }
        }
        step_size = round_to_dtype(c.dtype, step_size * 0.5);    // :152 (restore :151 is implicit)
        if (c.max_halvings > 0 && ++halvings >= c.max_halvings) {
            DZO_TRY(core_reject(c));
            c.is_stuck = true;
            return DZO_OK;
        }
    }
}

}  // namespace dzo

using namespace dzo;

extern "C" {

// LineSearchEvaluator call (src/DZOptimization.jl:65-92)
int32_t dzo_line_search_eval(dzo_constraint_fn constraint, dzo_objective_fn objective, dzo_gradient_fn gradient,
                             void *cb_ctx, int64_t n, int32_t dtype, const void *x_dev, double current_objective_value,
                             const void *d_dev, double overlap, double step_size, int32_t compute_gradient,
                             void *trial_point_dev, void *trial_gradient_dev, double *trial_objective_value,
                             double *improvement_ratio, double *slope_ratio) {
    DZO_TRY(require_init());
    DZO_REQUIRE(objective && x_dev && d_dev && trial_point_dev && trial_objective_value && improvement_ratio &&
                    slope_ratio && n >= 1,
                DZO_ERR_INVALID, "bad argument");
    DZO_REQUIRE(dtype == DZO_F32 || dtype == DZO_F64, DZO_ERR_INVALID, "bad dtype %d", dtype);
    // :44-54: the evaluator's arrays share one backend
    DZO_TRY(require_same_backend("LineSearchEvaluator", "src/DZOptimization.jl:44-54", x_dev, "current_point", d_dev, "step_direction"));
    DZO_TRY(require_same_backend("LineSearchEvaluator", "src/DZOptimization.jl:44-54", trial_point_dev, "trial_point", trial_gradient_dev, "trial_gradient"));
    hipStream_t s = ctx().stream;
    // :69-70  trial = x; trial += t*d   (one fused rounding per element either way)
    DZO_DISPATCH(dtype, launch_axpy_oop<T>(s, n, (T *)trial_point_dev, (T)step_size, (const T *)d_dev, (const T *)x_dev));
    DZO_HIP(hipGetLastError());
    DZO_HIP(hipStreamSynchronize(s));
    const double tmax = dtype == DZO_F32 ? 3.4028234663852886e38 : 1.7976931348623157e308;
    if (constraint && !constraint(cb_ctx, trial_point_dev)) {          // :71-79
        *trial_objective_value = tmax;
        *improvement_ratio = -tmax;                                     // typemin(T) is -Inf for floats in Julia;
        *slope_ratio = tmax;                                            // the finite extreme is used here
        return DZO_OK;
    }
    const double f_new = round_to_dtype(dtype, objective(cb_ctx, trial_point_dev));   // :80-81
    *trial_objective_value = f_new;
    *improvement_ratio = round_to_dtype(dtype, (f_new - current_objective_value) / round_to_dtype(dtype, step_size * overlap));  // :84
    if (compute_gradient) {                                             // :85-90
        DZO_REQUIRE(gradient && trial_gradient_dev, DZO_ERR_ASSERT, "@assert !isnothing(gradient_function!) (src/DZOptimization.jl:86)");
        gradient(cb_ctx, trial_gradient_dev, trial_point_dev);
        double ov = 0;
        DZO_TRY(dot_blocking(s, n, dtype, trial_gradient_dev, d_dev, ctx().scratch, ctx().host_scalar, &ov));
        *slope_ratio = round_to_dtype(dtype, round_to_dtype(dtype, ov) / overlap);
    }
    return DZO_OK;
}

}  // extern "C"
