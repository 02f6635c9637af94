/*
 * dzo.h -- C ABI of the MI355X-native BFGS / L-BFGS step!() hot path.
 *
 * This is the drop-in boundary for dzhang314/DZOptimization.jl's in-place quasi-Newton
 * step.  The reference has NO FFI: its boundary is Julia multiple
 * dispatch on the array type parameter A<:AbstractArray{T} plus three user callbacks
 * (src/DZOptimization.jl:321-325,347-356).  A Julia host reaches this library with `ccall`
 * (dzoptimization.jl_amd/julia/DZOptimizationAMD.jl; INTEGRATION.md shows the binding); the
 * Python mirror used by the tests binds the same symbols with ctypes.
 *
 * Conventions
 *   - every function returns int32_t: 0 = DZO_OK, otherwise a DZO_ERR_* code;
 *     dzo_last_error() gives the message of the calling thread's last failure.  The
 *     reference signals only through @assert (AssertionError); those map to DZO_ERR_ASSERT.
 *   - plain pointers and sizes only.  `*_dev` pointers are device (HBM) addresses, everything
 *     else is host memory.  dtype is DZO_F32 or DZO_F64; scalars cross the ABI as double.
 *   - vectors are dense, contiguous, length n (N-d Julia arrays are treated linearly, as the
 *     reference does); matrices are n x n column-major (legacy/DZOptimization.jl:746).
 *   - one HIP stream per optimizer handle; calls on one handle must be serialised by the
 *     caller, different handles may be driven from different host threads (the reference's
 *     "independent optimizers" model, README.md:12).
 *   - asynchrony: step functions return as soon as the host-side decisions of the step are
 *     known; kernels that finish the step (delta_point, gradient, delta_gradient, rho) may still
 *     be running on the handle's stream.  Every getter (get_ptr, get_rho, ...) and
 *     dzo_synchronize() waits for them; touch an optimizer's device arrays from your own
 *     streams only after one of those, or enqueue on the handle's stream (dzo_lbfgs_stream).
 *   - there is no CPU fallback: every entry point needs the HIP device selected by dzo_init.
 *
 * All citations are file:line in the reference snapshot (2025-09-05).
 */
#ifndef DZO_H
#define DZO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DZO_VERSION 100 /* 0.1.0 */

/* element types (type parameter T of the reference's optimizers) */
#define DZO_F32 0
#define DZO_F64 1

/* status codes */
#define DZO_OK 0
#define DZO_ERR_INVALID 1     /* bad argument */
#define DZO_ERR_HIP 2         /* a HIP runtime call failed */
#define DZO_ERR_ASSERT 3      /* a reference @assert would have fired */
#define DZO_ERR_NOMEM 4
#define DZO_ERR_UNSUPPORTED 5
#define DZO_ERR_STATE 6       /* entry point called out of sequence */

/* built-in synthetic objectives; the reference's user callbacks */
#define DZO_PROBLEM_ROSENBROCK2D 0      /* legacy/ExampleFunctions.jl:10-24, README.md:25-31 */
#define DZO_PROBLEM_ROSENBROCK_CHAIN 1  /* sum 100(x[i+1]-x[i]^2)^2 + (1-x[i])^2 */
#define DZO_PROBLEM_QUADRATIC 2         /* 1/2 x'Ax, dense symmetric A */
#define DZO_PROBLEM_LSE 3               /* log sum exp(x) + lambda/2 |x-c|^2 */
#define DZO_PROBLEM_QUADRATIC_CHAIN 4   /* sum 1/2 (x[i+1]-x[i])^2 + lambda/2 (x[i]-1)^2: the large-n convex quadratic (tridiagonal Hessian) */

/* how compute_lbfgs_step_direction! is executed on the device */
#define DZO_TWOLOOP_CHAIN 0 /* 2k+1 fused axpy+dot links in the reference's op order */
#define DZO_TWOLOOP_GRAM 1  /* one Gram pass + O(k^2) scalar recurrence + one combine pass */

#define DZO_LINE_SEARCH_BACKTRACKING 0 /* take_backtracking_step! (src/DZOptimization.jl:107-154), the reference */
#define DZO_LINE_SEARCH_WOLFE 1        /* strong Wolfe on the LineSearchEvaluator quotients (:65-92) */

/* dense BFGS last_step_type (legacy/DZOptimization.jl:727-731) */
#define DZO_STEP_NULL 0
#define DZO_STEP_GRADIENT_DESCENT 1
#define DZO_STEP_BFGS 2

typedef struct dzo_problem_s *dzo_problem_t;
typedef struct dzo_lbfgs_s *dzo_lbfgs_t;
typedef struct dzo_adgd_s *dzo_adgd_t;
typedef struct dzo_bfgs_s *dzo_bfgs_t;
typedef struct dzo_bfgs_batch_s *dzo_bfgs_batch_t;
typedef struct dzo_comm_s *dzo_comm_t;

/* User callbacks, in the reference's order constraint / objective / gradient
 * (src/DZOptimization.jl:323-325).  They receive DEVICE pointers and must enqueue their
 * work on the stream returned by dzo_*_stream() or synchronise themselves.
 *   constraint_function!(x)::Bool  (:71-72,134-135)  -> nonzero = feasible; NULL = `nothing`
 *   objective_function(x)::T       (:80,138)         -> value as double
 *   gradient_function!(g, x)       (:87,479)         -> return ignored */
typedef int32_t (*dzo_constraint_fn)(void *ctx, void *x_dev);
typedef double (*dzo_objective_fn)(void *ctx, const void *x_dev);
typedef void (*dzo_gradient_fn)(void *ctx, void *g_dev, const void *x_dev);

/* ---------------------------------------------------------------------------------------
 * lifecycle
 * ------------------------------------------------------------------------------------- */
/* dzo_init selects `device` for the CALLING THREAD (hipSetDevice + the library's per-device context:
 * stream, scratch) and may be called for several devices; handle-less entry points (vector primitives,
 * dzo_malloc, constructors) work on the device the calling thread selected last.  dzo_memcpy_* find the
 * device from the pointer.  Optimizer handles other than the batched one must be driven with their
 * creating device selected. */
int32_t dzo_init(int32_t device);
int32_t dzo_shutdown(void);
int32_t dzo_version(void);
const char *dzo_last_error(void);
int32_t dzo_device_info(char *name, int32_t name_len, int32_t *compute_units, int64_t *hbm_bytes);
/* the number of HIP devices the process sees; needs no dzo_init (one shard per GPU from one process: dzo_comm_init_all) */
int32_t dzo_device_count(int32_t *count);
int32_t dzo_synchronize(void);

/* Per-kernel HIP-event timing on the launching stream (bench.py's roofline leg).
 * Entry i of the table is one kernel name with its launch count and total milliseconds. */
int32_t dzo_profile_enable(int32_t level); /* 0 off, 1 roofline kernels only, 2 every kernel */
int32_t dzo_profile_reset(void);
int32_t dzo_profile_count(int32_t *count);
int32_t dzo_profile_get(int32_t i, char *name, int32_t name_len, int64_t *launches, double *total_ms);

/* Diagnostic.  Kernels hand short results to the host (a trial's outcome, the line searches' state: what the reference
 * keeps in host scalars, src/DZOptimization.jl:328-335) through pinned memory: the result words, their SEAL (the xor of
 * their bits and of a ticket), then the ticket the host spins on.  The host reads the results only once the seal
 * matches: on MI355X it has been seen to read a new ticket next to the previous publish's words (1 to 5 times in
 * 10 000 waits when they lie in different cache lines).  *count = how often a wait had to look twice, process-wide. */
int32_t dzo_unsealed_first_reads(int64_t *count);

/* Calibration (SURVEY.md 8(d) "calibrate on the box"): GB/s of a plain read-only streaming kernel over
 * `bytes` of device memory re-read `repeats` times.  <= ~200 MiB stays in the Infinity Cache (the ceiling of
 * config 2, H = 128 MiB); several GiB give the HBM streaming ceiling. */
/* Self-test: the quotient the L-BFGS recurrence forms without a division (fd_div, csrc/dzo_lbfgs.hip) against a / b on the
 * device, `pairs` operand pairs of kind `mode` (0 random, 1 special divisors, 2 exact-ish quotients, 3 quotients next to a
 * rounding boundary, 4 small integers); *mismatches must come back 0.  first4: a, b, a / b, fd_div of the first mismatch. */
int32_t dzo_selftest_fast_div(uint64_t seed, int64_t pairs, int32_t mode, int64_t *checked, int64_t *mismatches, double *first4);
int32_t dzo_calibrate_read_bandwidth(int64_t bytes, int32_t repeats, double *gbps);
/* the same reader over a device buffer of the caller's, whatever it holds */
int32_t dzo_calibrate_read_bandwidth_of(const void *buf_dev, int64_t bytes, int32_t repeats, double *gbps);

/* ---------------------------------------------------------------------------------------
 * device memory (what `similar` / `copy` / `Array(x)` do for a GPU array type A)
 * ------------------------------------------------------------------------------------- */
int32_t dzo_malloc(void **ptr_dev, int64_t bytes);
int32_t dzo_free(void *ptr_dev);
int32_t dzo_memcpy_h2d(void *dst_dev, const void *src_host, int64_t bytes);
int32_t dzo_memcpy_d2h(void *dst_host, const void *src_dev, int64_t bytes);
int32_t dzo_memcpy_d2d(void *dst_dev, const void *src_dev, int64_t bytes);

/* ---------------------------------------------------------------------------------------
 * L1 vector primitives: the method set a Julia `HipVector{T}` needs so the
 * reference's generic code runs unmodified.  Blocking where they return a host scalar.
 * ------------------------------------------------------------------------------------- */
/* y += alpha*x            LinearAlgebra.axpy!  src/DZOptimization.jl:70,124,441,448 */
int32_t dzo_axpy(int64_t n, int32_t dtype, double alpha, const void *x_dev, void *y_dev);
/* y = alpha*x + beta*y    LinearAlgebra.axpby! src/DZOptimization.jl:145,308,480 */
int32_t dzo_axpby(int64_t n, int32_t dtype, double alpha, const void *x_dev, double beta, void *y_dev);
/* x *= alpha              LinearAlgebra.rmul!  src/DZOptimization.jl:387,444 */
int32_t dzo_scal(int64_t n, int32_t dtype, double alpha, void *x_dev);
/* dst = src               Base.copy!           src/DZOptimization.jl:69,118,151,306,386,438,478 */
int32_t dzo_copy(int64_t n, int32_t dtype, const void *src_dev, void *dst_dev);
/* x .= value              Base.fill!           src/DZOptimization.jl:222,227,369,374,384 */
int32_t dzo_fill(int64_t n, int32_t dtype, double value, void *x_dev);
/* x.y                     LinearAlgebra.dot    src/DZOptimization.jl:88,440,444,447,505 */
int32_t dzo_dot(int64_t n, int32_t dtype, const void *x_dev, const void *y_dev, double *result);
/* |x|_2                   LinearAlgebra.norm   src/DZOptimization.jl:230,292,294,381 */
int32_t dzo_nrm2(int64_t n, int32_t dtype, const void *x_dev, double *result);
/* isequal(a, b)           Base.isequal         src/DZOptimization.jl:128 (NaN==NaN, -0.0!=0.0) */
int32_t dzo_isequal(int64_t n, int32_t dtype, const void *a_dev, const void *b_dev, int32_t *result);
/* dst = t*d + x           out-of-place trial point, legacy/Kernels.jl:127-135,
 *                         legacy/DZOptimization.jl:33, src/DZOptimization.jl:69-70 */
int32_t dzo_trial_point(int64_t n, int32_t dtype, void *dst_dev, double t, const void *d_dev,
                        const void *x_dev);

/* The legacy primitives that have no LinearAlgebra twin above:
 *   norm2(x)            sum of squares -- NOT its square root     legacy/Kernels.jl:49-55,139
 *   inv_norm(x)         rsqrt(norm2(x))                           legacy/Kernels.jl:141
 *   negate!(x)          x[i] = -x[i]                              legacy/Kernels.jl:76-83,143
 *   scale!(dst, a, x)   dst[i] = a*x[i], out of place             legacy/Kernels.jl:96-104
 * (in-place scale! is dzo_scal, delta! is dzo_axpby(1, x, -1, y), both axpy! forms are dzo_axpy /
 * dzo_trial_point, dot is dzo_dot.) */
int32_t dzo_norm2(int64_t n, int32_t dtype, const void *x_dev, double *result);
int32_t dzo_inv_norm(int64_t n, int32_t dtype, const void *x_dev, double *result);
int32_t dzo_negate(int64_t n, int32_t dtype, void *x_dev);
int32_t dzo_scal_oop(int64_t n, int32_t dtype, void *dst_dev, double alpha, const void *x_dev);

/* ---------------------------------------------------------------------------------------
 * built-in objectives (device-side twins of the user's callbacks)
 * ------------------------------------------------------------------------------------- */
int32_t dzo_problem_create(int32_t kind, int64_t n, int32_t dtype, const void *A_dev,
                           const void *c_dev, double lambda, dzo_problem_t *out);
int32_t dzo_problem_destroy(dzo_problem_t p);
/* Decorators of legacy/DZOptimization.jl:219-296, applied on the device:
 *   set_l2            L2RegularizationWrapper (f + lambda*norm2(x), :231-232) and L2GradientWrapper
 *                     (g += 2*lambda*x, :241-249); lambda = 0 switches it off
 *   set_box_gradient  UniformBoxGradientWrapper (:282-296): g[i] = 0 where x[i] sits on a bound
 *                     and the gradient pushes outward
 *   set_box_constraint UniformBoxConstraint (:264-272) used as the optimizers' constraint_function!
 *                     (clamp, always feasible) when the optimizer is created from this problem */
int32_t dzo_problem_set_l2(dzo_problem_t p, double lambda);
int32_t dzo_problem_set_box_gradient(dzo_problem_t p, int32_t enable, double lower_bound, double upper_bound);
int32_t dzo_problem_set_box_constraint(dzo_problem_t p, int32_t enable, double lower_bound, double upper_bound);
/* x[i] = clamp(x[i], lo, hi)   UniformBoxConstraint call, legacy/DZOptimization.jl:264-272 */
int32_t dzo_box_clamp(int64_t n, int32_t dtype, void *x_dev, double lower_bound, double upper_bound);
int32_t dzo_problem_eval(dzo_problem_t p, const void *x_dev, double *f);
int32_t dzo_problem_grad(dzo_problem_t p, void *g_dev, const void *x_dev);
/* The built-in objectives in the SHAPE of the reference's three callbacks (src/DZOptimization.jl:323-325,
 * called at :134-138 and :479): pass these function pointers with ctx = the dzo_problem_t to
 * dzo_lbfgs_create_callbacks / dzo_adgd_set_callbacks and the optimizer runs its general (callback)
 * path on a device-side objective -- what a Julia host does when its callbacks are closures over
 * dzo_problem_eval / dzo_problem_grad, without a host-language frame in the loop.  The constraint
 * callback projects into the problem's box (legacy/DZOptimization.jl:264-272) when one is set and
 * reports the point feasible; an error inside a callback is reported as +Inf / a NaN-filled gradient
 * / infeasible, since the reference's callbacks have no error channel. */
double dzo_problem_objective_cb(void *problem, const void *x_dev);
void dzo_problem_gradient_cb(void *problem, void *g_dev, const void *x_dev);
int32_t dzo_problem_constraint_cb(void *problem, void *x_dev);

/* ---------------------------------------------------------------------------------------
 * LBFGSOptimizer  (src/DZOptimization.jl:321-509)
 * ------------------------------------------------------------------------------------- */
/* Full constructor (:347-397).  ALIASES x_dev and g_dev as current_point / current_gradient
 * exactly as the reference aliases initial_point / initial_gradient (:393,:395); allocates
 * delta_point, delta_gradient (zero-filled), step_direction = -step*g/|g| and the (s, y)
 * ring.  initial_step_length <= 0 -> DZO_ERR_ASSERT (:380).  history_length must be 1..64. */
int32_t dzo_lbfgs_create(int64_t n, int32_t history_length, int32_t dtype, void *x_dev,
                         void *g_dev, double initial_objective_value,
                         double initial_step_length, dzo_lbfgs_t *out);
/* Convenience constructor (:400-427): checks the constraint, evaluates f0 and g0 through the
 * callbacks into a library-owned gradient vector, then calls the full constructor. */
int32_t dzo_lbfgs_create_callbacks(dzo_constraint_fn constraint, dzo_objective_fn objective,
                                   dzo_gradient_fn gradient, void *ctx, int64_t n,
                                   int32_t history_length, int32_t dtype, void *x_dev,
                                   double initial_step_length, dzo_lbfgs_t *out);
/* Same with a built-in problem as the callback triple (constraint = nothing). */
int32_t dzo_lbfgs_create_problem(dzo_problem_t problem, int32_t history_length, void *x_dev,
                                 double initial_step_length, dzo_lbfgs_t *out);
int32_t dzo_lbfgs_destroy(dzo_lbfgs_t opt);
int32_t dzo_lbfgs_set_callbacks(dzo_lbfgs_t opt, dzo_constraint_fn constraint,
                                dzo_objective_fn objective, dzo_gradient_fn gradient, void *ctx);
int32_t dzo_lbfgs_set_problem(dzo_lbfgs_t opt, dzo_problem_t problem);
int32_t dzo_lbfgs_set_two_loop_mode(dzo_lbfgs_t opt, int32_t mode);
int32_t dzo_lbfgs_set_max_halvings(dzo_lbfgs_t opt, int64_t max_halvings);

/* Optional safeguards, OFF by default (off = the live reference's step!).  SURVEY.md 8(f):
 *   descent_check  -- legacy/DZOptimization.jl:682-692: after the two-loop, if g.d is not
 *       finite the optimizer stops (is_stuck); if g.d >= 0 the direction is replaced by
 *       -(last_step_length/||g||) g, last_step_length = ||delta_point|| of the last step (:625-627).
 *   steepest_descent_fallback -- legacy :588-610: when the search along a quasi-Newton direction
 *       fails, retry along -(last_step_length/||g||) g; success clears the (s, y) history
 *       (_history_count[] = 0, :609), failure sets is_stuck.
 * dzo_lbfgs_get_i fields 8 / 9 / 10: history resets, descent-check replacements, kind of the
 * last step (0 quasi-Newton, 1 replaced by the descent check, 2 fallback); fields 11 / 12: steps
 * taken as one sweep over the history (single-pass step) and how many of those had their first
 * trial rejected; field 13: how many of the rejected ones were continued by the same pass at
 * t/2; field 14: layout of the history in HBM -- 0 slabs, 1 tiles of pairs, 2 tiles of points
 * (DESIGN.md "point ring"); field 15: arrangement of the tiles -- 1 tile-major, 2 stream-major
 * (informational); field 16: 1 when the pass over the point ring recomputes the points' gradients from the
 * point tiles instead of streaming them; field 17: register sets per wave of that pass (1 = two waves per
 * SIMD, 2 = one), 0 when the optimizer is not on the point ring (both informational); field 18: steps that first
 * compared the aliased arrays with the point ring because a pointer to them had been handed out (dzo_lbfgs_read hands
 * out none);
 * dzo_lbfgs_get_s field 2: last_step_length. */
int32_t dzo_lbfgs_set_safeguards(dzo_lbfgs_t opt, int32_t descent_check, int32_t steepest_descent_fallback);

/* Line search used by step!: DZO_LINE_SEARCH_BACKTRACKING (reference, default) or
 * DZO_LINE_SEARCH_WOLFE -- the consumer of the quotients the reference's LineSearchEvaluator
 * defines but never uses (src/DZOptimization.jl:84 improvement_ratio, :88-89 slope_ratio):
 * bisection / doubling from t = 1 until improvement_ratio >= c1 and |slope_ratio| <= c2, at most
 * max_evals evaluator calls (each = objective + gradient at the trial point).  Guarantees
 * delta_point . delta_gradient > 0 for every pushed pair.  The accepted trial gradient becomes
 * current_gradient (no extra gradient call).  c1, c2, max_evals <= 0 keep the defaults
 * (1e-4, 0.9, 40). */
int32_t dzo_lbfgs_set_line_search(dzo_lbfgs_t opt, int32_t kind, double c1, double c2, int32_t max_evals);

/* step!(opt) (:454-509) -- the whole step, callbacks invoked from inside. */
int32_t dzo_lbfgs_step(dzo_lbfgs_t opt);

/* Split entry points for hosts that drive the loop themselves (the objective / gradient
 * callbacks run between them, :138 and :479):
 *   direction      compute_lbfgs_step_direction! (:430-451), K1
 *   begin_search   copy!(delta_point, current_point) (:118)
 *   trial          axpy!(t, d, x) from the saved point + isequal test (:124,:128), K2
 *   accept         delta_f, f, delta_point = x - x_old (:142-145), K3
 *   reject         copy!(x, delta_point) (:151), K4
 *   pre_gradient   copy!(delta_gradient, g) (:478)
 *   post_gradient  delta_gradient = g - delta_gradient, ring push, rho, count (:480-507), K5+K6
 * dzo_lbfgs_direction only enqueues its kernels on the optimizer's stream (like a KernelAbstractions
 * launch); dzo_lbfgs_get_ptr(4) or dzo_synchronize waits for step_direction. */
int32_t dzo_lbfgs_direction(dzo_lbfgs_t opt);
int32_t dzo_lbfgs_begin_search(dzo_lbfgs_t opt);
int32_t dzo_lbfgs_trial(dzo_lbfgs_t opt, double step_size, int32_t *changed);
int32_t dzo_lbfgs_accept(dzo_lbfgs_t opt, double next_objective_value);
int32_t dzo_lbfgs_reject(dzo_lbfgs_t opt);
int32_t dzo_lbfgs_pre_gradient(dzo_lbfgs_t opt);
int32_t dzo_lbfgs_post_gradient(dzo_lbfgs_t opt);

/* State (all of it is public in the reference, README.md:11).
 * get_i:   0 is_stuck  1 iteration_count  2 n  3 history_length  4 length(history)
 *          5 objective evaluations in the last step  6 two-loop mode  7 dtype
 * get_s:   0 current_objective_value  1 delta_objective_value
 * get_ptr: 0 current_point  1 delta_point  2 current_gradient  3 delta_gradient
 *          4 step_direction  5 delta_point_history[idx]  6 delta_gradient_history[idx]
 *          (idx 0 = newest, the reference's index 1)
 * Validity: 0 and 2 are the arrays the constructor was given (aliased, :393) and stay valid for the
 * optimizer's life; a get_ptr / dzo_synchronize / dzo_memcpy_* call settles the live copy into them.
 * 1, 3 and 4 are valid until the next step.  An optimizer made by dzo_lbfgs_create_problem on the
 * built-in chained Rosenbrock with history_length <= 20 keeps its history tile-major in HBM (DESIGN.md,
 * "blocked ring" / "point ring"): for it 1, 3, 4, 5 and 6 return contiguous COPIES formed by the call, valid
 * until the next step -- read-only; install pairs with dzo_lbfgs_set_history.  Every other optimizer
 * returns the live vectors. */
int32_t dzo_lbfgs_get_i(dzo_lbfgs_t opt, int32_t what, int64_t *value);
int32_t dzo_lbfgs_get_s(dzo_lbfgs_t opt, int32_t what, double *value);
int32_t dzo_lbfgs_set_s(dzo_lbfgs_t opt, int32_t what, double value);
int32_t dzo_lbfgs_set_stuck(dzo_lbfgs_t opt, int32_t is_stuck);
int32_t dzo_lbfgs_get_ptr(dzo_lbfgs_t opt, int32_t what, int32_t idx, void **ptr_dev);
/* Field `what` (get_ptr's numbering; n elements of the optimizer's dtype) copied to host memory, synchronously.  Unlike
 * get_ptr this hands out no pointer: the host cannot have written into current_point / current_gradient through it, so the
 * step behind a read does not have to check the aliased arrays against the optimizer's own state (after get_ptr,
 * dzo_synchronize or dzo_memcpy_* it does: the reference's step! walks from whatever those arrays hold, :393).  The way to
 * watch an optimization from the host. */
int32_t dzo_lbfgs_read(dzo_lbfgs_t opt, int32_t what, int32_t idx, void *host_dst);
/* rho_history / alpha_history (:341-342), newest first; *count receives the length */
int32_t dzo_lbfgs_get_rho(dzo_lbfgs_t opt, double *out, int32_t capacity, int32_t *count);
int32_t dzo_lbfgs_get_alpha(dzo_lbfgs_t opt, double *out, int32_t capacity, int32_t *count);
/* Install k pairs (rows of S_dev / Y_dev, k x n row-major, row 0 newest) and, optionally,
 * their rho = s.y values; used for checkpoint/resume and frozen-state parity tests. */
int32_t dzo_lbfgs_set_history(dzo_lbfgs_t opt, int32_t k, const void *S_dev, const void *Y_dev,
                              const double *rho_or_null, int64_t iteration_count);
int32_t dzo_lbfgs_stream(dzo_lbfgs_t opt, void **hip_stream);

/* ---------------------------------------------------------------------------------------
 * LineSearchEvaluator call (src/DZOptimization.jl:65-92): trial point x + t*d, objective,
 * Armijo quotient (:84) and, optionally, trial gradient + curvature quotient (:85-90).
 * ------------------------------------------------------------------------------------- */
int32_t dzo_line_search_eval(dzo_constraint_fn constraint, dzo_objective_fn objective,
                             dzo_gradient_fn gradient, void *ctx, int64_t n, int32_t dtype,
                             const void *x_dev, double current_objective_value,
                             const void *d_dev, double overlap, double step_size,
                             int32_t compute_gradient, void *trial_point_dev,
                             void *trial_gradient_dev, double *trial_objective_value,
                             double *improvement_ratio, double *slope_ratio);

/* ---------------------------------------------------------------------------------------
 * AdGDOptimizer (src/DZOptimization.jl:179-312) -- SURVEY.md 8(f) rank 1
 * ------------------------------------------------------------------------------------- */
int32_t dzo_adgd_create(int64_t n, int32_t dtype, void *x_dev, void *g_dev,
                        double initial_objective_value, double initial_step_length,
                        dzo_adgd_t *out);
int32_t dzo_adgd_create_problem(dzo_problem_t problem, void *x_dev, double initial_step_length,
                                dzo_adgd_t *out);
int32_t dzo_adgd_destroy(dzo_adgd_t opt);
int32_t dzo_adgd_set_callbacks(dzo_adgd_t opt, dzo_constraint_fn constraint,
                               dzo_objective_fn objective, dzo_gradient_fn gradient, void *ctx);
int32_t dzo_adgd_step(dzo_adgd_t opt);
/* With the built-in chained Rosenbrock objective step! is ONE pass over x and g (first trial, objective,
 * gradient, both deltas and the two norms the next step's :292-294 need) that reads x and g and writes
 * the trial point and its gradient into twin buffers (6 n T of traffic); after a rejected trial the pass
 * is repeated at half the step from the untouched x and g (:151-152).
 * get_i: 0 is_stuck 1 iteration_count 2 n 3 fused steps 4 of them after a rejected first trial
 *        5 passes that were already in flight when their step! was called (the next pass is enqueued behind
 *        every decision, its step size from the device-side evaluation of :285-299 / :152)
 *        6 passes in flight that were dropped (a pointer was handed out, an option changed)
 *        7 adopted passes whose device-side step size differed from the host's evaluation (expected 0)
 *        8 steps that took the generic kernels because the caller's current_gradient array (which the optimizer
 *          aliases, src/DZOptimization.jl:216-239, and step! walks along, :301) no longer held the gradient of the
 *          current point when the step began -- the host wrote into it, or passed its own initial_gradient;
 * get_s: 0 f 1 delta_f 2 current_step_size 3 previous_step_size;
 * get_ptr: 0 x 1 delta_point 2 g 3 delta_gradient.  0 and 2 are the constructor's arrays (aliased, :261)
 * for the optimizer's life: get_ptr / dzo_synchronize / dzo_memcpy_* settle the live copy into them.
 * 1 and 3 are valid until the next step (delta_gradient alternates between two buffers). */
int32_t dzo_adgd_get_i(dzo_adgd_t opt, int32_t what, int64_t *value);
int32_t dzo_adgd_get_s(dzo_adgd_t opt, int32_t what, double *value);
int32_t dzo_adgd_get_ptr(dzo_adgd_t opt, int32_t what, void **ptr_dev);
/* as dzo_lbfgs_read: field `what` to host memory without handing out a pointer */
int32_t dzo_adgd_read(dzo_adgd_t opt, int32_t what, void *host_dst);

/* ---------------------------------------------------------------------------------------
 * BFGSOptimizer (legacy/DZOptimization.jl:733-994; README.md:33-41)
 * ------------------------------------------------------------------------------------- */
/* Constructor (:762-810): COPIES x0 (:769), evaluates f0 / g0, H0 = I (:783), d0 = g (:784),
 * last_step_length = initial_step_length (:779).  Argument order objective, gradient,
 * constraint follows :762-766.  constraint may be NULL (NULL_CONSTRAINT, :759). */
int32_t dzo_bfgs_create_callbacks(dzo_objective_fn objective, dzo_gradient_fn gradient,
                                  dzo_constraint_fn constraint, void *ctx, int64_t n,
                                  int32_t dtype, const void *x0_dev, double initial_step_length,
                                  dzo_bfgs_t *out);
int32_t dzo_bfgs_create_problem(dzo_problem_t problem, const void *x0_dev,
                                double initial_step_length, dzo_bfgs_t *out);
/* Re-precision constructors BFGSOptimizer(::Type{T}, opt) / (::Type{T}, f, g!, c!, opt)
 * (legacy/DZOptimization.jl:812-862): a NEW optimizer of the target element type that continues
 * from `src`: x, H, delta_point, delta_gradient are converted elementwise (T.(..)), f and g are
 * re-evaluated in the new precision, d = H*g is recomputed (:833-836), iteration_count /
 * last_step_length / last_step_type carry over, has_terminated restarts as false (:849).
 * The target type is the new problem's dtype, or `dtype` for the callback form. */
int32_t dzo_bfgs_convert_problem(dzo_bfgs_t src, dzo_problem_t problem_of_target_dtype, dzo_bfgs_t *out);
int32_t dzo_bfgs_convert_callbacks(dzo_bfgs_t src, int32_t dtype, dzo_objective_fn objective,
                                   dzo_gradient_fn gradient, dzo_constraint_fn constraint, void *ctx,
                                   dzo_bfgs_t *out);
int32_t dzo_bfgs_destroy(dzo_bfgs_t opt);
/* step!(opt) (:891-994): competitive quadratic line searches, accept / reset / terminate. */
int32_t dzo_bfgs_step(dzo_bfgs_t opt);
/* update_inverse_hessian! (:864-889) on raw device arrays: rescales d in place (:874),
 * t = H*dg into scratch (:875), rank-2 update of H (:878-886).  If g_dev and d_next_dev are
 * non-NULL the next direction d_next = H_new*g (:958-960) is produced in the same pass. */
int32_t dzo_bfgs_update(int64_t n, int32_t dtype, void *H_dev, double step_length, void *d_dev,
                        const void *dg_dev, void *scratch_dev, const void *g_dev,
                        void *d_next_dev);
/* The same update with the rank-2 term on the matrix cores (v_mfma_f64_16x16x4_f64, K = 2 padded
 * to 4): fp64, n % 16 == 0, H only (no fused direction).  A measured alternative, NOT the default:
 * the kernel is HBM-bound either way, rounding follows an fma chain instead of :882-884 and H loses
 * bit-exact symmetry. */
int32_t dzo_bfgs_update_mfma(int64_t n, int32_t dtype, void *H_dev, double step_length, void *d_dev,
                             const void *dg_dev, void *scratch_dev);
/* out = H*v for symmetric H (mul!, :875,:958-960) */
int32_t dzo_symv(int64_t n, int32_t dtype, const void *H_dev, const void *v_dev, void *out_dev);
/* quadratic_line_search(functor, f0, t0) as defined in DESIGN.md from
 * find_three_point_bracket (:49-172) + QuadraticLineSearch (:191-216); direction 0 = -g, 1 = -d */
int32_t dzo_bfgs_line_search(dzo_bfgs_t opt, int32_t use_gradient_direction, double t0,
                             double *t_best, double *f_best);
int32_t dzo_bfgs_set_max_increases(dzo_bfgs_t opt, int32_t max_increases);
/* approximate_inverse_hessian <- I (identity_matrix!, :712-720) and next_step_direction <- gradient:
 * the reset step! performs itself after a gradient-descent step (:981-986), for hosts that restart. */
int32_t dzo_bfgs_reset(dzo_bfgs_t opt);
/* get_i: 0 has_terminated (== has_converged, README.md:38) 1 iteration_count 2 n
 *        3 last_step_type 4 objective evaluations so far
 * get_s: 0 current_objective_value 1 last_step_length
 * get_ptr: 0 current_point 1 delta_point 2 current_gradient 3 delta_gradient
 *          4 next_step_direction 5 approximate_inverse_hessian 6 scratch */
int32_t dzo_bfgs_get_i(dzo_bfgs_t opt, int32_t what, int64_t *value);
int32_t dzo_bfgs_get_s(dzo_bfgs_t opt, int32_t what, double *value);
int32_t dzo_bfgs_get_ptr(dzo_bfgs_t opt, int32_t what, void **ptr_dev);
/* Install state ("save/load data in the middle of optimization", README.md:11; the re-precision
 * constructor :812-862 moves a whole optimizer the same way).  The device arrays (x, g, delta_point,
 * delta_gradient, next_step_direction, H) are the optimizer's state itself: write them through the
 * pointers of dzo_bfgs_get_ptr (dzo_memcpy_h2d / d2d) between steps.  The host-side fields:
 *   set_s: 0 current_objective_value (NaN -> DZO_ERR_ASSERT, :773)  1 last_step_length
 *          2 delta_objective_value (gradient-descent handles)
 *   set_i: 0 has_terminated  1 iteration_count  3 last_step_type  4 objective evaluations so far */
int32_t dzo_bfgs_set_s(dzo_bfgs_t opt, int32_t what, double value);
int32_t dzo_bfgs_set_i(dzo_bfgs_t opt, int32_t what, int64_t value);

/* ---------------------------------------------------------------------------------------
 * Legacy GradientDescentOptimizer (legacy/DZOptimization.jl:305-449) with
 * QuadraticLineSearch() (:181-216) as its line_search_function!.  Argument order constraint,
 * objective, gradient follows :330-337.  The handle type and the getters are the BFGS ones
 * (dzo_bfgs_get_i / get_s / get_ptr / destroy; get_s field 2 = delta_objective_value, get_ptr
 * field 4 = next_step_direction = -last_step_length * g/|g|, field 5 is NULL: no Hessian).
 * ------------------------------------------------------------------------------------- */
int32_t dzo_gd_create_callbacks(dzo_constraint_fn constraint, dzo_objective_fn objective,
                                dzo_gradient_fn gradient, void *ctx, int64_t n, int32_t dtype,
                                const void *x0_dev, double initial_step_length, dzo_bfgs_t *out);
int32_t dzo_gd_create_problem(dzo_problem_t problem, const void *x0_dev, double initial_step_length,
                              dzo_bfgs_t *out);
/* step!(opt) (:393-449) */
int32_t dzo_gd_step(dzo_bfgs_t opt);

/* ---------------------------------------------------------------------------------------
 * Batched dense BFGS: B independent BFGSOptimizer instances on one device, one workgroup per
 * instance, the whole step (both line searches included) on the device.  "run multiple
 * optimizers in parallel" (README.md:12); instances shard over the GPUs of a node with no data-path
 * collective, only the convergence flag is all-reduced (dzo_comm_* below).  The step kernel keeps
 * the LOWER triangle of every inverse Hessian only (H is symmetric bit for bit); dzo_bfgs_batch_get_ptr
 * mirrors it before handing out H.
 * ------------------------------------------------------------------------------------- */
int32_t dzo_bfgs_batch_create(int32_t problem_kind, int64_t batch, int64_t n, int32_t dtype,
                              const void *x0_dev /* batch x n row-major */,
                              double initial_step_length, dzo_bfgs_batch_t *out);
/* From a problem handle: chained Rosenbrock, or the dense quadratic 1/2 x'Ax with ONE matrix A shared by
 * every instance (one matrix per instance: dzo_bfgs_batch_create_problem_matrices below); the decorators set on the handle (dzo_problem_set_l2 / set_box_gradient /
 * set_box_constraint, legacy/DZOptimization.jl:219-296) are applied inside the step kernel.  device < 0:
 * the device the calling thread selected (dzo_init). */
int32_t dzo_bfgs_batch_create_problem(dzo_problem_t problem, int64_t batch, const void *x0_dev,
                                      double initial_step_length, int32_t device, dzo_bfgs_batch_t *out);
/* The same with a DIFFERENT quadratic per instance ("run multiple optimizers in parallel", README.md:12, each with its own
 * objective): the handle gives kind (DZO_PROBLEM_QUADRATIC), n, dtype and the decorators; instance b minimises
 * 1/2 x'A_b x with A_b the symmetric n x n column-major matrix at matrices_dev + b * matrix_stride ELEMENTS
 * (matrix_stride >= n*n).  The caller owns matrices_dev and keeps it alive and unchanged while the batch exists. */
int32_t dzo_bfgs_batch_create_problem_matrices(dzo_problem_t problem, int64_t batch, const void *matrices_dev,
                                               int64_t matrix_stride, const void *x0_dev,
                                               double initial_step_length, int32_t device, dzo_bfgs_batch_t *out);
/* QuadraticLineSearch.max_increases (legacy/DZOptimization.jl:181-188, :138-151) of every instance; 0 = no cap */
int32_t dzo_bfgs_batch_set_max_increases(dzo_bfgs_batch_t b, int32_t max_increases);
int32_t dzo_bfgs_batch_destroy(dzo_bfgs_batch_t b);
/* runs `steps` synchronous step! calls on every live instance; *all_done = 1 when every
 * instance has_terminated.  Does not block unless all_done is non-NULL. */
int32_t dzo_bfgs_batch_step(dzo_bfgs_batch_t b, int32_t steps, int32_t *all_done);
/* get_ptr: 0 x (B x n) 1 g 2 H (B x n x n col-major) 3 f (B doubles) 4 has_terminated (B int32)
 *          5 iteration_count (B int64) 6 delta_point 7 delta_gradient 8 d 9 last_step_length
 *          10 last_step_type (B int32) */
int32_t dzo_bfgs_batch_get_ptr(dzo_bfgs_batch_t b, int32_t what, void **ptr_dev);
int32_t dzo_bfgs_batch_count_active(dzo_bfgs_batch_t b, int64_t *active);
/* The arrays behind dzo_bfgs_batch_get_ptr ARE the state of the instances (x, g, H, d, delta_point,
 * delta_gradient, f, last_step_length, last_step_type, has_terminated, iteration_count): writing them
 * between two dzo_bfgs_batch_step calls installs state (checkpoint / resume, README.md:11; the per-step
 * parity tests upload the CPU reference's state that way). */

/* The same constructor on an explicit device, for a host that drives the shards of several GPUs from
 * one process (dzo_comm_init_all).  x0_dev must live on `device`.  Every entry point of a batched
 * handle enters the handle's device by itself. */
int32_t dzo_bfgs_batch_create_on(int32_t device, int32_t problem_kind, int64_t batch, int64_t n, int32_t dtype,
                                 const void *x0_dev, double initial_step_length, dzo_bfgs_batch_t *out);
int32_t dzo_bfgs_batch_device(dzo_bfgs_batch_t b, int32_t *device);

/* ---------------------------------------------------------------------------------------
 * The one collective: the global convergence flag of sharded independent optimizers
 * (block partition of instances over the GPUs of a node, no data-path collective;
 * "run multiple optimizers in parallel", README.md:12).  all-reduce(MIN) of one int32 per rank over
 * RCCL / xGMI.  RCCL is loaded with dlopen at first use (no link-time dependency).
 *   dzo_comm_init_all   one process, several devices (ncclCommInitAll): local rank i = devices[i]
 *   dzo_comm_unique_id + dzo_comm_init_rank
 *                       one process per GPU: rank 0 creates the 128-byte id, the launcher carries
 *                       it to the other ranks, every rank joins with the device it selected (dzo_init)
 *   dzo_flag_allreduce_min   local_flags: one int32 per LOCAL rank; blocking
 *   dzo_flag_allreduce_min_n the same with the number of flags passed: DZO_ERR_INVALID unless it equals the
 *                       communicator's local rank count (what bindings with sized arrays should call)
 *   dzo_bfgs_batch_all_done  counts the live instances of every local shard (concurrently), then one
 *                       all-reduce: *all_done = 1 when every instance of every shard has_terminated.
 *                       comm may be NULL (no collective); otherwise batches[i] must live on the
 *                       device of local rank i.
 * ------------------------------------------------------------------------------------- */
#define DZO_COMM_UNIQUE_ID_BYTES 128
int32_t dzo_comm_unique_id(void *id128);
int32_t dzo_comm_init_rank(const void *id128, int32_t nranks, int32_t rank, dzo_comm_t *out);
int32_t dzo_comm_init_all(const int32_t *devices, int32_t ndev, dzo_comm_t *out);
int32_t dzo_comm_destroy(dzo_comm_t comm);
/* any of the outputs may be NULL; collectives = all-reduces issued so far */
int32_t dzo_comm_info(dzo_comm_t comm, int32_t *nranks, int32_t *nlocal, int32_t *first_rank, int64_t *collectives);
int32_t dzo_flag_allreduce_min(dzo_comm_t comm, const int32_t *local_flags, int32_t *global_flag);
int32_t dzo_flag_allreduce_min_n(dzo_comm_t comm, const int32_t *local_flags, int32_t nflags, int32_t *global_flag);
int32_t dzo_bfgs_batch_all_done(dzo_comm_t comm_or_null, const dzo_bfgs_batch_t *batches, int32_t nbatches,
                                int32_t *all_done);

#ifdef __cplusplus
}
#endif
#endif /* DZO_H */
