// dzo_optcore.h -- state and kernels shared by the optimizers that use
// take_backtracking_step! (src/DZOptimization.jl:107-154): L-BFGS and AdGD.
#pragma once
#include "dzo_common.h"
#include "dzo_problems.h"

namespace dzo {

// The fields take_backtracking_step! touches (:118-152) plus the callback triple.
struct OptCore {
    int64_t n = 0;
    int32_t dtype = DZO_F64;
    hipStream_t stream = nullptr;

    dzo_constraint_fn constraint = nullptr;   // :323  (NULL = `nothing`)
    dzo_objective_fn objective = nullptr;     // :324
    dzo_gradient_fn gradient = nullptr;       // :325
    void *cb_ctx = nullptr;
    dzo_problem_s *problem = nullptr;         // built-in objective; used when callbacks are NULL
    bool box_on = false; double box_lo = 0, box_hi = 0;   // built-in UniformBoxConstraint (legacy :258-272)

    bool is_stuck = false;                    // :327
    int64_t iteration_count = 0;              // :328
    void *x = nullptr;                        // :330 current_point (aliased)
    void *dx = nullptr;                       // :331 delta_point
    double f = 0;                             // :332
    double df = 0;                            // :333
    void *g = nullptr;                        // :334 current_gradient (aliased unless owns_g)
    void *dg = nullptr;                       // :335 delta_gradient
    bool owns_g = false;

    double *ws = nullptr;         // device workspace: [partials 2*kMaxPartialBlocks][result 8]
    double *host = nullptr;       // pinned host mirror of result[8]
    double *host_dev = nullptr;   // the same buffer as the device sees it (decide_kernel writes the outcome there)
    bool flag_armed = false;      // flag() is known to be zero (decide_kernel resets it after reading)
    double ticket = 0;            // number of the last decision launched; decide_kernel writes it to host[7] after the outcome
    int64_t max_halvings = 4096;  // build-added escape from the NaN loop
    int64_t last_trials = 0;
    bool search_open = false;     // begin_search called, first trial not yet taken
    bool defer_delta = false;     // accept leaves x_old in dx; a fused kernel finishes delta_point
    // Speculative tail: when set, it is enqueued right after every trial's decision kernel and
    // BEFORE the host learns the outcome; the kernels it launches must be gated on status().
    int32_t (*speculative_tail)(void *self, const int32_t *gate) = nullptr;
    void *speculative_self = nullptr;
    hipEvent_t decided = nullptr;

    double *partials() const { return ws; }
    double *result() const { return ws + 2 * kMaxPartialBlocks; }          // [0]=f_new [1]=misc
    int32_t *flag() const { return reinterpret_cast<int32_t *>(ws + 2 * kMaxPartialBlocks + 4); }
    int32_t *status() const { return reinterpret_cast<int32_t *>(ws + 2 * kMaxPartialBlocks + 3); }  // 0 reject 1 accept 2 stuck
    bool has_objective() const { return objective != nullptr || problem != nullptr; }
    bool has_gradient() const { return gradient != nullptr || problem != nullptr; }
};

int32_t core_alloc(OptCore &c);
void core_free(OptCore &c);

// value rounded to the element type, as the reference's objective returns T
inline double round_to_dtype(int32_t dtype, double v) { return dtype == DZO_F32 ? (double)(float)v : v; }

// :118  (deferred: the first trial writes the backup while it reads x)
int32_t core_begin_search(OptCore &c);
// :124 + :128   x = fma(t, dir, x_old), *changed = !isequal(x, x_old).  If `fuse_objective`
// and the objective is a built-in problem with no constraint, f(x) is evaluated in the same
// stream submission and returned in *f_new with a single host sync.
int32_t core_trial(OptCore &c, double t, const void *dir, bool fuse_objective, int32_t *changed,
                   double *f_new, bool *f_valid);
// :142-145
int32_t core_accept(OptCore &c, double f_new);
// :151
int32_t core_reject(OptCore &c);
// :134-138 through callbacks or the built-in problem (blocking)
int32_t core_objective(OptCore &c, double *f_new);
int32_t core_constraint(OptCore &c, bool *feasible);
int32_t core_gradient(OptCore &c);
// take_backtracking_step!(opt, step_size, dir)  :107-154
int32_t core_backtracking_step(OptCore &c, double step_size, const void *dir, int trials_rejected = 0);
void launch_decide(OptCore &c, const double *partials, int64_t count, double scale);
int32_t core_wait_decision(OptCore &c);

// The decision of take_backtracking_step! (:128, :139) as kernel arguments, so that it can also run as one more
// block of another launch (the Gram reduction behind the single-pass step).
struct DecideArgs {
    double *result;            // [0] receives f_new
    const double *partials;    // objective partials to sum in fixed order (nullptr: result[0] already holds f_new)
    const double *partials2;   // optional second set (the objective at half the step, carried by the single pass): host_out[1]
    int64_t count;
    double scale;
    int32_t *changed;
    double f_cur;
    int to_f32;
    int32_t *status;
    double *host_out;          // pinned host mirror: [0] f_new, [3] status, [4] changed, [7] ticket
    double ticket;
    // L2RegularizationWrapper (legacy/DZOptimization.jl:231-232) riding on the decorated point pass: per-block partials of
    // norm2 of the trial point (and of the point at half the step); f_new = f + lambda * norm2(x) in T arithmetic, as
    // l2_finish_kernel (dzo_problems.hip) forms it.  nullptr = no L2 term.
    const double *l2_partials = nullptr, *l2_partials2 = nullptr;
    double l2_lambda = 0;
};
DecideArgs decide_args(OptCore &c, const double *partials, int64_t count, double scale);   // (takes the next ticket)

__device__ __forceinline__ void decide_body(const DecideArgs &a, double *lds) {
    double f_new;
    double f_second = 0;
    // (the host spins on what this block publishes: the change flag and both sets of partials are requested together and
    // the two sums share one pair of barriers -- block_sum_multi, bit for bit block_sum's values)
    const int32_t ch_early = threadIdx.x == 0 ? *a.changed : 0;
    if (a.partials && a.partials2) {
        __shared__ double lds2[2 * kWaves];
        double v[2] = {0, 0}, out[2];
        for (int64_t i = threadIdx.x; i < a.count; i += kBlock) { v[0] += a.partials[i]; v[1] += a.partials2[i]; }
        block_sum_multi<2>(v, lds2, out);
        f_new = a.scale * out[0];
        f_second = a.scale * out[1];
        if (a.l2_partials) {                                     // (uniform; only the decorated pass sets it)
            double w[2] = {0, 0}, ss[2];
            for (int64_t i = threadIdx.x; i < a.count; i += kBlock) { w[0] += a.l2_partials[i]; w[1] += a.l2_partials2[i]; }
            block_sum_multi<2>(w, lds2, ss);
            if (a.to_f32) {
                f_new = (double)((float)f_new + (float)a.l2_lambda * (float)ss[0]);
                f_second = (double)((float)f_second + (float)a.l2_lambda * (float)ss[1]);
            } else {
                f_new = f_new + a.l2_lambda * ss[0];
                f_second = f_second + a.l2_lambda * ss[1];
            }
        }
        if (threadIdx.x == 0) a.result[0] = f_new;
    } else {
        if (a.partials) {
            double v = 0;
            for (int64_t i = threadIdx.x; i < a.count; i += kBlock) v += a.partials[i];
            f_new = a.scale * block_sum(v, lds);
            if (threadIdx.x == 0) a.result[0] = f_new;
        } else {
            f_new = a.result[0];
        }
        if (a.partials2) {
            double v = 0;
            for (int64_t i = threadIdx.x; i < a.count; i += kBlock) v += a.partials2[i];
            f_second = a.scale * block_sum(v, lds);
        }
    }
    if (threadIdx.x == 0) {
        const double f_raw = f_new;
        if (a.to_f32) f_new = (double)(float)f_new;
        const int32_t ch = ch_early;
        int32_t st = 0;
        if (ch == 0) st = 2;                                     // :128 isequal -> stuck
        else if (f_new < a.f_cur) st = 1;                        // :139 strict decrease
        *a.status = st;
        *a.changed = 0;                                          // armed for the next trial (saves a memset launch)
        // outcome straight into the pinned host mirror: no D->H blit kernel on the critical path
        // (whole 64-bit words, and their seal in word 6: core_wait_decision checks it -- dzo_common.h, wait_sealed)
        const unsigned long long sw = (unsigned long long)(uint32_t)st, cw = (unsigned long long)(uint32_t)ch;
        // System-scope stores (sc0 sc1: written through to the host's memory at once) and no fence: the host takes the
        // outcome only when the six words, the ticket and the seal belong together, whatever order they arrive in.  (Until
        // round 4 a pair of __threadfence_system() stood here: this block is the last of its kernel to finish, and the
        // recurrence kernel of the next two-loop waits behind it.)
        unsigned long long *ho = reinterpret_cast<unsigned long long *>(a.host_out);
        auto put = [&](int i, unsigned long long w) { __hip_atomic_store(ho + i, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); };
        put(0, seal_bits(f_raw));
        put(1, seal_bits(f_second));
        put(2, 0ull); put(5, 0ull);                              // (core_wait_decision's seal covers words 0..5: the two this block does not use are
                                                                 // written too, whatever a host-driven trial's copy of result() left there)
        put(3, sw);
        put(4, cw);
        put(6, seal_bits(f_raw) ^ seal_bits(f_second) ^ sw ^ cw ^ seal_bits(a.ticket));
        put(7, seal_bits(a.ticket));                             // the host spins on this word (core_wait_decision), then checks the seal
    }
}

}  // namespace dzo
