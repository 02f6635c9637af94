// dzo_adgd.hip -- AdGDOptimizer + step! (src/DZOptimization.jl:179-312), SURVEY.md 8(f).1.
// Reuses the backtracking kernels of dzo_optcore.hip; adds only two nrm2 reductions.
#include "dzo_optcore.h"

#include "dzo_problems.h"
#include "dzo_rosen.h"

namespace dzo {
// Device-side twin of the step's scalar state: what the NEXT pass needs when it is enqueued before the host has
// seen the outcome of this one (see "pipelined step" below).  Written by adgd_seed_kernel (from the host's values)
// and advanced by adgd_decide_kernel.
struct AdgdDev {
    double t;                        // step size of the next pass (first trial or halved retry)
    double f_cur;                    // current_objective_value
    double prev, cur;                // previous_step_size / current_step_size as step! leaves them (:298-299)
    int32_t sel;                     // the pass reads point / gradient buffer sel and writes buffer (sel + 1) % 3
    int32_t halvings;
    int32_t stuck;                   // the search ended (unchanged point, halving limit): later passes do nothing
    int32_t seq;                     // number of the pass this state belongs to
};
// The rounding rule of the optimizer's element type and the halving limit: what the decision needs besides the sums.
struct AdgdRule {
    int32_t to_f32;
    double inv_sqrt_two;
    int64_t max_halvings;
};
}  // namespace dzo

struct dzo_adgd_s {
    dzo::OptCore core;
    double current_step_size = 0;    // :195
    double previous_step_size = 0;   // :196
    void *dx_buf = nullptr, *dg_buf = nullptr;
    // fused step (built-in chained Rosenbrock): one pass does :301 (first trial), :307 and the two sums of
    // squares that :292 / :294 of the NEXT step need -- 2 reads + 2 writes per element.  The pass never writes
    // x or g: point and gradient live in THREE buffer pairs used in rotation (0 = the arrays the optimizer
    // aliases, :253-254 of the reference constructor); a pass reads pair `cur` and writes pair cur + 1, an
    // accepted trial makes that the current pair.  Nothing has to be backed up or restored (:118, :151), and
    // delta_point = x - x_old (:145), delta_gradient = g - g_old (:306, :308) are not written by the pass at all:
    // both operands stay intact in pairs cur and cur - 1 (the pass in flight writes pair cur + 1), so the two
    // vectors are formed when somebody asks for them (get_ptr, or a step on the generic kernels).  The caller's
    // arrays are settled whenever the host looks (get_ptr, dzo_synchronize, dzo_memcpy_*, destroy).
    bool fused = true;               // DZO_TUNE_ADGD_FUSED=0 forces the generic kernel sequence
    void *twin = nullptr;
    void *xbuf[3] = {nullptr, nullptr, nullptr}, *gbuf[3] = {nullptr, nullptr, nullptr};
    int cur = 0;                     // core.x == xbuf[cur], core.g == gbuf[cur]
    bool dx_lazy = false, dg_lazy = false;   // delta_point / delta_gradient = pair cur - pair (cur + 2) % 3, not formed yet
    bool g_valid[3] = {true, true, true};    // gbuf[i] holds the gradient of xbuf[i] (the one-pass step writes points only)
    void *x_user = nullptr, *g_user = nullptr;
    bool unsettled = false;
    // The fused pass walks along the gradient it RECOMPUTES from the point; the reference walks along
    // opt.current_gradient, which is the caller's array (:216-239, :301).  Whenever the host may have written into that
    // array -- the initial_gradient it passed to the constructor, or anything after a settle handed the arrays back --
    // the next step first checks that it still is the gradient of the point, and takes the generic kernels (which read
    // the array) when it is not.  (the L-BFGS point ring has lbfgs_adopt_host_writes for the same rule.)
    bool g_host_may_write = true;
    int64_t host_gradient_steps = 0; // steps that took the generic kernels because of it
    int device = 0;
    std::recursive_mutex mu;
    bool norms_ready = false;        // |delta_point|^2, |delta_gradient|^2 of the last step are in norm2[]
    double norm2[2] = {0, 0};
    int64_t fused_steps = 0, fused_rejections = 0;
    // Pipelined step: behind every pass the NEXT pass is enqueued at once; it takes its step size and buffer roles
    // from the device-side state (the decision runs the recurrence of :285-299 / the halving of :152 on the
    // device), so the GPU never waits for the host's round trip.  Round 4: the decision on pass s is no kernel of
    // its own any more -- every block of pass s + 1 evaluates it in its prologue (the same fixed-order sums of the
    // partials of pass s, so every block arrives at the same bits), block 0 stores the new state and publishes
    // the outcome to the host; the pass, the state and the changed-flag rotate through 2 / 2 / 3 buffers so that
    // nothing a prologue reads is written by the pass it belongs to.  A stand-alone decision kernel remains for
    // the pass behind which nothing is enqueued (DZO_TUNE_ADGD_PIPELINE=0) and for DZO_TUNE_ADGD_PROLOGUE=0, the
    // round-3 sequence pass, decision, pass, decision.  Between two step! calls exactly one pass is in
    // flight and it writes only buffers the current state does not live in: a getter that hands out a pointer or
    // a changed option just drains the stream and the next pass starts from the host's state.  The host evaluates
    // the same recurrence and checks the value the device used (spec_corrected counts disagreements: none, both
    // sides evaluate the same IEEE expressions).
    bool pipeline = true;            // DZO_TUNE_ADGD_PIPELINE=0: one host round trip per pass
    bool nt_stores = true;           // DZO_TUNE_ADGD_NT_STORES=0: plain stores of the trial point / gradient (measured: no difference)
    bool prologue = true;            // DZO_TUNE_ADGD_PROLOGUE=0: a decision kernel behind every pass
    dzo::AdgdDev *dev = nullptr;     // [2] states (pass s uses dev[s & 1]) + 4 int32 changed-flags behind them (pass s raises flag s % 3)
    double *pass_partials = nullptr; // [2][3][kMaxPartialBlocks] partial sums of the fused pass (pass s writes set s & 1)
    int pass_bpc = 4;                // DZO_TUNE_ADGD_BPC: blocks of the fused pass per CU (measured: 4 / 6 / 8 -> 22.0 / 21.4 / 20.4 k step!()/s at n = 1e7)
    double *slots = nullptr, *slots_dev = nullptr;   // pinned: 2 x 8 doubles, outcome of the last two decisions (6 words + their seal)
    bool spec_pending = false;       // a pass is enqueued whose outcome the host has not consumed
    int spec_slot = 0;
    int32_t seq = 0;                 // decisions enqueued so far (mod 2 = slot of the next one)
    int64_t spec_adopted = 0, spec_discarded = 0, spec_corrected = 0;
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
// 1 read + 1 write per element: g_old is the 3-point stencil of x_old, which the wave-row already holds, so it is
// recomputed (rosen_grad_elem: the function that produced the stored gradient, so the same bits) instead of read, and
// g_new is not written -- no pass reads a gradient array any more; the host forms current_gradient / delta_gradient from
// the point buffers when somebody asks (adgd_ensure_g).  The halo lanes lack their outer neighbour and do not need it:
// lane 1 takes only the LAST element of lane 0's trial point and lane 62 the FIRST of lane 63's, and the stencils of
// those inner elements lie inside the row.  (Round 2's pass read and wrote the gradients: 4 n T.)
constexpr int kAdgdOwn = 62;

template <typename T> struct AdgdFusedParams {
    int64_t n;
    AdgdDev *st;                               // [2]: the state of pass s is st[s & 1]
    T *x0, *g0, *x1, *g1, *x2, *g2;            // the three pairs
    double *partials;                          // [2][3][kMaxPartialBlocks]: objective, |dx|^2, |dg|^2 of pass s in set s & 1
    int32_t *flags;                            // [3]: pass s raises flags[s % 3] when the trial point differs (:128)
    double *slots;                             // [2][8] pinned: the outcome of pass s goes to slot s & 1
    AdgdRule rule;
    int32_t seq;                               // the number of this pass
    int32_t mode;                              // 0: decide pass seq - 1 first (prologue); 1: st[seq & 1] is ready
    int nt_stores;                             // DZO_TUNE_ADGD_NT_STORES
};

constexpr int64_t kAdgdPartialSet = 3 * (int64_t)kMaxPartialBlocks;

__device__ __forceinline__ double adgd_pack2(int32_t lo, int32_t hi) {
    return __longlong_as_double((long long)(((unsigned long long)(uint32_t)hi << 32) | (unsigned long long)(uint32_t)lo));
}

// The :128 / :139 decision on the pass that ran from state `st` (v = its three sums, ch = its changed flag) and the
// state of the pass behind it: after an accepted trial the step-size recurrence of :285-299 (expression by expression
// as adgd_step evaluates it on the host, which checks the value before it adopts the pass), after a rejected one half
// the step (:152).  status 3: the search was over before that pass, it did nothing.
__device__ __forceinline__ AdgdDev adgd_next_state(const AdgdDev &st, const double (&v)[3], int32_t ch, const AdgdRule &r, int32_t *status_out) {
    AdgdDev nx = st;
    nx.seq = st.seq + 1;
    if (st.stuck) { *status_out = 3; return nx; }
    auto rnd = [&](double x) { return r.to_f32 ? (double)(float)x : x; };
    auto root = [&](double x) { return r.to_f32 ? (double)sqrtf((float)x) : sqrt(x); };
    const double f_new = rnd(v[0]);
    int32_t status = 0;
    if (ch == 0) status = 2;
    else if (f_new < st.f_cur) status = 1;
    *status_out = status;
    if (status == 1) {
        const double previous = st.prev, current = st.cur;                   // :285-286 of the next step!
        double next = current;                                               // :287
        if (previous != 0.0) {                                               // (:289 asserts it; the host raises the error)
            const double theta = rnd(current / previous);                    // :290
            next = rnd(next * root(1.0 + theta));                            // :291
            const double dgn = root(v[2]);                                   // :292
            if (dgn != 0.0) {                                                // :293
                const double inv_L = rnd(root(v[1]) / dgn);                  // :294
                const double cap = rnd(r.inv_sqrt_two * inv_L);
                next = next < cap ? next : cap;                              // :295
            }
        } else {
            nx.stuck = 1;
        }
        nx.prev = current; nx.cur = next; nx.t = next;                       // :298-299, :301
        nx.f_cur = f_new;
        nx.sel = (st.sel + 1) % 3;
        nx.halvings = 0;
    } else if (status == 0) {
        nx.t = rnd(st.t * 0.5);                                              // :152
        nx.halvings = st.halvings + 1;
        if (r.max_halvings > 0 && nx.halvings >= r.max_halvings) nx.stuck = 1;
    } else {
        nx.stuck = 1;
    }
    return nx;
}

// The outcome of the pass that ran from `st` into the host's pinned slot: {sum f, t used, (roles used, pass number),
// (status, changed), |dx|^2, |dg|^2} and the seal over the six words and the ticket st.seq + 1 -- the host
// (wait_sealed) accepts the slot only when all seven words belong together, so no ordering between the stores is needed.
__device__ __forceinline__ void adgd_publish(double *__restrict__ slot, const AdgdDev &st, const double (&v)[3], int32_t status, int32_t ch) {
    const double w[6] = {status == 3 ? 0.0 : v[0], st.t, adgd_pack2(st.sel, st.seq), adgd_pack2(status, ch),
                         status == 3 ? 0.0 : v[1], status == 3 ? 0.0 : v[2]};
    unsigned long long x = seal_bits((double)(st.seq + 1));
    // system-scope stores (sc0 sc1: written through to the host's memory now, not when the kernel ends -- the pass this
    // runs in the prologue of lasts 30 us and the host is waiting; a __threadfence_system() would do it too, by writing
    // back every dirty line of the L2 the pass is filling)
    unsigned long long *out = reinterpret_cast<unsigned long long *>(slot);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        __hip_atomic_store(out + i, seal_bits(w[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        x ^= seal_bits(w[i]);
    }
    __hip_atomic_store(out + 6, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// thread 0 of the calling block obtains the three sums of one set of partials (fixed order: what every block and the
// stand-alone decision kernel compute alike)
__device__ __forceinline__ void adgd_sum_partials(const double *__restrict__ set, int grid, double *lds, double (&v)[3]) {
    double a[3] = {0, 0, 0};
    // (every block of a pass runs this in its prologue: the twelve loads of four strides are requested together, the adds
    // keep the plain loop's order)
    for (int i0 = threadIdx.x; i0 < grid; i0 += 4 * kBlock) {
        double t[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * kBlock < grid ? i0 + u * kBlock : i0;
            t[u][0] = set[i];
            t[u][1] = set[(int64_t)kMaxPartialBlocks + i];
            t[u][2] = set[2 * (int64_t)kMaxPartialBlocks + i];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i0 + u * kBlock < grid) { a[0] += t[u][0]; a[1] += t[u][1]; a[2] += t[u][2]; }
        }
    }
    block_sum_multi<3>(a, lds, v);
}

// One kernel for the first trial and for every later trial of the same step (:151-152): x and g still hold
// x_old / g_old, whatever happened before.
template <typename T>
__global__ __launch_bounds__(kBlock) void adgd_fused_rosen_kernel(AdgdFusedParams<T> p) {
    constexpr int N = Vec16<T>::N;
    __shared__ double lds[3 * kWaves];
    __shared__ int lds_flag;
    __shared__ double s_t;
    __shared__ int s_sel, s_stuck;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t nvec = p.n / N;
    const int64_t rows = (nvec + kAdgdOwn - 1) / kAdgdOwn;
    const bool halo_lane = lane == 0 || lane == 63;
    const int q = p.seq & 1;
    double fobj = 0, sdx = 0, sdg = 0;
    bool diff = false;
    double t_state;
    int sel, stuck;
    if (p.mode == 0) {
        // the decision on the pass before this one, by every block for itself (block 0 keeps the books)
        const AdgdDev old = p.st[q ^ 1];                                     // (uniform: scalar loads, in flight with the partials)
        const int32_t ch = p.flags[(p.seq + 2) % 3];
        double v[3];
        adgd_sum_partials(p.partials + (q ^ 1) * kAdgdPartialSet, (int)gridDim.x, lds, v);
        if (threadIdx.x == 0) {
            int32_t status;
            const AdgdDev nx = adgd_next_state(old, v, ch, p.rule, &status);
            s_t = nx.t; s_sel = nx.sel; s_stuck = nx.stuck;
            if (blockIdx.x == 0) {
                p.st[q] = nx;
                adgd_publish(p.slots + 8 * (q ^ 1), old, v, status, ch);
                p.flags[(p.seq + 1) % 3] = 0;                                // (the flag of the NEXT pass: last read by the pass before this one)
            }
        }
        __syncthreads();
        t_state = s_t; sel = s_sel; stuck = s_stuck;
    } else {
        const AdgdDev st = p.st[q];
        t_state = st.t; sel = st.sel; stuck = st.stuck;
        if (blockIdx.x == 0 && threadIdx.x == 0) p.flags[(p.seq + 1) % 3] = 0;
    }
    if (stuck) return;                                                       // (uniform: the search is over, nothing to do)
    const T t = (T)(-t_state);
    const T *__restrict__ xin = sel == 0 ? p.x0 : (sel == 1 ? p.x1 : p.x2);
    T *__restrict__ xout = sel == 0 ? p.x1 : (sel == 1 ? p.x2 : p.x0);
    for (int64_t row = (int64_t)blockIdx.x * kWaves + wave; row < rows; row += (int64_t)gridDim.x * kWaves) {
        const int64_t v = row * kAdgdOwn - 1 + lane;
        const bool valid = v >= 0 && v < nvec;
        const bool owner = valid && !halo_lane;
        const int64_t vc = v < 0 ? 0 : (v >= nvec ? nvec - 1 : v);          // clamped: the value is never used
        const int64_t e0 = v * N;
        T xo[N], go[N];
        load16(xin + vc * N, xo);                                            // halo lanes read their neighbours directly
        {
            const T xop = lane_prev0<T>(xo[N - 1]);
            const T xoq = lane_next0<T>(xo[0]);
#pragma unroll
            for (int j = 0; j < N; ++j)
                go[j] = rosen_grad_elem<T>(e0 + j, p.n, j > 0 ? xo[(j + N - 1) % N] : xop, xo[j], j + 1 < N ? xo[(j + 1) % N] : xoq);
        }
        T xn[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            xn[j] = dfma(t, go[j], xo[j]);                                   // :124
            diff |= owner && !is_equal(xn[j], xo[j]);                      // :128
        }
        const T xprev = lane_prev0<T>(xn[N - 1]);
        const T xnext = lane_next0<T>(xn[0]);
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
            if (p.nt_stores) store16_nt(xout + v * N, xn);
            else store16(xout + v * N, xn);
        }
    }
    block_raise_flag(diff, p.flags + p.seq % 3, &lds_flag);
    const double sums[3] = {fobj, sdx, sdg};
    double out[3];
    block_sum_multi<3>(sums, lds, out);                                      // (bit for bit what three block_sum calls give)
    if (threadIdx.x == 0) {
        double *set = p.partials + q * kAdgdPartialSet;
        set[blockIdx.x] = out[0];
        set[(int64_t)kMaxPartialBlocks + blockIdx.x] = out[1];
        set[2 * (int64_t)kMaxPartialBlocks + blockIdx.x] = out[2];
    }
}

// host state -> device state (a pass that is not the continuation of the passes in flight); its changed-flag starts clear
__global__ void adgd_seed_kernel(AdgdDev *st, AdgdDev v, int32_t *flag) { *st = v; *flag = 0; }

// The decision on pass s as a kernel of its own: for the pass nothing is enqueued behind.  It leaves the state of pass
// s + 1 in place as well (a later pass may start from it in mode 1).
__global__ __launch_bounds__(kBlock) void adgd_decide_kernel(const double *__restrict__ set, int grid, const int32_t *__restrict__ changed,
                                                             const AdgdDev *__restrict__ st, AdgdDev *__restrict__ st_next, AdgdRule rule,
                                                             double *__restrict__ slot) {
    __shared__ double lds[3 * kWaves];
    const AdgdDev old = *st;
    const int32_t ch = *changed;
    double v[3];
    adgd_sum_partials(set, grid, lds, v);
    if (threadIdx.x == 0) {
        int32_t status;
        const AdgdDev nx = adgd_next_state(old, v, ch, rule, &status);
        *st_next = nx;
        adgd_publish(slot, old, v, status, ch);
        __threadfence_system();
    }
}

static int32_t adgd_settle_entry(void *h);

static void adgd_mark_unsettled(dzo_adgd_s *o) {
    // (... or the arrays are the live pair but no look at them is on record: the next dzo_synchronize / dzo_memcpy_* must
    // leave one, because the host may write into current_gradient behind it)
    const bool dirty = o->cur != 0 || !o->g_host_may_write;
    if (dirty && !o->unsettled) { unsettled_add(o, adgd_settle_entry); o->unsettled = true; }
    if (!dirty && o->unsettled) { unsettled_remove(o); o->unsettled = false; }
}

// Drop the pass in flight (if any): it wrote only buffers the current state does not live in, so waiting for it
// is all there is to do; the next pass starts from the host's state again.
static int32_t adgd_cancel_pipeline(dzo_adgd_s *o) {
    if (!o->spec_pending) return DZO_OK;
    DZO_HIP(hipStreamSynchronize(o->core.stream));
    o->spec_pending = false;
    o->spec_discarded += 1;
    return DZO_OK;
}

// gbuf[i] <- gradient of xbuf[i] when the one-pass step left it unwritten (the objective's own gradient kernel: the
// same rosen_grad_elem the pass evaluates in registers)
static int32_t adgd_ensure_g(dzo_adgd_s *o, int i) {
    if (o->g_valid[i] || !o->xbuf[i]) return DZO_OK;
    OptCore &c = o->core;
    DZO_TRY(problem_grad_async(c.problem, c.stream, o->gbuf[i], o->xbuf[i]));
    o->g_valid[i] = true;
    return DZO_OK;
}

// delta_point / delta_gradient as vectors (after steps on the fused pass they exist only as pair cur - pair cur-1)
static int32_t adgd_materialize_deltas(dzo_adgd_s *o) {
    std::lock_guard<std::recursive_mutex> lk(o->mu);
    OptCore &c = o->core;
    if (!o->dx_lazy && !o->dg_lazy) return DZO_OK;
    DZO_TRY(adgd_cancel_pipeline(o));
    const int prev = (o->cur + 2) % 3;
    // fma(-1, old, new) = new - old, rounded once: the value of :145 / :308
    if (o->dx_lazy) DZO_DISPATCH(c.dtype, launch_axpy_oop<T>(c.stream, c.n, (T *)o->dx_buf, (T)-1, (const T *)o->xbuf[prev], (const T *)o->xbuf[o->cur]));
    if (o->dg_lazy) {
        DZO_TRY(adgd_ensure_g(o, prev)); DZO_TRY(adgd_ensure_g(o, o->cur));
        DZO_DISPATCH(c.dtype, launch_axpy_oop<T>(c.stream, c.n, (T *)o->dg_buf, (T)-1, (const T *)o->gbuf[prev], (const T *)o->gbuf[o->cur]));
    }
    DZO_HIP(hipGetLastError());
    o->dx_lazy = o->dg_lazy = false;
    return DZO_OK;
}

// current_point / current_gradient back into the arrays the optimizer aliases
static int32_t adgd_settle(dzo_adgd_s *o, bool hand_out = true) {
    std::lock_guard<std::recursive_mutex> lk(o->mu);
    OptCore &c = o->core;
    DZO_TRY(adgd_cancel_pipeline(o));
    if (o->cur == 0 && o->twin) DZO_TRY(adgd_ensure_g(o, 0));            // (the live pair is the caller's: its gradient array must be current)
    if (o->cur != 0) {
        DZO_TRY(adgd_materialize_deltas(o));                                 // (pair 0 may be the old point the deltas are formed from)
        DZO_TRY(adgd_ensure_g(o, o->cur));
        const size_t bytes = (size_t)c.n * dtype_size(c.dtype);
        DZO_HIP(hipMemcpyAsync(o->x_user, c.x, bytes, hipMemcpyDeviceToDevice, c.stream));
        DZO_HIP(hipMemcpyAsync(o->g_user, c.g, bytes, hipMemcpyDeviceToDevice, c.stream));
        o->cur = 0; c.x = o->x_user; c.g = o->g_user;
        o->g_valid[0] = true;
    }
    if (hand_out) o->g_host_may_write = true;                                // (whoever is handed the arrays may write into them; dzo_adgd_read hands out nothing)
    adgd_mark_unsettled(o);
    return DZO_OK;
}

// is the caller's gradient array still the gradient of the caller's point?  (only asked with the live pair in the
// caller's arrays, i.e. right after a settle or the constructor; pair 1's gradient array is free then and serves as
// scratch.)  One gradient kernel, one compare, one host round trip -- paid by the step that follows a look at the arrays.
static int32_t adgd_gradient_array_is_current(dzo_adgd_s *o, bool *current) {
    OptCore &c = o->core;
    *current = true;
    if (o->cur != 0 || !o->twin) return DZO_OK;
    DZO_TRY(problem_grad_async(c.problem, c.stream, o->gbuf[1], o->xbuf[0]));
    o->g_valid[1] = false;
    int32_t *flag = reinterpret_cast<int32_t *>(c.partials());               // (idle between two steps)
    DZO_HIP(hipMemsetAsync(flag, 0, sizeof(int32_t), c.stream));
    DZO_DISPATCH(c.dtype, launch_isequal<T>(c.stream, c.n, (const T *)o->gbuf[1], (const T *)o->gbuf[0], flag));
    DZO_HIP(hipGetLastError());
    int32_t differs = 0;
    DZO_HIP(hipMemcpyAsync(&differs, flag, sizeof(int32_t), hipMemcpyDeviceToHost, c.stream));
    DZO_HIP(hipStreamSynchronize(c.stream));
    *current = differs == 0;
    return DZO_OK;
}

static int32_t adgd_settle_entry(void *h) {
    dzo_adgd_s *o = static_cast<dzo_adgd_s *>(h);
    DeviceScope scope(o->device);
    DZO_TRY(adgd_settle(o));
    DZO_HIP(hipStreamSynchronize(o->core.stream));
    return DZO_OK;
}

// Eligibility of the one-pass step, INCLUDING its buffers (two more point / gradient pairs, the device
// state and the pinned outcome slots): they are allocated here, before step! changes anything, and a failed
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
        if (hipMalloc(&o->twin, 4 * slot + 16 * 1024) != hipSuccess) { (void)hipGetLastError(); o->twin = nullptr; o->fused = false; return false; }
        char *b = (char *)o->twin + 5 * 1024;
        o->xbuf[0] = o->x_user; o->gbuf[0] = o->g_user;
        o->xbuf[1] = b; o->gbuf[1] = b + slot; o->xbuf[2] = b + 2 * slot; o->gbuf[2] = b + 3 * slot;
    }
    if (!o->dev) {
        const size_t state_bytes = 2 * sizeof(AdgdDev) + 4 * sizeof(int32_t);
        bool ok = hipMalloc((void **)&o->dev, state_bytes) == hipSuccess;
        ok = ok && hipMemsetAsync(o->dev, 0, state_bytes, c.stream) == hipSuccess;
        ok = ok && hipMalloc((void **)&o->pass_partials, sizeof(double) * 2 * kAdgdPartialSet) == hipSuccess;
        ok = ok && hipHostMalloc((void **)&o->slots, sizeof(double) * 16, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess;
        ok = ok && hipHostGetDevicePointer((void **)&o->slots_dev, o->slots, 0) == hipSuccess;
        if (!ok) { (void)hipGetLastError(); o->fused = false; return false; }
        memset(o->slots, 0, sizeof(double) * 16);
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
    fp.st = o->dev;
    fp.x0 = (T *)o->xbuf[0]; fp.g0 = (T *)o->gbuf[0]; fp.x1 = (T *)o->xbuf[1]; fp.g1 = (T *)o->gbuf[1];
    fp.x2 = (T *)o->xbuf[2]; fp.g2 = (T *)o->gbuf[2];
    fp.partials = o->pass_partials;
    int32_t *flags = reinterpret_cast<int32_t *>(o->dev + 2);
    fp.flags = flags;
    fp.slots = o->slots_dev;
    fp.rule.to_f32 = c.dtype == DZO_F32 ? 1 : 0;
    fp.rule.inv_sqrt_two = c.dtype == DZO_F32 ? (double)sqrtf(0.5f) : sqrt(0.5);
    fp.rule.max_halvings = (int64_t)c.max_halvings;
    fp.nt_stores = o->nt_stores ? 1 : 0;
    // grid: four blocks per CU.  (The pass holds 46 registers, so eight would be resident, and a wave has only one row in
    // flight at a time -- but more blocks do not make the pass faster, 33.2-33.7 us by HIP events at 4 / 6 / 8 per CU, and
    // the decision kernel behind it sums three partials per block: 22.0 / 21.4 / 20.4 k step!()/s at n = 1e7, round 3.)
    int64_t blocks = (rows + kWaves - 1) / kWaves;
    const int64_t cap = (int64_t)ctx().cus * (o->pass_bpc < 1 ? 1 : (o->pass_bpc > 8 ? 8 : o->pass_bpc));
    if (blocks > cap) blocks = cap;
    if (blocks > kMaxPartialBlocks) blocks = kMaxPartialBlocks;
    const int grid = (int)(blocks < 1 ? 1 : blocks);
    auto roles = [&]() { return o->cur; };                                   // the pair a pass from the CURRENT state reads
    // pass number o->seq.  from_state: its state is in dev[seq & 1] already (the seed, or the decision kernel behind
    // the pass before it); otherwise the pass decides on its predecessor in its prologue.
    auto enqueue_pass = [&](bool from_state, bool retry) -> int32_t {
        fp.seq = o->seq;
        fp.mode = from_state ? 1 : 0;
        {
            DZO_TIMED(retry ? "adgd_fused_retry" : "adgd_fused_step", s);
            hipLaunchKernelGGL(adgd_fused_rosen_kernel<T>, dim3(grid), dim3(kBlock), 0, s, fp);
        }
        DZO_HIP(hipGetLastError());
        o->seq += 1;
        return DZO_OK;
    };
    auto enqueue_decision = [&](int32_t q) -> int32_t {                      // the decision on pass q as a kernel: outcome into slot q & 1
        hipLaunchKernelGGL(adgd_decide_kernel, dim3(1), dim3(kBlock), 0, s, (const double *)(o->pass_partials + (q & 1) * kAdgdPartialSet), grid,
                           (const int32_t *)(flags + q % 3), (const AdgdDev *)(o->dev + (q & 1)), o->dev + ((q + 1) & 1), fp.rule,
                           o->slots_dev + 8 * (q & 1));
        DZO_HIP(hipGetLastError());
        return DZO_OK;
    };
    auto give_up = [&]() -> int32_t {                                        // :118 delta_point holds x_old when the search gives up
        DZO_TRY(adgd_cancel_pipeline(o));                                    // (delta_gradient stays the previous step's: :128-130 returns
        DZO_HIP(hipMemcpyAsync(o->dx_buf, c.x, (size_t)c.n * sizeof(T), hipMemcpyDeviceToDevice, s));   // before anything else is written)
        o->dx_lazy = false;
        return DZO_OK;
    };
    *done = true;
    c.last_trials = 0;
    int64_t halvings = 0;
    double t = step;
    for (bool first = true;;) {                                              // :121
        // a pass for this trial: the one already in flight, or a fresh one from the host's state
        const bool in_flight = o->spec_pending;
        if (in_flight) {
            o->spec_pending = false;
        } else {
            AdgdDev v;
            v.t = t; v.f_cur = c.f; v.prev = o->previous_step_size; v.cur = o->current_step_size;
            v.sel = roles();
            v.halvings = (int32_t)halvings; v.stuck = 0; v.seq = o->seq;
            hipLaunchKernelGGL(adgd_seed_kernel, dim3(1), dim3(1), 0, s, o->dev + (o->seq & 1), v, flags + o->seq % 3);
            DZO_TRY(enqueue_pass(true, !first));
            if (!o->prologue) DZO_TRY(enqueue_decision(o->seq - 1));
        }
        const int32_t my_seq = o->seq - 1;
        if (o->pipeline) {                                                   // the pass after this one, whatever this one's outcome:
            DZO_TRY(enqueue_pass(!o->prologue, false));                      // its prologue decides on pass my_seq
            if (!o->prologue) DZO_TRY(enqueue_decision(o->seq - 1));
            o->spec_pending = true;
        } else if (o->prologue) {
            DZO_TRY(enqueue_decision(my_seq));
        }
        const double *out = o->slots + 8 * (my_seq & 1);
        DZO_TRY(wait_sealed(s, out, 6, out + 6, (double)(my_seq + 1)));
        int32_t out_sel, out_seq, status, out_changed;
        {
            long long w2, w3;
            memcpy(&w2, out + 2, sizeof(w2)); memcpy(&w3, out + 3, sizeof(w3));
            out_sel = (int32_t)(uint32_t)(w2 & 0xffffffffll); out_seq = (int32_t)(uint32_t)((unsigned long long)w2 >> 32);
            status = (int32_t)(uint32_t)(w3 & 0xffffffffll); out_changed = (int32_t)(uint32_t)((unsigned long long)w3 >> 32);
        }
        (void)out_changed;
        if (in_flight) {
            if (status == 3) {                                                   // the device had closed the search: nothing ran
                DZO_TRY(adgd_cancel_pipeline(o));
                continue;                                                    // the same trial again, from the host's state
            }
            // The pass continued the device's own chain: it must be decision number my_seq with the buffer roles
            // the host derives from its pointers (anything else is a bug, and the pass after it may already have
            // written over the current state: stop).
            DZO_REQUIRE(out_seq == my_seq && out_sel == roles(), DZO_ERR_STATE,
                        "AdGD pipeline out of step with the host (decision %d / %d, buffer pair %d / %d)", (int)out_seq, (int)my_seq, (int)out_sel, roles());
            // Its step size comes from the device-side evaluation of :285-299 / :152 -- the same IEEE expressions
            // the host evaluates, so the two agree bit for bit; if they ever do not, the pass that ran is still a
            // valid step! with the device's value, which then is the optimizer's step size.
            if (out[1] != t) {
                o->spec_corrected += 1;
                t = out[1];
                if (first) o->current_step_size = t;
            }
            o->spec_adopted += 1;
        }
        if (status == 2) {                                                   // :128-130 (x_new == x_old bit for bit)
            DZO_TRY(give_up());
            c.is_stuck = true;
            return DZO_OK;
        }
        DZO_REQUIRE(status == 0 || status == 1, DZO_ERR_STATE, "AdGD pass reported status %d", status);
        c.last_trials += 1;
        if (status == 1) {                                                   // :139
            const double f_new = round_to_dtype(c.dtype, out[0]);
            c.df = round_to_dtype(c.dtype, f_new - c.f);                     // :142-143
            c.f = f_new;                                                     // :144
            o->norm2[0] = out[4]; o->norm2[1] = out[5];
            o->norms_ready = true;
            o->cur = (o->cur + 1) % 3;                                       // the trial point is the current one now; its gradient exists
            c.x = o->xbuf[o->cur]; c.g = o->gbuf[o->cur];                    // in no array yet (adgd_ensure_g)
            o->g_valid[o->cur] = false;
            o->dx_lazy = o->dg_lazy = true;                                  // = pair cur - pair cur-1 (:145, :308), formed on demand
            adgd_mark_unsettled(o);
            o->fused_steps += 1;
            if (!first) o->fused_rejections += 1;
            return DZO_OK;
        }
        t = round_to_dtype(c.dtype, t * 0.5);                                // :152
        first = false;
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
    bool fused_ok = adgd_fused_ok(o);                                // (may allocate; before any state changes)
    if (fused_ok && o->g_host_may_write) {                           // :301 walks along the caller's array, whatever it holds
        bool current = true;
        DZO_TRY(adgd_cancel_pipeline(o));
        DZO_TRY(adgd_gradient_array_is_current(o, &current));
        if (!current) { fused_ok = false; o->host_gradient_steps += 1; }
    }
    o->g_host_may_write = false;
    adgd_mark_unsettled(o);
    if (!fused_ok) DZO_TRY(adgd_cancel_pipeline(o));
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
    DZO_TRY(adgd_materialize_deltas(o));                             // (the generic kernels work on the vectors)
    DZO_TRY(adgd_ensure_g(o, o->cur));                               // (... and on a gradient ARRAY)
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
    DZO_TRY(require_same_backend("AdGDOptimizer", "src/DZOptimization.jl:216-217", x_dev, "initial_point", g_dev, "initial_gradient"));
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
        (void)hipGetLastError(); return DZO_ERR_NOMEM;
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
    o->pipeline = getenv("DZO_TUNE_ADGD_PIPELINE") ? atoi(getenv("DZO_TUNE_ADGD_PIPELINE")) != 0 : true;
    o->nt_stores = getenv("DZO_TUNE_ADGD_NT_STORES") ? atoi(getenv("DZO_TUNE_ADGD_NT_STORES")) != 0 : true;
    o->prologue = getenv("DZO_TUNE_ADGD_PROLOGUE") ? atoi(getenv("DZO_TUNE_ADGD_PROLOGUE")) != 0 : true;
    o->pass_bpc = getenv("DZO_TUNE_ADGD_BPC") ? atoi(getenv("DZO_TUNE_ADGD_BPC")) : 4;
    *out = o;
    return DZO_OK;
}

int32_t dzo_adgd_create_problem(dzo_problem_t problem, void *x_dev, double initial_step_length, dzo_adgd_t *out) {
    DZO_TRY(require_init());
    DZO_REQUIRE(problem && x_dev && out, DZO_ERR_INVALID, "null argument");
    DZO_TRY(require_same_backend("AdGDOptimizer", "src/DZOptimization.jl:254,264", x_dev, "initial_point", nullptr, ""));
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
    DeviceScope scope(o->device);
    if (o->core.stream && o->x_user) (void)adgd_settle(o);           // the caller's arrays end up holding the final point / gradient
    unsettled_retire(o);                                             // (a dzo_synchronize on another thread may be settling this handle right now)
    if (o->core.stream) (void)hipStreamSynchronize(o->core.stream);
    if (o->dx_buf) (void)hipFree(o->dx_buf);
    if (o->dg_buf) (void)hipFree(o->dg_buf);
    if (o->twin) (void)hipFree(o->twin);
    if (o->dev) (void)hipFree(o->dev);
    if (o->pass_partials) (void)hipFree(o->pass_partials);
    if (o->slots) (void)hipHostFree(o->slots);
    core_free(o->core);
    delete o;
    return DZO_OK;
}

int32_t dzo_adgd_set_callbacks(dzo_adgd_t o, dzo_constraint_fn constraint, dzo_objective_fn objective,
                               dzo_gradient_fn gradient, void *cb_ctx) {
    DZO_REQUIRE(o, DZO_ERR_INVALID, "null optimizer");
    DeviceScope scope(o->device);
    {
        std::lock_guard<std::recursive_mutex> lk(o->mu);
        DZO_TRY(adgd_cancel_pipeline(o));
    }
    o->core.constraint = constraint; o->core.objective = objective; o->core.gradient = gradient;
    o->core.cb_ctx = cb_ctx;
    return DZO_OK;
}

int32_t dzo_adgd_step(dzo_adgd_t o) {
    DZO_REQUIRE(o, DZO_ERR_INVALID, "null optimizer");
    DeviceScope scope(o->device);
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
    case 5: *value = o->spec_adopted; break;
    case 6: *value = o->spec_discarded; break;
    case 7: *value = o->spec_corrected; break;
    case 8: *value = o->host_gradient_steps; break;
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

static int32_t adgd_field_ptr(dzo_adgd_t o, int32_t what, bool hand_out, void **ptr_dev) {
    if (what == 1 || what == 3) DZO_TRY(adgd_materialize_deltas(o)); // after one-pass steps the two vectors are formed here
    DZO_TRY(adgd_settle(o, hand_out));                               // current_point / current_gradient ARE the caller's arrays again
    DZO_HIP(hipStreamSynchronize(o->core.stream));
    // the caller may write through the pointer: do not trust what the fused step cached about it
    if (hand_out && (what == 1 || what == 3)) o->norms_ready = false;
    switch (what) {
    case 0: *ptr_dev = o->core.x; break;
    case 1: *ptr_dev = o->core.dx; break;
    case 2: *ptr_dev = o->core.g; break;
    case 3: *ptr_dev = o->core.dg; break;
    default: set_error("dzo_adgd_get_ptr: unknown field %d", what); return DZO_ERR_INVALID;
    }
    return DZO_OK;
}

int32_t dzo_adgd_get_ptr(dzo_adgd_t o, int32_t what, void **ptr_dev) {
    DZO_REQUIRE(o && ptr_dev, DZO_ERR_INVALID, "null argument");
    DeviceScope scope(o->device);
    return adgd_field_ptr(o, what, true, ptr_dev);
}

int32_t dzo_adgd_read(dzo_adgd_t o, int32_t what, void *host_dst) {
    DZO_REQUIRE(o && host_dst, DZO_ERR_INVALID, "null argument");
    DeviceScope scope(o->device);
    std::lock_guard<std::recursive_mutex> lk(o->mu);
    void *src = nullptr;
    DZO_TRY(adgd_field_ptr(o, what, false, &src));
    DZO_HIP(hipMemcpy(host_dst, src, (size_t)o->core.n * dtype_size(o->core.dtype), hipMemcpyDeviceToHost));
    return DZO_OK;
}

}  // extern "C"
