// dzo_lbfgs.hip -- LBFGSOptimizer + step! (src/DZOptimization.jl:321-509) for gfx950.
//
// HBM layout
//   S, Y      two slabs of (m+1) slots x `stride` elements; slot stride is n rounded up to
//             64 elements so every slot starts 256/512-byte aligned for 16-B-per-lane loads.
//             The reference's Vector{A} newest-first order (:339-340, pushfirst! :483,:491) is
//             a ring: logical pair i lives in slot (newest - i) mod (m+1).  The extra slot is
//             the "spare": delta_point / delta_gradient ARE the spare slots of S / Y, so the
//             history push (:482-496) is an index rotation, not two copies.
//   rho       fp64 per slot, device-resident (s.y itself, not its inverse, :505).
//   Gyy, Gsy  (m+1)^2 fp64 Gram caches indexed by slot (GRAM mode).
//
// compute_lbfgs_step_direction! (:430-451) has two device implementations:
//   CHAIN  2k+1 launches, each a fused "axpy_i + dot_{i+1}" pass in the reference's exact
//          operation order; coefficients stay on the device (no host sync inside).  Moves
//          (8k+2)*n elements.
//   GRAM   one pass that streams every s_i, y_i once and accumulates s_i.g, y_i.g and the
//          new row/column of Y'Y and S'Y; an O(k^2) single-wave recurrence that reproduces
//          alpha_i and beta_i from those dot products; one combine pass that applies the
//          2k+1 coefficients per element IN THE REFERENCE'S ELEMENTWISE ORDER (fma chain,
//          then the rmul! scale, then the second fma chain).  Moves (4k+3)*n elements --
//          the algorithmic floor of SURVEY.md 8(d) plus one re-read of g.
#include <cmath>
#include <cstdlib>
#include <limits>
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

#include "dzo_optcore.h"
#include "dzo_rosen.h"

namespace dzo {

constexpr int kGramValues = 5;  // per pair: s.g, y.g, y.y_p, y.s_p, s.y_p  (p = pivot pair)

struct SlotMap {
    uint8_t slot[kMaxHistory];
};

struct RingDecor {
    double l2 = 0; bool bg_on = false; double bg_lo = 0, bg_hi = 0; bool cons_on = false; double cons_lo = 0, cons_hi = 0;
    bool any() const { return l2 != 0.0 || bg_on || cons_on; }
    bool operator==(const RingDecor &o) const {
        return l2 == o.l2 && bg_on == o.bg_on && cons_on == o.cons_on && (!bg_on || (bg_lo == o.bg_lo && bg_hi == o.bg_hi)) &&
               (!cons_on || (cons_lo == o.cons_lo && cons_hi == o.cons_hi));
    }
};

constexpr int kRowOwn = 62;                     // wave-row geometry of the single-pass kernel (see there)
constexpr int kRowLead = (64 - kRowOwn) / 2;
constexpr int kTileBytes = 64 * 16;             // one stream's share of a wave-row in the blocked ring

}  // namespace dzo

struct dzo_lbfgs_s {
    dzo::OptCore core;
    int32_t m = 0;                  // :338 history_length
    int32_t k = 0;                  // length(delta_point_history)
    int32_t newest = 0;             // slot of pair 0
    int64_t stride = 0;             // elements between slots
    void *S = nullptr, *Y = nullptr;
    bool interleaved = true;
    void *d = nullptr;              // :337 step_direction
    int32_t mode = DZO_TWOLOOP_GRAM;
    int32_t n_alpha = 0;            // length(alpha_history) (:498-500)
    int32_t gram_stale = 0;         // pushes since the last Gram pass (>1 => rebuild)
    bool gram_rebuild = false;      // history installed from outside
    // device fp64 scalars
    double *rho = nullptr;          // [m+1] by slot
    double *alpha = nullptr;        // [kMaxHistory] logical order (newest first)
    double *coef = nullptr;         // [kMaxHistory] alpha_i + beta_i
    double *scale = nullptr;        // [1] -rho_1 / (y_1.y_1)   (:444)
    double *alpha_sp = nullptr, *coef_sp = nullptr, *scale_sp = nullptr;   // the NEXT step's scalars, computed speculatively
    bool spec_scalars = false;      // ... and valid (swapped in when that step starts)
    double *Gyy = nullptr, *Gsy = nullptr;  // [(m+1)^2]
    double *sg = nullptr, *yg = nullptr;    // [kMaxHistory]
    double *gram_partials = nullptr;        // [kGramValues*kMaxHistory][gram_grid]
    double *link_partials = nullptr;        // [4][kMaxPartialBlocks] ping-pong + yy
    int gram_grid = 0;
    bool speculate = true;          // enqueue the accepted-step tail before the host sees the decision
    unsigned int *gram_ticket = nullptr;   // (base of the scalar block)
    int gram_variant = 1;           // 1 = lane-distributed accumulators
    int gram_peel = 1;              // predicate-free path for full tiles
    int gram_fresh_plain = 1, gram_skip0 = 1, combine_fresh_plain = 1;
    bool rho_pending = false;       // rho partials of the newest pair await their final sum
    int tail_grid = 0;              // grid of the last speculative tail (its partial count)
    int rho_pending_count = 0, rho_pending_slot = 0;
    // optional safeguards, off by default (= the live reference); SURVEY.md 8(f) rows 2 and 4
    bool descent_check = false;     // legacy/DZOptimization.jl:682-692
    bool sd_fallback = false;       // legacy/DZOptimization.jl:588-610 (steepest descent + history reset)
    int32_t line_search = 0;        // 0 take_backtracking_step!, 1 strong Wolfe on the evaluator quotients
    double wolfe_c1 = 1e-4, wolfe_c2 = 0.9;
    int32_t wolfe_max_evals = 40;
    double last_step_length = 0;    // legacy :625-627
    int64_t history_resets = 0, descent_resets = 0;
    int32_t last_step_kind = 0;     // 0 quasi-Newton, 1 descent-check replacement, 2 fallback
    bool reset_on_push = false;
    void *xt = nullptr, *gt = nullptr;   // LineSearchEvaluator trial_point / trial_gradient (:26-27)
    // single-pass step (lbfgs_single_pass_kernel)
    bool single_pass = true;        // DZO_TUNE_SINGLE_PASS
    bool gram_ready = false;        // gram_partials hold the dots of the CURRENT history (from the last single pass)
    int gram_ready_grid = 0;
    bool scalars_ready = false;     // alpha / coef / scale are valid for the current history and gradient
    // The pass never writes x or g: the trial point and its gradient go to TWIN buffers and the two
    // pairs of pointers swap when the trial is accepted (no backups of x_old / g_old, nothing to restore
    // after a rejected trial).  x_user / g_user are the arrays the optimizer aliases (:393, :395);
    // whenever the host looks (get_ptr, dzo_synchronize, dzo_memcpy_*, destroy) the current point and
    // gradient are settled back into them.
    void *twin_slab = nullptr, *x_twin = nullptr, *g_twin = nullptr, *d_alloc = nullptr;
    void *x_user = nullptr, *g_user = nullptr;
    bool unsettled = false;         // registered in the list that dzo_synchronize / dzo_memcpy_* settle
    std::recursive_mutex mu;        // a step vs a settle coming from another host thread (recursive: a callback may call a getter)
    int64_t single_pass_steps = 0, single_pass_rejections = 0, single_pass_retries = 0;
    bool fused_post = true;         // use the problem's fused accept+gradient+delta kernel when it has one
    bool combine_nts = true;        // non-temporal stores for d in the combine pass
    int gram_u = 4, combine_u = 4, combine_blocks_per_cu = 0, gram_bpc = 0;   // tuning knobs (DZO_TUNE_* env, dev only)

    int device = 0;                 // the GPU this optimizer lives on
    int32_t nslots = 0;             // ring slots: m + 1 (pairs + the spare); m + 2 for a blocked ring (it may hold m + 1 POINTS + the spare)
    int slot_of(int i) const { return ((newest - i) % nslots + nslots) % nslots; }
    int spare() const { return (newest + 1) % nslots; }
    // layout 1 (default): ONE slab, slots interleaved s_0 y_0 s_1 y_1 ... (Y = S + stride, pair
    // stride 2*stride): consecutive streams sit an odd number of KiB apart.  layout 0: two slabs.
    int64_t pair_stride = 0;        // elements between consecutive slots of the same history
    // layout 2 (blocked, see "blocked history ring"): S is the ring, a slot's s / y stream starts at tile
    // 2*slot / 2*slot + 1 of row 0, consecutive rows are rowbytes apart.  delta_point / delta_gradient are
    // then contiguous vectors of their own (dx_lin / dg_lin): the two-pass step works on them and scatters
    // them into the spare slot at the push; after a single-pass step (which writes the tiles directly) they
    // are gathered back only when somebody asks.
    bool blocked = false;
    int64_t rowbytes = 0, ring_rows = 0;
    // Two arrangements of the tiles: TILE-major (tile_stride = 1 KiB, rowbytes = 2 nslots KiB: a wave-row's tiles of
    // all streams adjacent) and STREAM-major (tile_stride = one whole stream, rowbytes = 1 KiB: every stream
    // contiguous).  Reads do not care (tools/pointbench.hip: 495 vs 503 us for the 42 tile reads per row); writes do:
    // a stream that is written lands in consecutive DRAM pages only when it is contiguous (two written tiles per
    // row: 650 us tile-major, 585 us stream-major; in bursts of 16 rows 609 vs 550-560 us).  Stream-major is used
    // whenever the ring's byte offsets fit 32 bits (config 3: 3.63 GB).
    int64_t tile_stride = dzo::kTileBytes;
    size_t ring_bytes = 0;
    void *dx_lin = nullptr, *dg_lin = nullptr;
    bool lin_stale = false;         // dx_lin / dg_lin do not hold the newest pair (a single-pass step pushed it)
    void *export_slab = nullptr;    // contiguous copies of S[i] / Y[i] handed out by get_ptr (2 m vectors, lazily)
    // Point ring (see lbfgs_point_pass_kernel): the blocked ring holds the last k + 1 points / gradients, slot_of(j)
    // = point j (0 = current), pair i = point i - point i+1.  Every trial of every step is one pass; the caller's
    // arrays (core.x / core.g stay x_user / g_user) are gathered from point 0 when the host looks.  Anything else
    // (installed pairs, options, the split entry points, CHAIN mode) first turns the ring into the pair ring in
    // place (lbfgs_leave_points) and continues on the kernels above.
    bool points = false;
    bool xg_lin_stale = false;      // point 0 is newer than the contiguous x_user / g_user
    // The passes recompute the gradients of the ring's points from the point tiles and write only the new POINT; the
    // gradient tiles of a slot are formed on demand (lbfgs_ensure_g) when the host asks for current_gradient /
    // delta_gradient / a Y[i], or the ring is turned into the pair ring.  Bit j: slot j's gradient tiles are valid.
    uint32_t g_valid = 0;
    // The decorators (legacy :219-296) the points of the ring were stored under: the passes recompute every point's
    // gradient, so a pair y_i = g_i - g_i+1 is only the reference's stored pair while the decorators are the ones that
    // were in force when those gradients were first formed.  A change on the user's problem handle (this build lets
    // them be changed between steps) therefore turns the ring into the pair ring under the OLD set (lbfgs_step).
    // General path: the accepted step's tail (delta_point, delta_gradient, rho) deferred into the next step's Gram pass
    // (gram_pass_lanes_kernel<POST>): until then the newest pair's s slot holds x_old, post_g_old the gradient buffer that
    // was current, and whoever needs the pair or the caller's arrays first calls lbfgs_flush_post
    bool post_pending = false;
    void *post_g_old = nullptr;
    dzo::RingDecor ring_dec;
    int ring_obj = 0;               // the objective the passes recompute (ChainObj: 0 Rosenbrock, 1 chained quadratic; 2: log-sum-exp, its own kernels) ...
    double ring_obj_lambda = 0;     // ... and its parameter
    double *pscal = nullptr;        // log-sum-exp: [nslots][2] max / sum exp of the point in each slot (device)
    const void *lse_c = nullptr;    // ... the caller's centre vector (its tiles live in ring slot `nslots`)
    // The caller's arrays ARE current_point / current_gradient (:393): what the host writes into them between two
    // steps must be what the next step starts from.  On the point ring they are copies of point 0, so whenever the
    // host may have looked (a gather into them; construction) the next step first compares them with point 0 and,
    // if somebody changed them, continues on the pair ring with the caller's values (lbfgs_adopt_host_writes).
    bool xg_host_may_write = false;
    int64_t host_write_checks = 0;   // steps that compared the caller's arrays with the ring first (lbfgs_adopt_host_writes)
    int32_t *xg_differs = nullptr;  // device flag of that comparison
    // step_direction is not written by the passes (nothing on the point ring reads it: every trial recomputes it in
    // registers): it is formed when somebody asks, by one more pass over the view of the ring the step started from
    const void *stage_kern = nullptr; size_t stage_bytes = 0; bool stage_small = false;   // dynamic-LDS attribute of the point pass
    bool lazy_d = true;             // DZO_TUNE_LAZY_D
    bool fused_finish = false;      // DZO_TUNE_FUSED_FINISH: reduce + finish of a Gram pass in one launch when the partials are few (measured slower)
    int64_t fused_finish_max = 65536;   // DZO_TUNE_FUSED_FINISH_MAX: ... at most this many partial sums
    int point_sets = 1;             // DZO_TUNE_POINT_SETS: register sets per wave of the point pass (1: two waves per SIMD; the default where it fits)
    bool d_stale = false;
    int dview_k = 0, dview_newest = 0;
    template <typename T> T *s_slot(int slot) const {
        return blocked ? (T *)((char *)S + (size_t)(2 * slot) * (size_t)tile_stride) : (T *)S + (int64_t)slot * pair_stride;
    }
    template <typename T> T *y_slot(int slot) const {
        return blocked ? (T *)((char *)S + (size_t)(2 * slot + 1) * (size_t)tile_stride) : (T *)Y + (int64_t)slot * pair_stride;
    }
    void *s_slot_v(int slot) const {
        return blocked ? (void *)((char *)S + (size_t)(2 * slot) * (size_t)tile_stride) : (void *)((char *)S + (size_t)slot * pair_stride * dzo::dtype_size(core.dtype));
    }
    void *y_slot_v(int slot) const {
        return blocked ? (void *)((char *)S + (size_t)(2 * slot + 1) * (size_t)tile_stride) : (void *)((char *)Y + (size_t)slot * pair_stride * dzo::dtype_size(core.dtype));
    }
    void refresh_delta_ptrs() {
        if (blocked) { core.dx = dx_lin; core.dg = dg_lin; }
        else { core.dx = s_slot_v(spare()); core.dg = y_slot_v(spare()); }
    }
};

namespace dzo {

template <typename T, bool VEC> struct Ld {
    static constexpr int N = VEC ? Vec16<T>::N : 1;
    static __device__ __forceinline__ void load(const T *p, T (&v)[N]) {
        if constexpr (VEC) load16(p, v); else v[0] = p[0];
    }
    static __device__ __forceinline__ void load_nt(const T *p, T (&v)[N]) {
        if constexpr (VEC) load16_nt(p, v); else v[0] = __builtin_nontemporal_load(p);
    }
    static __device__ __forceinline__ void store_nt(T *p, const T (&v)[N]) {
        if constexpr (VEC) store16_nt(p, v); else __builtin_nontemporal_store(v[0], p);
    }
    static __device__ __forceinline__ void store(T *p, const T (&v)[N]) {
        if constexpr (VEC) store16(p, v); else p[0] = v[0];
    }
};

// ---------------------------------------------------------------------------- blocked history ring
// When the optimizer is eligible for the single-pass step its (s, y) ring is stored TILE-MAJOR instead of
// as 2(m+1) slabs: the unit is a wave-row of the single-pass kernel -- 62 owned 16-B vectors plus a copy of
// the neighbouring vector on each side = one 1-KiB tile [halo | 62 | halo] per stream -- and the
// 2(m+1) tiles of a row (s_0 y_0 s_1 y_1 ...) are adjacent.  A wave-row's 2k history loads are then one
// contiguous run of aligned full lines (k = 20: 40 KiB) instead of 40 unaligned 992-byte pieces 80 MB
// apart.  Vector v of a stream lives in row v / 62 at position v % 62 + 1; the first / last vector of a row
// is duplicated at position 63 / 0 of the previous / next row.  The two-pass kernels address the same
// layout through hist_ptr; contiguous views (S[i], Y[i], delta_point, delta_gradient, set_history) are
// gathered / scattered on demand.
template <bool BLK, typename T>
__device__ __forceinline__ const T *hist_ptr(const T *base, int64_t vi, int64_t rowbytes) {
    if constexpr (!BLK) return base + vi * Vec16<T>::N;
    const uint32_t v = (uint32_t)vi, r = v / (uint32_t)kRowOwn, q = v - r * (uint32_t)kRowOwn + (uint32_t)kRowLead;
    return reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + (int64_t)r * rowbytes + (int64_t)q * 16);
}

// ============================================================================ CHAIN mode
// One link of the two-loop recursion: out = post * fma(coef, v, in), with
//   first loop  (:440-441)  a = dot/rho,  coef = -a,            alpha[i] = a
//   second loop (:447-448)  b = dot/rho,  coef = -(alpha[i]+b), coef_out = alpha[i]+b
// `dot` is the fixed-order sum of the previous launch's per-block partials; the link also
// accumulates the NEXT link's dot (w . out) and, on request, v.v for the :444 scale.
struct LinkParams {
    int64_t n;
    const void *in;
    void *out;
    const void *v;
    const void *w;              // may be null
    const double *prev;         // partials of the dot feeding this link
    int prev_count;
    const double *rho;          // rho of this pair
    double *alpha;              // &alpha[i]
    double *coef_out;           // &coef[i] (second loop) or null
    int second_loop;
    const double *yy;           // partials of y_1.y_1 (when this link applies the :444 scale)
    int yy_count;
    const double *rho0;
    double *scale_out;
    double *dot_out;            // partials of w.out
    double *yy_out;             // partials of v.v, may be null
};

template <typename T, bool VEC>
__global__ __launch_bounds__(kBlock) void chain_link_kernel(LinkParams p) {
    using L = Ld<T, VEC>;
    constexpr int N = L::N;
    __shared__ double lds[kWaves];
    const double dot = reduce_partials_all(p.prev, p.prev_count, lds);
    const double q = dot / p.rho[0];
    double c;
    if (p.second_loop) {
        c = p.alpha[0] + q;
        if (blockIdx.x == 0 && threadIdx.x == 0) p.coef_out[0] = c;
    } else {
        c = q;
        if (blockIdx.x == 0 && threadIdx.x == 0) p.alpha[0] = q;
    }
    const T coef = (T)(-c);
    const bool has_post = p.yy != nullptr;
    T post = (T)1;
    if (has_post) {
        const double yy = reduce_partials_all(p.yy, p.yy_count, lds);
        const double sc = -p.rho0[0] / yy;                       // :444
        post = (T)sc;
        if (blockIdx.x == 0 && threadIdx.x == 0) p.scale_out[0] = sc;
    }
    const T *in = (const T *)p.in;
    T *out = (T *)p.out;
    const T *v = (const T *)p.v;
    const T *w = (const T *)p.w;
    double acc_w = 0, acc_v = 0;
    const int64_t nvec = p.n / N;
    const int64_t nthreads = (int64_t)gridDim.x * kBlock;
    auto body = [&](int64_t i, auto tag) {
        constexpr int M = decltype(tag)::value;
        T qv[M], vv[M], wv[M];
        if constexpr (M == N && VEC) { load16(in + i, qv); load16(v + i, vv); if (w) load16(w + i, wv); }
        else { qv[0] = in[i]; vv[0] = v[i]; if (w) wv[0] = w[i]; }
#pragma unroll
        for (int j = 0; j < M; ++j) {
            T r = dfma(coef, vv[j], qv[j]);
            if (has_post) r = post * r;
            qv[j] = r;
            if (w) acc_w = __builtin_fma((double)wv[j], (double)r, acc_w);
            if (p.yy_out) acc_v = __builtin_fma((double)vv[j], (double)vv[j], acc_v);
        }
        if constexpr (M == N && VEC) store16(out + i, qv); else out[i] = qv[0];
    };
    for (int64_t base = (int64_t)blockIdx.x * kBlock * 2; base < nvec; base += nthreads * 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t vi = base + (int64_t)u * kBlock + threadIdx.x;
            if (vi < nvec) body(vi * N, std::integral_constant<int, N>{});
        }
    }
    if constexpr (VEC) {
        const int64_t i = nvec * N + (int64_t)blockIdx.x * kBlock + threadIdx.x;
        if (i < p.n) body(i, std::integral_constant<int, 1>{});
    }
    if (p.dot_out) {
        double r = block_sum(acc_w, lds);
        if (threadIdx.x == 0) p.dot_out[blockIdx.x] = r;
    }
    if (p.yy_out) {
        double r = block_sum(acc_v, lds);
        if (threadIdx.x == 0) p.yy_out[blockIdx.x] = r;
    }
}

// first launch of the chain: partials of s_1.g and (k == 1 only) y_1.y_1
template <typename T, bool VEC>
__global__ __launch_bounds__(kBlock) void chain_head_kernel(int64_t n, const T *__restrict__ s, const T *__restrict__ g,
                                                            const T *__restrict__ y, double *__restrict__ dot_out,
                                                            double *__restrict__ yy_out) {
    using L = Ld<T, VEC>;
    constexpr int N = L::N;
    __shared__ double lds[kWaves];
    double a = 0, b = 0;
    const int64_t nvec = n / N;
    const int64_t nthreads = (int64_t)gridDim.x * kBlock;
    for (int64_t vi = (int64_t)blockIdx.x * kBlock + threadIdx.x; vi < nvec; vi += nthreads) {
        T sv[N], gv[N], yv[N];
        L::load(s + vi * N, sv);
        L::load(g + vi * N, gv);
        if (yy_out) L::load(y + vi * N, yv);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            a = __builtin_fma((double)sv[j], (double)gv[j], a);
            if (yy_out) b = __builtin_fma((double)yv[j], (double)yv[j], b);
        }
    }
    if constexpr (VEC) {
        const int64_t i = nvec * N + (int64_t)blockIdx.x * kBlock + threadIdx.x;
        if (i < n) {
            a = __builtin_fma((double)s[i], (double)g[i], a);
            if (yy_out) b = __builtin_fma((double)y[i], (double)y[i], b);
        }
    }
    double r = block_sum(a, lds);
    if (threadIdx.x == 0) dot_out[blockIdx.x] = r;
    if (yy_out) {
        r = block_sum(b, lds);
        if (threadIdx.x == 0) yy_out[blockIdx.x] = r;
    }
}

// ============================================================================ GRAM mode
// dev instrumentation: start / end clock of every wave of the last instrumented kernel (point pass: DZO_TUNE_SP_DEBUG & 1024;
// Gram pass / combine: DZO_TUNE_TP_DEBUG = 1 / 2), read back by tools/wave_times.py
__device__ unsigned long long g_dev_wave_times[1024 * 4 * 2];
__device__ __forceinline__ void dev_stamp(int debug_on, int which) {
    if (debug_on && blockIdx.x < 1024 && (threadIdx.x & 63) == 0) g_dev_wave_times[(blockIdx.x * kWaves + (threadIdx.x >> 6)) * 2 + which] = wall_clock64();
}

template <typename T> struct GramParams {
    int debug;
    int64_t n;
    const T *g;
    const T *sp;                // pivot pair (the pair whose Gram row/column is (re)computed)
    const T *yp;
    int k;
    int peel;                   // 1: full tiles run the predicate-free path
    int fresh_plain;            // 1: g / pivot pair (written by the previous kernel) with plain loads
    int pivot_first;            // 1: pivot == logical pair 0 -> pair 0 is served from the pivot registers
    const T *s[kMaxHistory];    // logical pair -> slot base (wave-uniform index -> scalar loads)
    const T *y[kMaxHistory];
    double *partials;           // [kGramValues * k][gridDim.x]
    int64_t rowbytes;           // blocked ring: bytes between consecutive wave-rows (BLK kernels only)
    // POST kernels: the pivot pair does not exist yet -- it is the accepted step's (delta_point, delta_gradient), formed here
    // from x, x_old (which the first trial saved in the pivot's s slot, :118) and g (new), g_old, and WRITTEN to the pivot's slots
    const T *x, *g_old;
    T *sp_out, *yp_out;
};

__device__ __forceinline__ void wave_sum5(const double (&t)[5], int lane, double (&tot)[5]) {
    const bool A = (lane & 1) != 0, B = (lane & 2) != 0, C = (lane & 8) != 0;
    // lane^1: lanes with A = 0 keep {t0,t1,t2}, lanes with A = 1 keep {t3,t4}
    const double r0 = lane_xor1(A ? t[0] : t[3]);
    const double r1 = lane_xor1(A ? t[1] : t[4]);
    const double r2 = lane_xor1(A ? t[2] : 0.0);
    const double a0 = (A ? t[3] : t[0]) + r0;       // t0 | t3
    const double a1 = (A ? t[4] : t[1]) + r1;       // t1 | t4
    const double a2 = t[2] + r2;                    // t2 | (unused)
    // lane^2: (A,B) = 00 keeps {t0,t1}, 01 keeps {t2}, 10 keeps {t3}, 11 keeps {t4}
    const double u0 = A ? (B ? a0 : a1) : (B ? a0 : a2);
    const double u1 = (!A && B) ? a1 : 0.0;
    const double v0 = lane_xor2(u0);
    const double v1 = lane_xor2(u1);
    const double c0 = (A ? (B ? a1 : a0) : (B ? a2 : a0)) + v0;   // t0 | t2 | t3 | t4
    const double c1 = a1 + v1;                                         // t1 (group 00 only)
    // lane^8: group 00 splits {t0,t1} by C; the other groups hold one value already
    const bool g00 = !A && !B;
    const double x = lane_xor8(g00 ? (C ? c0 : c1) : c0);
    double e = (g00 ? (C ? c1 : c0) : c0) + x;
    e += lane_xor4(e);
    e = sum_xor16(e);
    e = sum_xor32(e);
    tot[0] = readlane_f64(e, 0);
    tot[1] = readlane_f64(e, 8);
    tot[2] = readlane_f64(e, 2);
    tot[3] = readlane_f64(e, 1);
    tot[4] = readlane_f64(e, 3);
}

struct GramFinishParams {
    int k;
    int m1;                     // m + 1 (leading dimension of the slot-indexed Gram caches)
    int pivot;                  // logical index of the pivot pair
    int grid;                   // blocks of the Gram pass
    int do_recurrence;
    SlotMap map;
    const double *partials;     // reduced values [kGramValues * k]
    double *rho;                // by slot
    int rho_from_vals;          // 1: rho[pivot] = s_p.y_p taken from the reduced values (single-pass step)
    int rho_to_f32;
    const int32_t *gate;        // speculative launch: run only if *gate == 1
    double *Gyy, *Gsy;
    double *sg, *yg;
    double *alpha, *coef, *scale;
    int fast_div;               // the recurrence's quotients by fd_div (DZO_TUNE_FAST_DIV, default 1; 0: every one an IEEE division)
};

// Second stage of the Gram pass: one block per value sums that value's per-block partials in
// a fixed order (value-major layout -> contiguous reads).
// The last block (index nvals) optionally finishes the pending rho = delta_point.delta_gradient
// of the pair pushed by the previous step (:505), saving that step a launch of its own.
__device__ __forceinline__ void gram_reduce_body(const double *__restrict__ partials, int grid, double *__restrict__ vals, int nvals,
                                                 const double *__restrict__ rho_partials, int rho_count,
                                                 double *__restrict__ rho_dst, int rho_to_f32, double *lds) {
    if ((int)blockIdx.x == nvals) {
        const double r = reduce_partials_all(rho_partials, rho_count, lds);
        if (threadIdx.x == 0) rho_dst[0] = rho_to_f32 ? (double)(float)r : r;
        return;
    }
    const double *src = partials + (int64_t)blockIdx.x * grid;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int b = threadIdx.x; b < grid; b += 4 * kBlock) {
        a0 += src[b];
        if (b + kBlock < grid) a1 += src[b + kBlock];
        if (b + 2 * kBlock < grid) a2 += src[b + 2 * kBlock];
        if (b + 3 * kBlock < grid) a3 += src[b + 3 * kBlock];
    }
    const double r = block_sum((a0 + a1) + (a2 + a3), lds);
    if (threadIdx.x == 0) vals[blockIdx.x] = r;
}

__global__ __launch_bounds__(kBlock) void gram_reduce_kernel(const double *__restrict__ partials, int grid,
                                                             double *__restrict__ vals, int nvals,
                                                             const double *__restrict__ rho_partials, int rho_count,
                                                             double *__restrict__ rho_dst, int rho_to_f32,
                                                             const int32_t *__restrict__ gate = nullptr) {
    __shared__ double lds[kWaves];
    if (gate && *gate != 1) return;            // speculative launch: only after an accepted trial
    gram_reduce_body(partials, grid, vals, nvals, rho_partials, rho_count, rho_dst, rho_to_f32, lds);
}

// The reduction behind a single-pass step, with the step's :128 / :139 decision as one more block (index nvals):
// one launch instead of two between the pass and gram_finish.  The sums are formed whatever the decision will be
// (they are scratch); gram_finish, the next launch, is the one that is gated on the status this launch publishes.
__global__ __launch_bounds__(kBlock) void gram_reduce_decide_kernel(const double *__restrict__ partials, int grid,
                                                                    double *__restrict__ vals, int nvals, DecideArgs dec) {
    __shared__ double lds[kWaves];
    if ((int)blockIdx.x == nvals) { decide_body(dec, lds); return; }
    gram_reduce_body(partials, grid, vals, nvals, nullptr, 0, nullptr, 0, lds);
}
// (1) refresh of the pivot row/column of the slot-indexed Gram caches from the reduced values,
// a / b without the division's instruction chain, for the dependent quotients of the recurrence below (2k of them per
// two-loop, ~200 cycles each as the compiler expands an IEEE division: more than half of gram_finish_kernel's time).
// y = RN(1 / b) is formed once per lane, off the chain, by a real division.  Then q0 = RN(a y) is within two ulps of
// a / b; r0 = RN(a - q0 b), q1 = RN(q0 + r0 y) is a faithful rounding; r1 = a - q1 b is exact (fma) and
// q2 = RN(q1 + r1 y) is the correctly rounded quotient (Markstein's theorem: y correctly rounded, q1 faithful) -- the
// value `a / b` has, bit for bit, whenever nothing under- or overflows on the way.  fd_mid() keeps the operands in
// the middle of the exponent range (zeros, subnormals, infinities and NaNs excluded with it); anything else takes the
// division.  dzo_selftest_fast_div compares the two on the device over as many operand pairs as the caller likes.
__device__ __forceinline__ bool fd_mid(double v) {
    const int e = (int)((__double_as_longlong(v) >> 52) & 0x7ff);
    return e > 1023 - 500 && e < 1023 + 500;
}
__device__ __forceinline__ double fd_div(double a, double b, double y) {
    const double q0 = a * y;
    const double r0 = __builtin_fma(-q0, b, a);
    const double q1 = __builtin_fma(r0, y, q0);
    const double r1 = __builtin_fma(-q1, b, a);
    return __builtin_fma(r1, y, q1);
}

// (2) the two-loop recursion on SCALARS by one wave, lane i owning pair i:
//     s_i.q_i = s_i.g - sum_{j<i} alpha_j (s_i.y_j)                     (:440)
//     y_i.r_i = scale*(y_i.g - sum_j alpha_j y_i.y_j) - sum_{l>i} c_l (s_l.y_i)   (:447)
// One block.  vals: kGramValues*k reduced values already in LDS; yy / sy: k x (k+1) LDS scratch.
// gate_ok: the speculative launch's gate as the caller read it (the read is requested first and looked at only here,
// behind the other requests: whatever was fetched before that is dropped, nothing has been stored outside LDS)
__device__ __forceinline__ void gram_finish_body(const GramFinishParams &p, double *vals, double *yy, double *sy, int gate_ok = 1) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = p.k, ld = k + 1;
    // (requested first, used last: rho of lane i's pair, for the recurrence below -- the pivot's own entry may be
    // rewritten further down by this very launch, and the lane that owns it then takes the new value from there)
    const bool on = wave == 0 && lane < k;
    double rho_i = on ? p.rho[p.map.slot[lane]] : 1.0;
    // logical k x k views of the caches (entries of non-pivot pairs were computed by the
    // passes in which THEY were the pivot)
    for (int e = threadIdx.x; e < k * k; e += (int)blockDim.x) {
        const int i = e / k, j = e % k;
        yy[i * ld + j] = p.Gyy[p.map.slot[i] * p.m1 + p.map.slot[j]];
        sy[i * ld + j] = p.Gsy[p.map.slot[i] * p.m1 + p.map.slot[j]];
    }
    if (!gate_ok) return;                                          // (uniform: every thread read the same word)
    __syncthreads();
    const int pv = p.pivot, ps = p.map.slot[pv];
    if (p.rho_from_vals) {                                         // :505 for the pair pushed by the single pass
        const double r = vals[pv * kGramValues + 4];
        const double rr = p.rho_to_f32 ? (double)(float)r : r;
        if (threadIdx.x == 0) p.rho[ps] = rr;
        if (on && lane == pv) rho_i = rr;                          // (the value the early load could not see yet)
    }
    if (threadIdx.x < k) {
        const int i = threadIdx.x, si = p.map.slot[i];
        const double *v = vals + i * kGramValues;
        p.sg[i] = v[0];
        p.yg[i] = v[1];
        yy[i * ld + pv] = v[2]; yy[pv * ld + i] = v[2];
        p.Gyy[si * p.m1 + ps] = v[2]; p.Gyy[ps * p.m1 + si] = v[2];
        sy[pv * ld + i] = v[3];                 // s_p . y_i
        p.Gsy[ps * p.m1 + si] = v[3];
        sy[i * ld + pv] = v[4];                 // s_i . y_p
        p.Gsy[si * p.m1 + ps] = v[4];
    }
    __syncthreads();
    if (!p.do_recurrence || wave != 0) return;

    // The three loops are chains of dependent steps (a division, a v_readlane, an fma per pair), and this kernel sits
    // between two passes of every step: each loop fetches the Gram entry of its NEXT iteration before it works on the
    // current one, so the LDS round trip runs under the division instead of in front of it.  Same operations on the same
    // operands in the same order -- the scalars are bit for bit what the plain loops gave.
    const int row = on ? lane : 0;                                 // (lanes beyond k read row 0 and use nothing)
    const double y_i = 1.0 / rho_i;                                // (for fd_div: one real division per lane, off the chains)
    const bool fd_b = p.fast_div != 0 && fd_mid(rho_i) && fd_mid(y_i);
    double acc = on ? vals[lane * kGramValues + 0] : 0.0;          // s_i.g
    double alpha_i = 0;
    double nxt = sy[row * ld + 0];
    for (int j = 0; j < k; ++j) {                                  // :439 newest -> oldest
        const double syj = nxt;
        if (j + 1 < k) nxt = sy[row * ld + j + 1];
        double cand = fd_div(acc, rho_i, y_i);                     // :440 acc / rho_i (lane j's value counts: ...
        if (lane == j && !(fd_b && fd_mid(acc))) cand = acc / rho_i;   // ... the division itself when ITS operands are unusual)
        const double aj = readlane_f64(cand, j);                    // (uniform j: v_readlane, no LDS round trip)
        if (lane == j) alpha_i = cand;
        if (on && lane > j) acc = __builtin_fma(-aj, syj, acc);
    }
    if (on) p.alpha[lane] = alpha_i;
    const double scale = -readlane_f64(rho_i, 0) / yy[0];          // :444 (rho of the newest pair = lane 0's)
    if (lane == 0) p.scale[0] = scale;
    // y_i.q_k with q_k = g - sum_j alpha_j y_j
    double base = on ? vals[lane * kGramValues + 1] : 0.0;         // y_i.g
    nxt = yy[row * ld + 0];
    for (int j = 0; j < k; ++j) {
        const double yyj = nxt;
        if (j + 1 < k) nxt = yy[row * ld + j + 1];
        const double aj = readlane_f64(alpha_i, j);
        if (on) base = __builtin_fma(-aj, yyj, base);
    }
    acc = scale * base;
    double c_i = 0;
    nxt = sy[(k - 1) * ld + row];
    for (int l = k - 1; l >= 0; --l) {                             // :446 oldest -> newest
        const double syl = nxt;
        if (l > 0) nxt = sy[(l - 1) * ld + row];
        double quo = fd_div(acc, rho_i, y_i);                      // :447-448 alpha_i + acc / rho_i (lane l's value counts)
        if (lane == l && !(fd_b && fd_mid(acc))) quo = acc / rho_i;
        const double cand = alpha_i + quo;
        const double cl = readlane_f64(cand, l);
        if (lane == l) c_i = cand;
        if (on && lane < l) acc = __builtin_fma(-cl, syl, acc);
    }
    if (on) p.coef[lane] = c_i;
}

// bytes of dynamic LDS gram_finish_body needs: vals + yy + sy
static inline size_t gram_finish_lds_bytes(int k) { return sizeof(double) * ((size_t)kGramValues * k + 2 * (size_t)k * (k + 1)); }

// (Tried for config 4, n = 1e6, where the scalar stage's launches cost as much as a pass over the data:
// this block summing the raw partials itself -- 25.5 us instead of 6.6 + 8.3 for the two kernels, one
// block cannot pull 50 x 1000 values through one CU fast enough -- and a last-ticket epilogue inside the
// Gram pass, which needs either cache-wide fences, 195 us per pass, or write-through stores and the same
// single-block sum, 87 us.  The three-launch form stays.)
__global__ __launch_bounds__(kBlock) void gram_finish_kernel(GramFinishParams p) {
    const int gate_ok = p.gate ? (*p.gate == 1) : 1;               // (looked at in gram_finish_body, behind the other loads)
    extern __shared__ __attribute__((aligned(16))) double fin_lds[];
    const int k = p.k;
    double *vals = fin_lds, *yy = vals + kGramValues * k, *sy = yy + k * (k + 1);
    for (int v = threadIdx.x; v < kGramValues * k; v += kBlock) vals[v] = p.partials[v];
    gram_finish_body(p, vals, yy, sy, gate_ok);                    // (its first barrier stands behind these stores too)
}

// Reduce + finish in ONE launch of one 1024-thread block, for the sizes where the scalar stage's two launches cost
// as much as a pass over the data (config 4: n = 1e6, k = 10: 6.7 + 8.4 us and a kernel boundary against two passes
// of 15 us).  A WAVE sums a value's partials (16 values at a time), emulating gram_reduce_kernel's 256-thread block
// bit for bit: lane l plays threads l, l + 64, l + 128, l + 192 (each with its four strided accumulators), the four
// wave sums are taken in block_sum's order -- so an optimizer gets the same scalars whichever form runs.  (The
// round-2 attempt let the finish block's 256 threads walk all 50 x 768 partials value by value: 25.5 us.)
// MEASURED (round 3, config 4, rocprofv3): 22.0 us for this kernel against 6.4 + 8.3 us for the two launches, 51.6
// against 38.8-39.9 us of wall per direction -- one CU pulls 307 KB of partials in twelve dependent rounds of loads.
// Off by default (DZO_TUNE_FUSED_FINISH=1 selects it); the same trace shows 2.9 us of GPU-side gaps per direction, i.e.
// the launches were never the cost: round 2's 70.8 us per direction were 8 HIP event records per direction in the timed
// loop (the bench now times without events and takes the per-kernel numbers from a second, untimed stretch).
constexpr int kFusedFinishThreads = 1024;
__global__ __launch_bounds__(kFusedFinishThreads) void gram_reduce_finish_kernel(GramFinishParams p, const double *__restrict__ partials, int grid,
                                                                                 double *__restrict__ vals_out, int nvals,
                                                                                 const double *__restrict__ rho_partials, int rho_count,
                                                                                 double *__restrict__ rho_dst, int rho_to_f32) {
    if (p.gate && *p.gate != 1) return;
    extern __shared__ __attribute__((aligned(16))) double fin_lds[];
    __shared__ double rlds[kWaves];
    const int k = p.k;
    double *vals = fin_lds, *yy = vals + kGramValues * k, *sy = yy + k * (k + 1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int kW = kFusedFinishThreads / 64;
    // pending rho of the pair the previous step pushed (gram_reduce_body's extra block; reduce_partials_all's order)
    if (rho_count > 0) {
        double v = 0;
        if (threadIdx.x < kBlock) {
            for (int i = threadIdx.x; i < rho_count; i += kBlock) v += rho_partials[i];
            v = wave_sum_all(v);
            if (lane == 0) rlds[wave] = v;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const double r = ((rlds[0] + rlds[1]) + rlds[2]) + rlds[3];
            rho_dst[0] = rho_to_f32 ? (double)(float)r : r;
        }
        __threadfence_block();
        __syncthreads();
    }
    for (int v = wave; v < nvals; v += kW) {
        const double *src = partials + (int64_t)v * grid;
        double W[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int t = lane + 64 * q;                            // the thread of gram_reduce_body this lane plays
            double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
            for (int b = t; b < grid; b += 4 * kBlock) {
                a0 += src[b];
                if (b + kBlock < grid) a1 += src[b + kBlock];
                if (b + 2 * kBlock < grid) a2 += src[b + 2 * kBlock];
                if (b + 3 * kBlock < grid) a3 += src[b + 3 * kBlock];
            }
            W[q] = wave_sum((a0 + a1) + (a2 + a3));
        }
        if (lane == 0) {
            const double r = ((W[0] + W[1]) + W[2]) + W[3];          // block_sum: r = 0; r += lds[w]
            vals[v] = r;
            vals_out[v] = r;
        }
    }
    __syncthreads();
    gram_finish_body(p, vals, yy, sy);
}

// Gram pass, lane-distributed accumulators (the default).  All waves of all blocks walk the
// pairs in the same order and read 16-B-per-lane contiguous chunks of ONE stream at a time --
// the access pattern of combine_kernel, which HBM serves ~10 % faster than the pair-per-wave
// variant's 8 concurrent streams per block (measured: equal queue depth, lower DRAM
// efficiency).  The 5 dot products of a (tile, pair) are reduced across the wave by a
// butterfly and added into lane i's accumulators for pair i, so a lane carries 5 fp64
// accumulators whatever k is (k <= 64 = wave width), occupancy stays high, and no operand is
// loaded twice.  VALU/LDS cost of the butterflies: ~15 % of the memory time at U = 4.
// POST (general path, round 4): the accepted step's tail rides in this pass -- delta_point = x - x_old (:145), delta_gradient = g - g_old
// (:478-480) are formed per element, stored to the pivot pair's slots and used as the pivot pair at once (their dots include
// rho = delta_point . delta_gradient, :505); replaces accept_delta_rho_kernel + the re-read of the pair it wrote.
template <typename T, bool VEC, int U, bool BLK = false, bool POST = false>
__global__ __launch_bounds__(kBlock) void gram_pass_lanes_kernel(GramParams<T> p) {
    using L = Ld<T, VEC>;
    constexpr int N = L::N;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    double acc[kGramValues];
#pragma unroll
    for (int c = 0; c < kGramValues; ++c) acc[c] = 0;
    dev_stamp(p.debug, 0);
    const T *sp = p.sp, *yp = p.yp;
    const int k = p.k;
    const int64_t nvec = p.n / N;
    const int64_t nthreads = (int64_t)gridDim.x * kBlock;
    // One tile = kBlock*U vectors of every stream.  FULL tiles run without any bounds predicate:
    // with predicates every load sits behind an exec branch, the compiler can no longer count
    // outstanding loads and waits vmcnt(0) before each fma block, which serialises the
    // two-register-set prefetch below.
    auto do_tile = [&](int64_t base, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        T gv[U][N], spv[U][N], ypv[U][N];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t vi = base + (int64_t)u * kBlock + threadIdx.x;
            ok[u] = FULL || vi < nvec;
            if constexpr (POST) {
                if (ok[u]) {
                    T xv[N], go[N];
                    L::load(p.g + vi * N, gv[u]);
                    L::load(p.x + vi * N, xv);
                    L::load(sp + vi * N, spv[u]);                        // x_old
                    L::load(p.g_old + vi * N, go);
#pragma unroll
                    for (int j = 0; j < N; ++j) { spv[u][j] = xv[j] - spv[u][j]; ypv[u][j] = gv[u][j] - go[j]; }
                    L::store(p.sp_out + vi * N, spv[u]);
                    L::store(p.yp_out + vi * N, ypv[u]);
                } else {
#pragma unroll
                    for (int j = 0; j < N; ++j) { gv[u][j] = 0; spv[u][j] = 0; ypv[u][j] = 0; }
                }
            } else
            if (ok[u]) {
                if (p.fresh_plain) {
                    L::load(p.g + vi * N, gv[u]);
                    L::load(hist_ptr<BLK>(sp, vi, p.rowbytes), spv[u]);
                    L::load(hist_ptr<BLK>(yp, vi, p.rowbytes), ypv[u]);
                } else {
                    L::load_nt(p.g + vi * N, gv[u]);
                    L::load_nt(hist_ptr<BLK>(sp, vi, p.rowbytes), spv[u]);
                    L::load_nt(hist_ptr<BLK>(yp, vi, p.rowbytes), ypv[u]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < N; ++j) { gv[u][j] = 0; spv[u][j] = 0; ypv[u][j] = 0; }
            }
        }
        // two register sets: the loads of pair i+1 are in flight while pair i is multiplied and
        // reduced (the butterfly is a ~0.3 us dependent chain that would otherwise sit between
        // consecutive memory round trips of the wave)
        T sA[U][N], yA[U][N], sB[U][N], yB[U][N];
        auto fetch = [&](int i, T (&sv)[U][N], T (&yv)[U][N]) {
            const T *si = p.s[i];
            const T *yi = p.y[i];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (FULL || ok[u]) {
                    const int64_t vi = base + (int64_t)u * kBlock + threadIdx.x;
                    L::load_nt(hist_ptr<BLK>(si, vi, p.rowbytes), sv[u]);
                    L::load_nt(hist_ptr<BLK>(yi, vi, p.rowbytes), yv[u]);
                } else {
#pragma unroll
                    for (int j = 0; j < N; ++j) { sv[u][j] = 0; yv[u][j] = 0; }
                }
            }
        };
        auto consume = [&](int i, const T (&sv)[U][N], const T (&yv)[U][N]) {
            double t[kGramValues] = {0, 0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const double sx = (double)sv[u][j], yx = (double)yv[u][j];
                    t[0] = __builtin_fma(sx, (double)gv[u][j], t[0]);
                    t[1] = __builtin_fma(yx, (double)gv[u][j], t[1]);
                    t[2] = __builtin_fma(yx, (double)ypv[u][j], t[2]);
                    t[3] = __builtin_fma(yx, (double)spv[u][j], t[3]);
                    t[4] = __builtin_fma(sx, (double)ypv[u][j], t[4]);
                }
            }
            double tot[kGramValues];
            wave_sum5(t, lane, tot);
            if (lane == i) {
#pragma unroll
                for (int c = 0; c < kGramValues; ++c) acc[c] += tot[c];
            }
        };
        if (k > 0) {
            if (p.pivot_first) {
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int j = 0; j < N; ++j) { sA[u][j] = spv[u][j]; yA[u][j] = ypv[u][j]; }
            } else {
                fetch(0, sA, yA);
            }
        }
        // The steady-state loop has NO condition around its fetches: a branch there is a control-flow
        // join at which the compiler's s_waitcnt insertion must assume the fetch was skipped, and every
        // consume would then also wait for the pair just requested (vmcnt retires in order).  (At this
        // kernel's 16 waves per CU the occupancy hides it either way: 478 vs 477 us at config 3.)
        int i = 0;
        for (; i + 2 < k; i += 2) {
            fetch(i + 1, sB, yB);
            __builtin_amdgcn_sched_barrier(0);              // (keeps the consumer's fmas, and with them its
            consume(i, sA, yA);                             //  waits, below the requests)
            fetch(i + 2, sA, yA);
            __builtin_amdgcn_sched_barrier(0);
            consume(i + 1, sB, yB);
        }
        if (i + 1 < k) {                                    // two pairs left
            fetch(i + 1, sB, yB);
            consume(i, sA, yA);
            consume(i + 1, sB, yB);
        } else if (i < k) {                                 // one pair left
            consume(i, sA, yA);
        }
    };
    const int64_t tile_v = (int64_t)kBlock * U;
    const int64_t full_tiles = nvec / tile_v;
    // ragged last tile: taken FIRST, by the last block (which has the fewest full tiles).  Run
    // after the full tiles it was a slow predicated tile at the very end of one block's work:
    // the whole launch waited ~50 us for it (the Gram pass measured 579 us in step! against
    // 526 us for the same code without a ragged tile in tools/grambench.hip).
    if (full_tiles * tile_v < nvec && blockIdx.x == gridDim.x - 1) do_tile(full_tiles * tile_v, std::false_type{});
    // (Per-wave clocks at config 3: the blocks that arrived first on a CU get through their tiles at nearly twice the pace
    // of the last ones -- a SIMD serves the oldest of its waves first -- and are done at 283 us against 514 us.  A priority
    // that falls with a block's progress evens that out, 491 ... 528 us, and the kernel takes exactly as long, 550 us: the
    // pass is bound by the memory system, which does not care whose requests it serves.  Not kept.)
    if (p.peel) { for (int64_t tile = blockIdx.x; tile < full_tiles; tile += gridDim.x) do_tile(tile * tile_v, std::true_type{}); }
    else { for (int64_t tile = blockIdx.x; tile < full_tiles; tile += gridDim.x) do_tile(tile * tile_v, std::false_type{}); }
    (void)nthreads;
    // scalar tail (n not a multiple of the vector width): lane i of wave 0 in block 0 owns pair i
    if constexpr (POST) {
        // (the tail elements of the new pair, once, before anybody's dots read them: this wave is the only one that does)
        if (VEC && blockIdx.x == 0 && wave == 0 && lane == 0) {
            for (int64_t e = nvec * N; e < p.n; ++e) { p.sp_out[e] = p.x[e] - sp[e]; p.yp_out[e] = p.g[e] - p.g_old[e]; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    if (VEC && blockIdx.x == 0 && wave == 0 && lane < k) {
        for (int64_t e = nvec * N; e < p.n; ++e) {
            const double ge = (double)p.g[e], spe = (double)(POST ? p.sp_out[e] : sp[e]), ype = (double)(POST ? p.yp_out[e] : yp[e]);
            const double sx = (double)p.s[lane][e], yx = (double)p.y[lane][e];
            acc[0] = __builtin_fma(sx, ge, acc[0]);
            acc[1] = __builtin_fma(yx, ge, acc[1]);
            acc[2] = __builtin_fma(yx, ype, acc[2]);
            acc[3] = __builtin_fma(yx, spe, acc[3]);
            acc[4] = __builtin_fma(sx, ype, acc[4]);
        }
    }
    // combine the four waves of the block in a fixed order, then one partial per (value, block)
    __shared__ double wacc[kWaves][kMaxHistory][kGramValues];
    if (lane < k) {
#pragma unroll
        for (int c = 0; c < kGramValues; ++c) wacc[wave][lane][c] = acc[c];
    }
    __syncthreads();
    if (wave == 0 && lane < k) {
#pragma unroll
        for (int c = 0; c < kGramValues; ++c) {
            const double r = (wacc[0][lane][c] + wacc[1][lane][c]) + (wacc[2][lane][c] + wacc[3][lane][c]);
            p.partials[(int64_t)(lane * kGramValues + c) * gridDim.x + blockIdx.x] = r;
        }
    }
    dev_stamp(p.debug, 1);
}

template <typename T> struct CombineParams {
    int debug;
    int64_t n;
    const T *g;
    T *d;
    int k;
    int fresh_plain;            // 1: newest pair (written by the previous step's tail) with plain loads
    const double *alpha, *coef, *scale;
    const T *s[kMaxHistory];    // logical pair -> slot base
    const T *y[kMaxHistory];
    int64_t rowbytes;           // blocked ring (BLK kernels only)
};

// d[e] = the reference's elementwise recurrence (:438-449) with the scalars already known:
//   q = g[e]; q = fma(-alpha_i, y_i[e], q) (i = 1..k); q *= scale; q = fma(-c_i, s_i[e], q)
//   (i = k..1).  Given equal scalars this is bit-identical to the reference's d.
// The 2k coefficients are staged once per block in LDS (wave-uniform broadcast reads), the
// slot pointers come from the kernel-argument segment (scalar loads), and the history is
// streamed with non-temporal 16-B loads, UI of them in flight per stream step.
template <typename T, bool VEC, int U, bool NTS, bool BLK = false>
__global__ __launch_bounds__(kBlock) void combine_kernel(CombineParams<T> p) {
    using L = Ld<T, VEC>;
    constexpr int N = L::N;
    __shared__ T a_s[kMaxHistory], c_s[kMaxHistory];
    const int k = p.k;
    if (threadIdx.x < k) {
        a_s[threadIdx.x] = (T)(-p.alpha[threadIdx.x]);
        c_s[threadIdx.x] = (T)(-p.coef[threadIdx.x]);
    }
    __syncthreads();
    const T scale = k > 0 ? (T)p.scale[0] : (T)1;
    const int64_t nvec = p.n / N;
    const int64_t nthreads = (int64_t)gridDim.x * kBlock;
    dev_stamp(p.debug, 0);
    for (int64_t base = (int64_t)blockIdx.x * kBlock * U; base < nvec; base += nthreads * U) {
        T q[U][N];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t vi = base + (int64_t)u * kBlock + threadIdx.x;
            ok[u] = vi < nvec;
            if (ok[u]) L::load(p.g + vi * N, q[u]);
        }
#pragma unroll 4
        for (int i = 0; i < k; ++i) {
            const T a = a_s[i];
            const T *yi = p.y[i];
            T v[U][N];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (ok[u]) {
                    if (p.fresh_plain && i == 0) L::load(hist_ptr<BLK>(yi, base + (int64_t)u * kBlock + threadIdx.x, p.rowbytes), v[u]);
                    else L::load_nt(hist_ptr<BLK>(yi, base + (int64_t)u * kBlock + threadIdx.x, p.rowbytes), v[u]);
                }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < N; ++j) q[u][j] = dfma(a, v[u][j], q[u][j]);
        }
        if (k > 0) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < N; ++j) q[u][j] = scale * q[u][j];
        }
#pragma unroll 4
        for (int i = k - 1; i >= 0; --i) {
            const T c = c_s[i];
            const T *si = p.s[i];
            T v[U][N];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (ok[u]) {
                    if (p.fresh_plain && i == 0) L::load(hist_ptr<BLK>(si, base + (int64_t)u * kBlock + threadIdx.x, p.rowbytes), v[u]);
                    else L::load_nt(hist_ptr<BLK>(si, base + (int64_t)u * kBlock + threadIdx.x, p.rowbytes), v[u]);
                }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < N; ++j) q[u][j] = dfma(c, v[u][j], q[u][j]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (ok[u]) {
                T *dst = p.d + (base + (int64_t)u * kBlock + threadIdx.x) * N;
                if constexpr (NTS) L::store_nt(dst, q[u]); else L::store(dst, q[u]);
            }
    }
    if constexpr (VEC) {
        const int64_t e = nvec * N + (int64_t)blockIdx.x * kBlock + threadIdx.x;
        if (e < p.n) {
            T q = p.g[e];
            for (int i = 0; i < k; ++i) q = dfma(a_s[i], p.y[i][e], q);
            if (k > 0) q = scale * q;
            for (int i = k - 1; i >= 0; --i) q = dfma(c_s[i], p.s[i][e], q);
            p.d[e] = q;
        }
    }
    dev_stamp(p.debug, 1);
}

// ============================================================================ single-pass step
// ONE pass over the history per accepted step (built-in chained Rosenbrock objective).
//
// The two-loop needs two dependent sweeps over (S, Y): the dot products with the new gradient
// (Gram pass) and the combine pass that forms d.  They cannot be fused WITHIN a step, but the
// combine pass of step t and the Gram pass of step t+1 can: with the history of a tile held in
// registers the kernel forms d (:438-449), the first trial point x + 1*d (:124), the objective
// terms of that point, its gradient (the stencil needs one neighbour on each side, taken from
// overlapping wave-rows), delta_point (:145), delta_gradient (:478-480) and -- still from the
// same registers -- every dot product the NEXT two-loop needs against the new g / s / y.  If the
// device-side decision then accepts the trial (the common case: 1.18 objective evaluations per
// step at config 3) the step is complete and the next direction's scalars follow from a
// reduce + finish launch; if it rejects, the host falls back to the ordinary kernels with the
// halved step.  Traffic per accepted step: (2k+2) reads + 7 writes instead of (4k+8) + 6.
//
// Layout: a wave owns a row of 64 consecutive 16-B vectors of which the two end vectors are
// halos (recomputed, not stored), rows overlap by two vectors.  x and g are updated in place by
// their owner lanes; the halo lanes take x_old / g_old from a snapshot of the row-boundary
// vectors made by halo_snapshot_kernel just before (so no wave reads an element another wave
// may already have overwritten).  x_old (:118) and g_old are also written to backups, needed
// when the trial is rejected.
// wave-row geometry: kRowOwn owned vectors in lanes [kRowLead, kRowLead + kRowOwn), one halo lane on
// each side.  62 uses the whole wave.  (56 owned vectors = seven whole 128-B lines per stream, so
// that rows start on line boundaries, measured slower: 770 vs 757 us -- 11 % more rows and
// redundant halo loads cost more than the partial first / last line of every row.)
constexpr int kPairMaxK = 20;                  // pairs the single-pass step over a PAIR ring holds (its largest instantiation)
// points the point pass holds: K = 20 is what fits two waves per SIMD (247 of 256 registers); fp64 has one more
// instantiation, K = 24 on two register sets and one wave per SIMD (508 of 512), so that m = 21 .. 24 do not fall
// back to the two-pass kernels (n = 1e7, m = 24: 683 step!()/s there)
static inline int point_max_k(int32_t dtype) { return dtype == DZO_F64 ? 24 : 20; }
constexpr int kFusedMaxK = 24;                 // two register sets of 2k history vectors: 2*2*20 x 16 B per lane

// The decorators of legacy/DZOptimization.jl:219-296 as the point pass applies them (DEC instantiations), per element:
//   L2GradientWrapper (:247)            g += (lambda + lambda) x          after the stencil, for every point of the ring and the trial point
//   UniformBoxGradientWrapper (:289-294) g = 0 where the box pushes back   after that (the order of problem_grad_async / grad_decorate_kernel)
//   UniformBoxConstraint (:264-272)      x = clamp(x, lo, hi)             the trial point, AFTER the change test of :128 and before :138
//   L2RegularizationWrapper (:231-232)   f += lambda norm2(x)             two more partial sums per block (the trial point and the one at t/2)
// A decorator that is off has neutral bounds (-inf, +inf), so only the L2 gradient term needs its flag.
template <typename T> struct PointDecor {
    T two_lambda;                              // :247 lambda + lambda, rounded to T
    T bg_lo, bg_hi;
    T cons_lo, cons_hi;
    int l2_on;
};
template <typename T> __device__ __forceinline__ T decor_grad(const PointDecor<T> &d, T g, T x) {
    const T g2 = dfma(d.two_lambda, x, g);
    g = d.l2_on ? g2 : g;                                          // (not fma(0, x, g): that turns a gradient of -0 into +0)
    const bool pushed_back = (x <= d.bg_lo && g >= (T)0) || (x >= d.bg_hi && g <= (T)0);
    return pushed_back ? (T)0 : g;
}
template <typename T> __device__ __forceinline__ T decor_clamp(const PointDecor<T> &d, T v) {
    return v < d.cons_lo ? d.cons_lo : (v > d.cons_hi ? d.cons_hi : v);    // clamp(x[i], lo, hi) :269, as box_clamp_kernel
}

template <typename T> struct FusedParams {
    int64_t n;
    int k;                                     // pairs read (history before the push)
    int k_next;                                // pairs after the push = min(k + 1, m)
    T t;                                       // trial step size (1; 1/2 when the pass is re-run after a rejected first trial)
    T t_half;                                  // t/2 (rounded): the objective there rides along, obj_partials[gridDim.x + b]
    const T *x, *g;                            // current_point / current_gradient: READ ONLY in this pass
    T *x_out, *g_out;                          // the trial point and its gradient go to the twin buffers
    T *d;                                      // step_direction
    const double *alpha, *coef, *scale;
    // blocked ring: ONE base pointer; the tile of stream q in wave-row r is at ring + r * rowbytes + off(q).
    // Every address of a row is then (one scalar row base) + (a 32-bit uniform offset) + 16 * lane -- eighty
    // separate stream pointers times a 64-bit row offset each do not fit the scalar register file.
    T *ring;
    uint32_t rowbytes;                         // bytes between consecutive wave-rows of the ring
    uint32_t soff[kFusedMaxK + 1];             // logical pair -> byte offset of its s tile within a row (y tile: + kTileBytes);
                                               // point ring: logical POINT j (0 = current) -> its x tile (gradient tile: + kTileBytes)
    uint32_t new_off;                          // s tile of the spare slot: delta_point / delta_gradient of this step
    uint32_t ystride;                          // from a slot's s (x) tile to its y (g) tile: 1 KiB tile-major, one stream stream-major
    double *gram_partials;                     // [kGramValues * k_next][gridDim.x], post-push order
    double *obj_partials;                      // [gridDim.x]
    int32_t *changed;
    int nt_tiles;                              // point ring: non-temporal stores for the new point's tiles (when they do not fit the Infinity Cache)
    int stage_rows;                            // point ring: rows whose new tiles a wave collects in LDS before it writes them in one burst
    int store_d;                               // point ring: write step_direction (it is formed on demand otherwise, see lbfgs_materialize_d)
    int prio;                                  // point pass: per-phase issue priority (two waves per SIMD)
    int leftover_even;                         // point pass: the last, partial round of rows goes to the even XCDs' blocks first
    int debug_skip;                            // dev ablation only: 1 = no pair dots, 2 = no stores, 4 = no combine chain, 64 = no tile stores, 128 = no gradient-tile stores (point pass)
    PointDecor<T> dec;                         // point pass, DEC instantiations: the decorators of legacy/DZOptimization.jl:219-296
    T obj_lambda;                              // point pass, OBJ = 1: the chained quadratic's lambda
};

// Every pointer a pass dereferences, checked on the host before the launch: a null here must be an error code, never a
// GPU memory fault (round 3 saw one "Memory access fault ... on address (nil)" from an uncommitted working tree of the
// edge-array experiment; a fault can reset every GPU of the host).  `pair`: the pair-ring pass also
// reads x / g and writes the twin buffers and d.
template <typename T> static inline bool fused_params_ok(const FusedParams<T> &fp, bool pair) {
    const bool common = fp.ring && fp.alpha && fp.coef && fp.scale && fp.gram_partials && fp.obj_partials && fp.changed && fp.rowbytes != 0 && fp.n > 0;
    return common && (!pair || (fp.x && fp.g && fp.x_out && fp.g_out && fp.d)) && (pair || fp.d);
}

// PLAIN: ablation build with plain instead of non-temporal history loads (DZO_TUNE_SP_DEBUG bit 256,
// fp64 K = 20 only); a run-time switch inside the kernel costs SGPRs the production kernel does not have
template <typename T, int K, bool PLAIN = false>
__global__ __launch_bounds__(kBlock, 1) void lbfgs_single_pass_kernel(FusedParams<T> p) {
    constexpr int N = Vec16<T>::N;
    constexpr int kOwn = kRowOwn, kLead = kRowLead;
    __shared__ T a_s[kFusedMaxK], c_s[kFusedMaxK];
    __shared__ double wacc[kWaves][kFusedMaxK + 1][kGramValues];
    __shared__ double lds[kWaves];
    __shared__ int lds_flag;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int k = p.k, kn = p.k_next;          // k <= K
    // pairs i >= k (history not full yet) get a zero coefficient and the host points their table
    // entries at pair 0, so that every load below is unconditional straight-line code
    if (threadIdx.x < K) {
        a_s[threadIdx.x] = threadIdx.x < k ? (T)(-p.alpha[threadIdx.x]) : (T)0;
        c_s[threadIdx.x] = threadIdx.x < k ? (T)(-p.coef[threadIdx.x]) : (T)0;
    }
    __syncthreads();
    const T scale = k > 0 ? (T)p.scale[0] : (T)1;
    const int64_t nvec = p.n / N;
    const int64_t rows = (nvec + kOwn - 1) / kOwn;
    const int64_t stride = (int64_t)gridDim.x * kWaves;
    double acc[kGramValues];
#pragma unroll
    for (int c = 0; c < kGramValues; ++c) acc[c] = 0;
    double fobj = 0, fobj_h = 0;
    bool diff = false;

    // Lanes beyond either end of x read a clamped (valid) address: their values only ever act as
    // the outer neighbour of x[0] / x[n-1], which the stencil ignores, and never reach a dot.
    // 32-bit byte offsets from wave-uniform bases (scalar base + vector offset addressing).
    auto byte_offset = [&](int64_t row) -> uint32_t {
        const int64_t v = row * kOwn - kLead + lane;
        const int64_t vc = v < 0 ? 0 : (v >= nvec ? nvec - 1 : v);
        return (uint32_t)(vc * (int64_t)sizeof(T) * N);
    };
    auto at = [](const T *base, uint32_t boff) { return reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + boff); };
    auto atw = [](T *base, uint32_t boff) { return reinterpret_cast<T *>(reinterpret_cast<char *>(base) + boff); };
    auto load_xg = [&](int64_t row, uint32_t boff, T (&xo)[N], T (&go)[N]) {
        // x and g are not written by this pass (the trial point goes to the twin buffers), so the halo
        // lanes simply read their neighbours' vectors
        load16(at(p.x, boff), xo);
        load16(at(p.g, boff), go);
    };

    // TWO register sets for the 2K history vectors of a row: all loads of row r+1 are in flight
    // while row r is computed (the kernel runs one wave per SIMD -- 2 x 2K x 16 B per lane leave no
    // room for a second resident wave -- so the overlap has to come from inside the wave).  Issuing
    // the next row's loads BEFORE this row's stores also keeps the store acknowledgements off the
    // critical path: vmcnt retires in issue order.  Measured at n = 1e7, k = 20: 845 us with one
    // register set refilled pair by pair during the dots, 537 us for the loads alone.
    // History loads: row r of every stream is one aligned 1-KiB tile, the tiles of a row adjacent; lane l
    // reads position l of the tile (position 0 / 63 hold the neighbouring rows' edge vectors).  The row base is
    // wave-uniform: it goes into the scalar base address, the vector offset is the constant 16 * lane.
    const uint32_t toff = (uint32_t)lane * 16u;
    auto rowbase = [&](int64_t row) {
        const uint32_t r = (uint32_t)__builtin_amdgcn_readfirstlane((int)row);       // rows < 2^31 (n T < 4 GiB)
        return reinterpret_cast<char *>(p.ring) + (uint64_t)r * p.rowbytes;
    };
    // (scalar row base) + (32-bit vector offset = tile offset + 16 * lane).  The empty asm keeps the tile offset
    // opaque inside the loop: hoisted, the forty loop-invariant sums would each pin a VGPR.
    auto tl = [&](char *rb, uint32_t uoff) {
        asm volatile("" : "+s"(uoff));
        return reinterpret_cast<const T *>(rb + (uint32_t)(uoff + toff));
    };
    auto issue = [&](int64_t row, uint32_t boff, T (&sv)[K][N], T (&yv)[K][N], T (&xo)[N], T (&go)[N]) {
        load_xg(row, boff, xo, go);
        char *rb = rowbase(row);
        if constexpr (PLAIN) {
#pragma unroll
            for (int i = 0; i < K; ++i) load16(tl(rb, p.soff[i] + p.ystride), yv[i]);
#pragma unroll
            for (int i = K - 1; i >= 0; --i) load16(tl(rb, p.soff[i]), sv[i]);
            return;
        }
        if (p.debug_skip & 16) { load16(tl(rb, p.soff[0] + p.ystride), yv[0]); load16(tl(rb, p.soff[0]), sv[0]); }
        else { load16_nt(tl(rb, p.soff[0] + p.ystride), yv[0]); load16_nt(tl(rb, p.soff[0]), sv[0]); }
#pragma unroll
        for (int i = 1; i < K; ++i) load16_nt(tl(rb, p.soff[i] + p.ystride), yv[i]);
#pragma unroll
        for (int i = K - 1; i >= 1; --i) load16_nt(tl(rb, p.soff[i]), sv[i]);
    };
    auto compute = [&](int64_t row, uint32_t boff, const T (&sv)[K][N], const T (&yv)[K][N], const T (&xo)[N], const T (&go)[N]) {
        const int64_t v = row * kOwn - kLead + lane;
        const bool valid = v >= 0 && v < nvec;
        const bool owner = valid && lane >= kLead && lane < kLead + kOwn;
        const int64_t e0 = v * N;
        // ---- d = the reference's elementwise recurrence (:438-449), as combine_kernel
        T q[N];
#pragma unroll
        for (int j = 0; j < N; ++j) q[j] = go[j];
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const T a = a_s[i];
#pragma unroll
            for (int j = 0; j < N; ++j) q[j] = dfma(a, yv[i][j], q[j]);
        }
        if (k > 0) {
#pragma unroll
            for (int j = 0; j < N; ++j) q[j] = scale * q[j];
        }
#pragma unroll
        for (int i = K - 1; i >= 0; --i) {
            const T c = c_s[i];
#pragma unroll
            for (int j = 0; j < N; ++j) q[j] = dfma(c, sv[i][j], q[j]);
        }
        // ---- first trial point (:124), change flag (:128)
        T xn[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            xn[j] = dfma(p.t, q[j], xo[j]);
            diff |= owner && !is_equal(xn[j], xo[j]);
        }
        // ---- the objective at HALF the step rides along (:152's next candidate): if this trial is rejected the
        //      host knows at once whether t/2 will be accepted, and then re-runs this pass with t/2
        {
            T xh[N];
#pragma unroll
            for (int j = 0; j < N; ++j) xh[j] = dfma(p.t_half, q[j], xo[j]);
            const T xhnext = lane_next<T>(xh[0]);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const T xq = j + 1 < N ? xh[(j + 1) % N] : xhnext;
                if (owner && e0 + j + 1 < p.n) fobj_h += rosen_term<T>(xh[j], xq);
            }
        }
        // ---- objective terms and gradient of the trial point; neighbours from the adjacent lanes
        const T xprev = lane_prev<T>(xn[N - 1]);
        const T xnext = lane_next<T>(xn[0]);
        T gn[N], sn[N], yn[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const T xp = j > 0 ? xn[(j + N - 1) % N] : xprev;
            const T xq = j + 1 < N ? xn[(j + 1) % N] : xnext;
            gn[j] = rosen_grad_elem<T>(e0 + j, p.n, xp, xn[j], xq);
            sn[j] = xn[j] - xo[j];                                  // :145
            yn[j] = gn[j] - go[j];                                  // :478-480
            if (owner && e0 + j + 1 < p.n) fobj += rosen_term<T>(xn[j], xq);
        }
        if (owner && !(p.debug_skip & 2)) {
            // non-temporal stores throughout: 7 n T of fresh dirty lines otherwise sit in the Infinity
            // Cache and their write-back lands on the next pass (measured inside step!: 910 -> 815 us)
            if (!(p.debug_skip & 128)) store16_nt(atw(p.d, boff), q);
            store16_nt(atw(p.x_out, boff), xn);
            store16_nt(atw(p.g_out, boff), gn);
            // the new pair goes straight into its tiles; the first / last owned vector of the row is also
            // the right / left halo copy of the neighbouring row's tile
            char *st = rowbase(row) + p.new_off;
            char *yt = st + p.ystride;
            if (!(p.debug_skip & 64)) {
                store16_nt(reinterpret_cast<T *>(st + toff), sn);
                store16_nt(reinterpret_cast<T *>(yt + toff), yn);
            }
            // (measured, n = 1e7 k = 20: the four one-lane halo stores cost 8 us of 690; the five output streams
            // ~25 us each, twice their bytes at the read rate -- write/read turnarounds in the HBM stacks; plain
            // instead of non-temporal stores are within the run-to-run noise inside step!)
            if (lane == kLead && row > 0) {
                store16(reinterpret_cast<T *>(st - (int64_t)p.rowbytes + 63 * 16), sn);
                store16(reinterpret_cast<T *>(yt - (int64_t)p.rowbytes + 63 * 16), yn);
            }
            if (lane == kLead + kOwn - 1 && row + 1 < rows) {
                store16(reinterpret_cast<T *>(st + p.rowbytes), sn);
                store16(reinterpret_cast<T *>(yt + p.rowbytes), yn);
            }
        }
        if (!owner) {
#pragma unroll
            for (int j = 0; j < N; ++j) { gn[j] = (T)0; sn[j] = (T)0; yn[j] = (T)0; }   // halos add nothing to the dots
        }
        // ---- dots of the NEXT two-loop (post-push order: new pair = 0, old pair i = i + 1)
        {
            double t5[kGramValues] = {0, 0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const double sx = (double)sn[j], yx = (double)yn[j], gx = (double)gn[j];
                t5[0] = __builtin_fma(sx, gx, t5[0]);
                t5[1] = __builtin_fma(yx, gx, t5[1]);
                t5[2] = __builtin_fma(yx, yx, t5[2]);
                t5[3] = __builtin_fma(yx, sx, t5[3]);
                t5[4] = __builtin_fma(sx, yx, t5[4]);
            }
            double tot[kGramValues];
            wave_sum5(t5, lane, tot);
            if (lane == 0) {
#pragma unroll
                for (int c = 0; c < kGramValues; ++c) acc[c] += tot[c];
            }
        }
#pragma unroll
        for (int i = 0; i < K; ++i) {
            if (i + 1 < kn && !(p.debug_skip & 1)) {
                double t5[kGramValues] = {0, 0, 0, 0, 0};
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const double sx = (double)sv[i][j], yx = (double)yv[i][j];
                    T gq = gn[j], yq = yn[j], sq = sn[j];
                    if constexpr (sizeof(T) == 4) {
                        // fp32: keep the compiler from holding fp64 copies of the new pair (12 doubles = 24
                        // VGPRs) live across the K pairs -- with them the K = 20 instantiation spills, and a
                        // scratch access in this loop costs far more than its bytes; the three
                        // conversions per pair and element are re-done instead
                        asm volatile("" : "+v"(gq), "+v"(yq), "+v"(sq));
                    }
                    t5[0] = __builtin_fma(sx, (double)gq, t5[0]);
                    t5[1] = __builtin_fma(yx, (double)gq, t5[1]);
                    t5[2] = __builtin_fma(yx, (double)yq, t5[2]);
                    t5[3] = __builtin_fma(yx, (double)sq, t5[3]);
                    t5[4] = __builtin_fma(sx, (double)yq, t5[4]);
                }
                double tot[kGramValues];
                wave_sum5(t5, lane, tot);
                if (lane == i + 1) {
#pragma unroll
                    for (int c = 0; c < kGramValues; ++c) acc[c] += tot[c];
                }
            }
        }
    };
    T sA[K][N], yA[K][N], xA[N], gA[N];
    T sB[K][N], yB[K][N], xB[N], gB[N];
    // The prefetch of the next row is UNCONDITIONAL (past the end it re-reads the last row): a branch
    // around the loads makes a control-flow join at which the compiler's s_waitcnt insertion must
    // assume the loads were skipped, so every wait on the current set would also wait for the set
    // just requested (vmcnt retires in order) and the two register sets would buy nothing.
    int64_t row = (int64_t)blockIdx.x * kWaves + wave;
    auto in_range = [&](int64_t r) { return r < rows ? r : rows - 1; };
    uint32_t boff = byte_offset(in_range(row));
    issue(in_range(row), boff, sA, yA, xA, gA);
    while (row < rows) {
        int64_t nrow = row + stride;
        uint32_t nboff = byte_offset(in_range(nrow));
        issue(in_range(nrow), nboff, sB, yB, xB, gB);
        compute(row, boff, sA, yA, xA, gA);
        row = nrow; boff = nboff;
        if (row >= rows) break;
        nrow = row + stride;
        nboff = byte_offset(in_range(nrow));
        issue(in_range(nrow), nboff, sA, yA, xA, gA);
        compute(row, boff, sB, yB, xB, gB);
        row = nrow; boff = nboff;
    }
    // ---- block results: Gram partials (fixed wave order), objective partial, change flag
    if (lane < kn) {
#pragma unroll
        for (int c = 0; c < kGramValues; ++c) wacc[wave][lane][c] = acc[c];
    }
    __syncthreads();
    if (wave == 0 && lane < kn) {
#pragma unroll
        for (int c = 0; c < kGramValues; ++c) {
            const double r = (wacc[0][lane][c] + wacc[1][lane][c]) + (wacc[2][lane][c] + wacc[3][lane][c]);
            p.gram_partials[(int64_t)(lane * kGramValues + c) * gridDim.x + blockIdx.x] = r;
        }
    }
    block_raise_flag(diff, p.changed, &lds_flag);
    const double fo = block_sum(fobj, lds);
    const double fh = block_sum(fobj_h, lds);
    if (threadIdx.x == 0) { p.obj_partials[blockIdx.x] = fo; p.obj_partials[gridDim.x + blockIdx.x] = fh; }
}

// ---------------------------------------------------------------------------- point ring
// The same pass over a ring that holds the last k + 1 POINTS and GRADIENTS instead of their differences: point 0
// is current_point / current_gradient, pair i is formed in registers as s_i = X_i - X_{i+1}, y_i = G_i - G_{i+1}
// (one subtraction per use: exactly the value the reference stored as delta_point / delta_gradient, :145, :480).
// The trial point and its gradient ARE the new ring entries, so the pass writes three streams (d, X, G) instead
// of five (d, x, g, delta_point, delta_gradient) and reads the same 2k + 2.  x and g are never written in place
// (the spare slot takes the trial), so a rejected trial just runs the pass again with a smaller t.
// FIRST: the first step! walks along the step_direction the constructor left (d0 = -(step / |g|) g, :386-387 -- a
// public field the caller may have changed): d is read instead of formed and not written back.
#ifndef DZO_PP_REGRAD
#define DZO_PP_REGRAD 1      // point pass: 1 = every lane recomputes the points' gradients from the point tiles (they are not streamed)
#endif
#ifndef DZO_PP_REFILL
#define DZO_PP_REFILL 1      // point pass: 1 = a register set is refilled tile by tile inside the dot-product loop; 0 = whole-set requests
#endif
// compile-time loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N - 1>)
template <int N, typename F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>());
    }
}

// SETS: 2 = two register sets of K + 1 points and gradients per wave (one wave per SIMD: 2 x 42 x 4 registers at K = 20);
// 1 = ONE set and two waves per SIMD (256 registers per wave): a wave's instruction stream is in order, so whenever one
// of its loads cannot issue (the memory pipeline is backed up -- the steady state of a bandwidth-bound sweep) or an
// instruction waits for a result, the SIMD idles unless a second wave is there to take the slot.
// The stencils carry no index tests (10 of a stencil's 34 instructions): what rosen_grad_elem decides from the element's
// index -- only the first and the last element of the vector differ -- is a set of per-element coefficients formed once
// per wave-row (RosenCoef, dzo_rosen.h), and every row runs the same straight-line code.
// (Measured and not kept, round 3: the small instantiations built for THREE blocks per CU -- K = 8 fits 149 registers, K = 12
// needs 56 bytes of scratch at 168 -- with 11 instead of 16 staged rows each: n = 1e7, m = 5: 141 against 144 us per pass,
// m = 8: 161 against 160, m = 10 / 12 (K = 12): 222 / 235 against 203 / 214; n = 1e6 slower throughout.  What did help the
// small instantiations is asking the occupancy query with the dynamic LDS the launch really uses, see points_grid.)
// DEC: the decorators ride along (PointDecor above); separate instantiations, so that the undecorated pass is untouched.
// OBJ: the chained objective (ChainObj, dzo_rosen.h): 0 the chained Rosenbrock, 1 the chained quadratic -- the pass asks
// the objective for per-element coefficients, the gradient stencil and the objective terms and knows nothing else of it.
template <typename T, int K, bool FIRST = false, int SETS = 2, bool DEC = false, int OBJ = 0>
__global__ __launch_bounds__(kBlock, (SETS == 1 ? 2 : 1)) void lbfgs_point_pass_kernel(FusedParams<T> p) {
    using Obj = ChainObj<T, OBJ>;
    constexpr int N = Vec16<T>::N;
    constexpr int kOwn = kRowOwn, kLead = kRowLead;
    __shared__ T a_s[kFusedMaxK], c_s[kFusedMaxK];
    __shared__ double wacc[kWaves][kFusedMaxK + 1][kGramValues];
    __shared__ double lds[kWaves];
    __shared__ int lds_flag;
    __shared__ int64_t stage_row_s[kWaves][40];                 // the rows of the batch being collected (at most 36)
    // The new point's tiles are not written row by row: a wave collects them in its own slice of LDS and writes
    // stage_rows rows in one burst.  HBM pays for every switch between reading and writing: with this pass's
    // memory shape alone (tools/pointbench.hip: 42 tile reads per row, no arithmetic) the sweep takes 495 us
    // without writes, 650 us with the two tiles written row by row and 609 us with bursts of 16 rows.
    extern __shared__ __attribute__((aligned(16))) char stage_lds[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int stage_rows = p.stage_rows;
    // (a pass that recomputes the gradients never reads a gradient tile: it does not write one either -- the host forms
    // the tiles of a point's gradient when somebody asks for them, lbfgs_ensure_g)
    constexpr bool kWriteG = !(DZO_PP_REGRAD != 0 && !FIRST);
    constexpr int kStageTiles = kWriteG ? 2 : 1;
    char *stage = stage_lds + (size_t)wave * stage_rows * (kStageTiles * kTileBytes);
    int staged = 0;
    const int k = p.k, kn = p.k_next;          // k <= K
    // pairs i >= k get a zero coefficient, and the host points the table entries of the points beyond k at point
    // k, so that those pairs are X_k - X_k = 0: every load below is unconditional straight-line code
    if (threadIdx.x < K) {
        a_s[threadIdx.x] = threadIdx.x < k ? (T)(-p.alpha[threadIdx.x]) : (T)0;
        c_s[threadIdx.x] = threadIdx.x < k ? (T)(-p.coef[threadIdx.x]) : (T)0;
    }
    // REGRAD: the gradients of the K + 1 points are not streamed at all.  G_j is a 3-point stencil of X_j and a tile of
    // X_j carries the neighbours of all its 62 owned vectors, so every lane evaluates rosen_grad_elem -- the function that
    // produced the stored gradients, on the same operands, so the same bits -- and the pass reads K + 1 tiles per row
    // instead of 2 (K + 1), for K + 1 stencil evaluations.  The two HALO lanes lack the neighbour on their outer side,
    // and do not need it: all anybody takes from lane 0 is the LAST element of its trial point (the left neighbour of
    // lane 1's first element) and from lane 63 the FIRST (right neighbour of lane 62's last), i.e. of their gradients
    // only the inner element, whose stencil lies inside the tile.  Their outer elements carry finite garbage that
    // nothing reads (halo lanes store nothing and add zeros to every sum).
    constexpr bool kRegrad = DZO_PP_REGRAD != 0 && !FIRST;      // (the first step, once per optimizer, streams its one gradient)
    __syncthreads();
    const bool scaled = k > 0;
    const T scale = scaled ? (T)p.scale[0] : (T)1;
    const int64_t nvec = (p.n + N - 1) / N;               // (a ragged n: the last vector is padded with phantom elements, see load_vec_tail)
    const int64_t rows = (nvec + kOwn - 1) / kOwn;
    const int64_t row_end = rows;
    const int64_t stride = (int64_t)gridDim.x * kWaves;     // wave-rows block-cyclically
    const int pstride = (int)gridDim.x;
    const int pcol = (int)blockIdx.x;                       // column of this block's partial sums
    // the 5 (K + 1) dot-product partials of a wave-row go through a streaming transposed reduction (TreeSum,
    // dzo_common.h): ~5 VALU instructions per value, and the wave-wide total of value v accumulates in ONE lane
    // (a 9-exchange butterfly + 10 v_readlane + 5 masked adds per pair was half of this kernel's VALU work)
    TreeSum<kGramValues * (K + 1)> dots;
    dots.init();
    double fobj = 0, fobj_h = 0;
    double fsq = 0, fsq_h = 0;                                  // DEC: sum of squares of the trial point / the point at t/2 (:232)
    bool diff = false;
    auto byte_offset = [&](int64_t row) -> uint32_t {           // of the lane's vector in the contiguous d
        const int64_t v = row * kOwn - kLead + lane;
        const int64_t vc = v < 0 ? 0 : (v >= nvec ? nvec - 1 : v);
        return (uint32_t)(vc * (int64_t)sizeof(T) * N);
    };
    auto atw = [](T *base, uint32_t boff) { return reinterpret_cast<T *>(reinterpret_cast<char *>(base) + boff); };
    const uint32_t toff = (uint32_t)lane * 16u;
    auto rowbase = [&](int64_t row) {
        const uint32_t r = (uint32_t)__builtin_amdgcn_readfirstlane((int)row);       // rows < 2^31 (n T < 4 GiB)
        return reinterpret_cast<char *>(p.ring) + (uint64_t)r * p.rowbytes;
    };
    auto tl = [&](char *rb, uint32_t uoff) {
        asm volatile("" : "+s"(uoff));                           // (see lbfgs_single_pass_kernel)
        return reinterpret_cast<const T *>(rb + (uint32_t)(uoff + toff));
    };
    // the staged rows (stage_row_s) to the spare slot's tiles: the owned vectors, and the
    // first / last owned vector of a row also as the right / left halo copy of the neighbouring row's tile.
    // PLAIN stores while the two streams fit the Infinity Cache (the next pass reads these tiles first: measured
    // inside step! at n = 1e7, k = 20, 695 us against 735 us with non-temporal stores -- the opposite of the pair
    // ring, whose five streams of non-reused outputs are better kept out of the cache); non-temporal beyond
    // (n = 3e7, 480 MB per pass: plain stores lose 8 %)
    auto flush_stage = [&]() {
        // (stage_row_s is written by lane 0 and read by every lane of the same wave: a wave's LDS operations execute in
        // order, and the wavefront-scope fence pair -- no instruction, an ordering for the compiler -- makes the hand-off
        // well defined in the memory model too)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int b = 0; b < staged; ++b) {
            const int64_t r = stage_row_s[wave][b];
            const int64_t v = r * kOwn - kLead + lane;
            const bool own = v >= 0 && v < nvec && lane >= kLead && lane < kLead + kOwn;
            T xs[N], gs[N];
            const char *sl = stage + (size_t)b * (kStageTiles * kTileBytes) + toff;
            load16(reinterpret_cast<const T *>(sl), xs);
            if constexpr (kWriteG) load16(reinterpret_cast<const T *>(sl + kTileBytes), gs);
            if (own) {
                char *xt = rowbase(r) + p.new_off;
                char *gt = xt + p.ystride;
                const bool with_g = kWriteG;
                if (p.nt_tiles) {
                    store16_nt(reinterpret_cast<T *>(xt + toff), xs);
                    if (with_g) store16_nt(reinterpret_cast<T *>(gt + toff), gs);
                } else {
                    store16(reinterpret_cast<T *>(xt + toff), xs);
                    if (with_g) store16(reinterpret_cast<T *>(gt + toff), gs);
                }
                if (lane == kLead && r > 0) {
                    store16(reinterpret_cast<T *>(xt - (int64_t)p.rowbytes + 63 * 16), xs);
                    if (with_g) store16(reinterpret_cast<T *>(gt - (int64_t)p.rowbytes + 63 * 16), gs);
                }
                if (lane == kLead + kOwn - 1 && r + 1 < rows) {
                    store16(reinterpret_cast<T *>(xt + p.rowbytes), xs);
                    if (with_g) store16(reinterpret_cast<T *>(gt + p.rowbytes), gs);
                }
            }
        }
        staged = 0;
    };
    // K + 1 points and K + 1 gradients of a row; two register sets (as lbfgs_single_pass_kernel)
    auto issue = [&](int64_t row, T (&xv)[K + 1][N], T (&gv)[K + 1][N]) {
        // (the order compute() refills a set in: the waits the compiler derives for the loop are then the same on
        // the first trip as on every other)
        char *rb = rowbase(row);
#pragma unroll
        for (int j = 0; j <= K; ++j) {
            if constexpr (!kRegrad) load16_nt(tl(rb, p.soff[j] + p.ystride), gv[j]);
            else {                                               // (defined on entry to the loop; compute() overwrites it)
#pragma unroll
                for (int e = 0; e < N; ++e) gv[j][e] = (T)0;
            }
            load16_nt(tl(rb, p.soff[j]), xv[j]);
        }
        if constexpr (DZO_PP_REFILL != 0) __builtin_amdgcn_sched_barrier(0);   // set by set: nothing of the next set moves up into this one
    };
    // refill: the row this register set serves next (two rows ahead).  Its tiles are requested from inside the
    // dot-product loop, point i once pair i -- the last reader of that point's registers -- is done, instead of
    // whole-set requests in front of the other set's compute (issue(next); compute(cur)): -10 us at config 3.
    auto compute = [&](int64_t row, uint32_t boff, T (&xv)[K + 1][N], T (&gv)[K + 1][N], int64_t refill_row) {
        char *nrb = rowbase(refill_row);
        constexpr bool kRefill = DZO_PP_REFILL != 0;
        const int64_t v = row * kOwn - kLead + lane;
        const bool valid = v >= 0 && v < nvec;
        const bool owner = valid && lane >= kLead && lane < kLead + kOwn;
        const int64_t e0 = v * N;
        // dev experiment (priority between the two waves of a SIMD; the older wave otherwise always wins the vector issue)
        // Issue priority between the two waves of a SIMD (SETS == 1).  The SIMD picks by priority, then AGE: left alone, the
        // wave of the block that arrived first wins every contested slot, runs at twice its partner's pace, and finishes
        // its rows 100 us early -- per-wave clocks: blocks 0..255 done at 237 us, blocks 256..511 at 335 us, the last
        // third of the sweep running at one wave per SIMD.  So a wave is low while it forms the direction (it has
        // nothing in flight then anyway) and high in the dot-product phase, where the next row's tiles are requested
        // (337 -> 322 us), and the younger half of the grid one level higher still there (-> 316 us, both halves
        // done within 1 us of each other).  Scheduling only: rows, sums and results are what they were.
        if (p.prio) __builtin_amdgcn_s_setprio(0);
        typename Obj::Coef rc[N];                                 // the objective's index tests as coefficients, once per row
#pragma unroll
        for (int e = 0; e < N; ++e) rc[e] = Obj::coef(e0 + e, p.n, p.obj_lambda);
        if constexpr (kRegrad) {
            // the gradients of this row's K + 1 points, every lane (a halo lane's outer element: see above)
#pragma unroll
            for (int j = 0; j <= K; ++j) {
                const T xp = lane_prev0<T>(xv[j][N - 1]);
                const T xq = lane_next0<T>(xv[j][0]);
#pragma unroll
                for (int e = 0; e < N; ++e) {
                    const T xl = e > 0 ? xv[j][(e + N - 1) % N] : xp;
                    const T xr = e + 1 < N ? xv[j][(e + 1) % N] : xq;
                    gv[j][e] = Obj::grad(rc[e], xl, xv[j][e], xr);
                    if constexpr (DEC) gv[j][e] = decor_grad<T>(p.dec, gv[j][e], xv[j][e]);
                }
                __builtin_amdgcn_sched_barrier(0);               // point by point (the temporaries of 21 stencils at once do not fit)
            }
        }
        // ---- d = the reference's elementwise recurrence (:438-449)
        T q[N];
        if constexpr (FIRST) {
            load16(reinterpret_cast<const T *>(reinterpret_cast<const char *>(p.d) + boff), q);   // (halo lanes: their neighbours' d)
        } else {
#pragma unroll
            for (int j = 0; j < N; ++j) q[j] = gv[0][j];
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const T a = a_s[i];
#pragma unroll
                for (int j = 0; j < N; ++j) q[j] = dfma(a, gv[i][j] - gv[i + 1][j], q[j]);
            }
            if (scaled) {
#pragma unroll
                for (int j = 0; j < N; ++j) q[j] = scale * q[j];
            }
#pragma unroll
            for (int i = K - 1; i >= 0; --i) {
                const T c = c_s[i];
#pragma unroll
                for (int j = 0; j < N; ++j) q[j] = dfma(c, xv[i][j] - xv[i + 1][j], q[j]);
            }
        }
        // ---- trial point (:124), change flag (:128)
        T xn[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            xn[j] = dfma(p.t, q[j], xv[0][j]);
            diff |= owner && !is_equal(xn[j], xv[0][j]);
            // :134-135 the projection into the box, after the change test (the phantom padding of a ragged n stays +0)
            if constexpr (DEC) { const T cl = decor_clamp<T>(p.dec, xn[j]); xn[j] = e0 + j < p.n ? cl : xn[j]; }
        }
        // ---- the objective at HALF the step rides along (:152's next candidate)
        {
            T xh[N];
#pragma unroll
            for (int j = 0; j < N; ++j) {
                xh[j] = dfma(p.t_half, q[j], xv[0][j]);
                if constexpr (DEC) {
                    const T cl = decor_clamp<T>(p.dec, xh[j]); xh[j] = e0 + j < p.n ? cl : xh[j];
                    if (owner) fsq_h = __builtin_fma((double)xh[j], (double)xh[j], fsq_h);      // :232 norm2(x), as sumsq_kernel
                }
            }
            const T xhnext = lane_next0<T>(xh[0]);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const T xq = j + 1 < N ? xh[(j + 1) % N] : xhnext;
                if (owner && Obj::has_term(e0 + j, p.n)) fobj_h += Obj::term(rc[j], xh[j], xq);
            }
        }
        // ---- objective terms and gradient of the trial point; neighbours from the adjacent lanes
        const T xprev = lane_prev0<T>(xn[N - 1]);
        const T xnext = lane_next0<T>(xn[0]);
        T gn[N], sn[N], yn[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const T xp = j > 0 ? xn[(j + N - 1) % N] : xprev;
            const T xq = j + 1 < N ? xn[(j + 1) % N] : xnext;
            gn[j] = Obj::grad(rc[j], xp, xn[j], xq);
            if constexpr (DEC) {
                gn[j] = decor_grad<T>(p.dec, gn[j], xn[j]);
                if (owner) fsq = __builtin_fma((double)xn[j], (double)xn[j], fsq);
            }
            sn[j] = xn[j] - xv[0][j];                               // :145
            yn[j] = gn[j] - gv[0][j];                               // :478-480
            if (owner && Obj::has_term(e0 + j, p.n)) fobj += Obj::term(rc[j], xn[j], xq);
        }
        if (!(p.debug_skip & 2)) {
            if constexpr (!FIRST) { if (owner && p.store_d) store16_nt(atw(p.d, boff), q); }
            if (!(p.debug_skip & 64)) {
                // the trial point and its gradient into this wave's LDS slice (every lane: the slice is private to
                // the wave, no barrier); they reach the spare slot's tiles in flush_stage()
                if (lane == 0) stage_row_s[wave][staged] = row;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                char *sl = stage + (size_t)staged * (kStageTiles * kTileBytes) + toff;
                store16(reinterpret_cast<T *>(sl), xn);
                if constexpr (kWriteG) store16(reinterpret_cast<T *>(sl + kTileBytes), gn);
                staged += 1;
            }
        }
        if (!owner) {
#pragma unroll
            for (int j = 0; j < N; ++j) { gn[j] = (T)0; sn[j] = (T)0; yn[j] = (T)0; }   // halos add nothing to the dots
        }
        // ---- dots of the NEXT two-loop (post-push order: new pair = 0, old pair i = i + 1); value 5 j + c of the row
        const bool want_dots = !(p.debug_skip & 1);
        if (p.prio) { if (blockIdx.x >= gridDim.x / 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1); }
        if (want_dots) {
            double t5[kGramValues] = {0, 0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const double sx = (double)sn[j], yx = (double)yn[j], gx = (double)gn[j];
                t5[0] = __builtin_fma(sx, gx, t5[0]);
                t5[1] = __builtin_fma(yx, gx, t5[1]);
                t5[2] = __builtin_fma(yx, yx, t5[2]);
                t5[3] = __builtin_fma(yx, sx, t5[3]);
                t5[4] = __builtin_fma(sx, yx, t5[4]);
            }
            dots.template push<0>(t5[0], lane); dots.template push<1>(t5[1], lane); dots.template push<2>(t5[2], lane);
            dots.template push<3>(t5[3], lane); dots.template push<4>(t5[4], lane);
        }
        auto pair_dots = [&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if (want_dots) {
                double t5[kGramValues] = {0, 0, 0, 0, 0};
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    // the pair is formed again here rather than kept from the combine chains above: 2K difference
                    // vectors held across the stencil would not fit the register file (the empty asm keeps the
                    // compiler from merging the two subtractions)
                    T xa = xv[i][j], ga = gv[i][j];
                    asm volatile("" : "+v"(xa), "+v"(ga));
                    const double sx = (double)(xa - xv[i + 1][j]), yx = (double)(ga - gv[i + 1][j]);
                    T gq = gn[j], yq = yn[j], sq = sn[j];
                    if constexpr (sizeof(T) == 4) asm volatile("" : "+v"(gq), "+v"(yq), "+v"(sq));   // (see lbfgs_single_pass_kernel)
                    t5[0] = __builtin_fma(sx, (double)gq, t5[0]);
                    t5[1] = __builtin_fma(yx, (double)gq, t5[1]);
                    t5[2] = __builtin_fma(yx, (double)yq, t5[2]);
                    t5[3] = __builtin_fma(yx, (double)sq, t5[3]);
                    t5[4] = __builtin_fma(sx, (double)yq, t5[4]);
                }
                constexpr int v0 = kGramValues * (i + 1);
                dots.template push<v0 + 0>(t5[0], lane); dots.template push<v0 + 1>(t5[1], lane); dots.template push<v0 + 2>(t5[2], lane);
                dots.template push<v0 + 3>(t5[3], lane); dots.template push<v0 + 4>(t5[4], lane);
            }
            // point i is dead: its registers take the same tiles of the row two ahead
            if constexpr (kRefill) {
                if constexpr (!kRegrad) load16_nt(tl(nrb, p.soff[i] + p.ystride), gv[i]);
                load16_nt(tl(nrb, p.soff[i]), xv[i]);
            }
        };
        static_for<K>(pair_dots);
        if constexpr (kRefill) {
            if constexpr (!kRegrad) load16_nt(tl(nrb, p.soff[K] + p.ystride), gv[K]);
            load16_nt(tl(nrb, p.soff[K]), xv[K]);
        }
        if (want_dots) dots.finish_row(lane);
        if constexpr (kRefill) __builtin_amdgcn_sched_barrier(0);   // (the other set's compute starts below this set's last request)
    };
    T xA[K + 1][N], gA[K + 1][N];
    T xB[SETS == 2 ? K + 1 : 1][N], gB[SETS == 2 ? K + 1 : 1][N];
    int64_t row = (int64_t)blockIdx.x * kWaves + wave;
    auto in_range = [&](int64_t r) { return r < rows ? r : rows - 1; };   // (past the end: the last row again, unconditionally)
    if constexpr (SETS == 1) {
        // The rows that do not fill a last round of all waves go to the blocks of the EVEN XCDs first (block b runs on XCD
        // b mod 8): per-wave clocks show the odd XCDs 3 % behind the even ones on every MI355X seen, and a 40th row on
        // one of their waves is what the kernel ends on.  (A guess about the machine that costs nothing when it is wrong:
        // somebody has to take those rows.)
        const int64_t full = rows / stride, rem = rows - full * stride;
        const int xcd = (int)(blockIdx.x & 7), per = (int)(blockIdx.x >> 3);
        const int64_t even_blocks = (int64_t)gridDim.x / 2;
        // (a bijection of the blocks only when the grid is a multiple of 8: otherwise the plain order)
        const int64_t rank = (p.leftover_even && gridDim.x % 8 == 0)
                                 ? ((xcd & 1) ? even_blocks + (xcd >> 1) + 4 * (int64_t)per : (xcd >> 1) + 4 * (int64_t)per)
                                 : (int64_t)blockIdx.x;
        auto row_at = [&](int64_t i) -> int64_t {
            if (i < full) return i * stride + (int64_t)blockIdx.x * kWaves + wave;
            const int64_t r = rank * kWaves + wave;
            return (i == full && r < rem) ? full * stride + r : rows;
        };
        int64_t it = 0;
        row = row_at(0);
        issue(in_range(row), xA, gA);
        if ((p.debug_skip & 1024) && blockIdx.x < 1024) { if (lane == 0) g_dev_wave_times[(blockIdx.x * kWaves + wave) * 2] = wall_clock64(); }
        while (row < row_end) {
            const int64_t next = row_at(it + 1);
            compute(row, byte_offset(row), xA, gA, in_range(next));
            if (staged >= stage_rows) flush_stage();
            ++it; row = next;
        }
    } else if constexpr (DZO_PP_REFILL != 0) {
        // (Where the refill requests end up is the compiler's business: with the exit between the two halves it sinks
        // the first half's refills below the second compute(), next to the other set's.  Forcing them to stay where
        // they are written -- no exit in the middle, -mllvm -disable-machine-sink -- gave exact per-set waits and the
        // same kernel time, 640-648 against 623-644 us over two boxes: the sweep is bound by its memory shape, not by
        // the wave's waits.  SQ counters of that build: 44 % of the wave cycles issuing, 43 % issue stalls, 13 % in
        // s_waitcnt; the round-2 kernel: 69 / 10 / 21 % with 1.85 x the vector instructions.)
        issue(in_range(row), xA, gA);
        issue(in_range(row + stride), xB, gB);
        while (row < row_end) {
            compute(row, byte_offset(row), xA, gA, in_range(row + 2 * stride));
            if (staged >= stage_rows) flush_stage();
            row += stride;
            if (row >= row_end) break;
            compute(row, byte_offset(row), xB, gB, in_range(row + 2 * stride));
            if (staged >= stage_rows) flush_stage();
            row += stride;
        }
    } else {                                                     // whole-set requests one row ahead (the round-2 loop)
        issue(in_range(row), xA, gA);
        while (row < row_end) {
            issue(in_range(row + stride), xB, gB);
            compute(row, byte_offset(row), xA, gA, 0);
            if (staged >= stage_rows) flush_stage();
            row += stride;
            if (row >= row_end) break;
            issue(in_range(row + stride), xA, gA);
            compute(row, byte_offset(row), xB, gB, 0);
            if (staged >= stage_rows) flush_stage();
            row += stride;
        }
    }
    flush_stage();
    {
        // lane l holds the totals of values 64 g + bitrev6(l); value 5 j + c = dot c of (post-push) pair j
        double *wflat = &wacc[0][0][0];
        constexpr int kValues = kGramValues * (K + 1);
#pragma unroll
        for (int g = 0; g < TreeSum<kValues>::kGroups; ++g) {
            const int v = TreeSum<kValues>::value_of(g, lane);
            if (v < kValues) wflat[wave * (kGramValues * (kFusedMaxK + 1)) + v] = dots.acc[g];
        }
    }
    __syncthreads();
    for (int v = threadIdx.x; v < kGramValues * kn; v += kBlock) {
        const double *wflat = &wacc[0][0][0];
        constexpr int ws = kGramValues * (kFusedMaxK + 1);
        const double r = (wflat[v] + wflat[ws + v]) + (wflat[2 * ws + v] + wflat[3 * ws + v]);
        p.gram_partials[(int64_t)v * pstride + pcol] = r;
    }
    block_raise_flag(diff, p.changed, &lds_flag);
    const double fo = block_sum(fobj, lds);
    const double fh = block_sum(fobj_h, lds);
    if (threadIdx.x == 0) { p.obj_partials[pcol] = fo; p.obj_partials[pstride + pcol] = fh; }
    if constexpr (DEC) {
        const double so = block_sum(fsq, lds);
        const double sh = block_sum(fsq_h, lds);
        if (threadIdx.x == 0) { p.obj_partials[2 * pstride + pcol] = so; p.obj_partials[3 * pstride + pcol] = sh; }
    }
    if ((p.debug_skip & 1024) && blockIdx.x < 1024) { if (lane == 0) g_dev_wave_times[(blockIdx.x * kWaves + wave) * 2 + 1] = wall_clock64(); }
}

// ============================================================================ log-sum-exp on the point ring (round 4)
// f = log sum exp(x) + lambda/2 |x - c|^2 (BASELINE configs[3]), g = softmax(x) + lambda (x - c): elementwise GIVEN two
// global scalars of the point (max, sum exp).  The k + 1 stored points carry theirs (pscal, by slot); the trial point's
// exist only when a pass over it has ended, so the dots of the next two-loop cannot ride in the pass that forms the trial
// point.  A step therefore is TWO passes over the POINT ring:
//   lse_trial_kernel   k + 1 points + c -> gradients in registers, d (:438-449), the trial point (its tiles into the spare
//                      slot), and per block: max(x_new), sum exp(x_new - max_0), sum (x_new - c)^2, the change flag (:128)
//   lse_decide_kernel  one block: the trial point's scalars, f_new, the decision of :128 / :139
//   lse_dots_kernel    (accepted) k + 2 points + c -> the gradients again, pairs by subtraction, the 5 (k_next) dot products
//                      of the NEXT two-loop in the point pass's layout (gram_reduce_kernel / gram_finish_kernel take over)
// (2k + 5) n T of traffic against the pair path's (4k + 18) n T, for 2 (k + 1) exponentials per element.  Per element the
// arithmetic is lse_grad_kernel's (dzo_problems.hip); the sum of exponentials of an accepted point is carried RELATIVE to
// the previous point's maximum and rescaled on the scalars (softmax is shift-invariant; the two forms differ by roundings).
// No stencil: the halo positions of a tile are neither read nor written here.
constexpr int kLseMaxK = 24;
template <typename T> struct LseParams {
    int64_t n;
    int k, k_next;
    T t;
    T *ring;
    uint32_t rowbytes;
    uint32_t soff[kLseMaxK + 1];               // point j (0 = current) -> byte offset of its x tile within a row (beyond k: point k)
    uint32_t new_off, c_off;                   // the spare slot's x tile; the tiles of c
    uint8_t slot[kLseMaxK + 1];                // point j -> ring slot (for pscal)
    uint8_t new_slot;
    double *pscal;                             // [slots][2]: max, sum exp(x - max) of the point in each slot
    double lambda;
    const double *alpha, *coef, *scale;
    T *d;                                      // step_direction (contiguous, padded): read by the first step, written when store_d
    int store_d, store_tile;
    double *partials;                          // trial pass: [3][grid] max / sum exp / sum squares
    int32_t *changed;
    double *gram_partials;                     // dots pass: [kGramValues * k_next][grid]
    double f_cur;                              // decide
};

template <typename T> __device__ __forceinline__ T lse_grad_elem(T x, T c, T mxT, double se, double lambda) {
    const double sm = exp((double)(x - mxT)) / se;            // lse_grad_kernel's expression
    return (T)(sm + lambda * (double)(x - c));
}

// (The history length is a RUN-TIME loop count here: a point's tile is requested where the recurrence needs it -- the second
// loop re-reads the tiles the first one used, out of the cache, 1 KiB per wave and point -- so the kernel holds two tiles
// at a time instead of k + 1; with all of them in registers the inlined exponentials of k + 1 gradients pushed the fp64
// K = 24 and fp32 K = 20 instantiations past 512 registers.)
template <typename T, bool FIRST>
__global__ __launch_bounds__(kBlock, 3) void lse_trial_kernel(LseParams<T> p) {
    constexpr int N = Vec16<T>::N;
    __shared__ T a_s[kLseMaxK], c_s[kLseMaxK], mx_s[kLseMaxK + 1];
    __shared__ double se_s[kLseMaxK + 1];
    __shared__ uint32_t off_s[kLseMaxK + 1];
    __shared__ double lds[kWaves];
    __shared__ int lds_flag;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = p.k;
    if ((int)threadIdx.x < k) {
        a_s[threadIdx.x] = (T)(-p.alpha[threadIdx.x]);
        c_s[threadIdx.x] = (T)(-p.coef[threadIdx.x]);
    }
    if ((int)threadIdx.x <= k) {
        const int sl = p.slot[threadIdx.x];
        mx_s[threadIdx.x] = (T)p.pscal[2 * sl];
        se_s[threadIdx.x] = p.pscal[2 * sl + 1];
        off_s[threadIdx.x] = p.soff[threadIdx.x];
    }
    __syncthreads();
    const bool scaled = k > 0;
    const T scale = scaled ? (T)p.scale[0] : (T)1;
    const int64_t nvec = (p.n + N - 1) / N;
    const int64_t rows = (nvec + kRowOwn - 1) / kRowOwn;
    const uint32_t toff = (uint32_t)lane * 16u;
    double mx_loc = -1.7976931348623157e308, se_loc = 0, sq_loc = 0;
    bool diff = false;
    for (int64_t row = (int64_t)blockIdx.x * kWaves + wave; row < rows; row += (int64_t)gridDim.x * kWaves) {
        const int64_t v = row * kRowOwn - kRowLead + lane;
        const bool owner = v >= 0 && v < nvec && lane >= kRowLead && lane < kRowLead + kRowOwn;
        char *rb = reinterpret_cast<char *>(p.ring) + (uint64_t)row * p.rowbytes;
        auto tile = [&](int j, T (&dst)[N]) { load16(reinterpret_cast<const T *>(rb + off_s[j] + toff), dst); };
        T cv[N], x0v[N];
        load16_nt(reinterpret_cast<const T *>(rb + p.c_off + toff), cv);
        tile(0, x0v);
        const int64_t e0 = v * N;
        T q[N];
        if constexpr (FIRST) {
#pragma unroll
            for (int e = 0; e < N; ++e) q[e] = (owner && e0 + e < p.n) ? p.d[e0 + e] : (T)0;    // :463 the constructor's direction
        } else {
            T gprev[N];
#pragma unroll
            for (int e = 0; e < N; ++e) { gprev[e] = lse_grad_elem<T>(x0v[e], cv[e], mx_s[0], se_s[0], p.lambda); q[e] = gprev[e]; }   // :438
            T xb[N], xnext[N];
            if (k > 0) tile(1, xb);
            for (int i = 0; i < k; ++i) {                     // :439-442 (the next point's tile is requested before this one's exponentials)
                tile(i + 2 <= k ? i + 2 : k, xnext);
                const T a = a_s[i];
#pragma unroll
                for (int e = 0; e < N; ++e) {
                    const T gi1 = lse_grad_elem<T>(xb[e], cv[e], mx_s[i + 1], se_s[i + 1], p.lambda);
                    q[e] = dfma(a, gprev[e] - gi1, q[e]);
                    gprev[e] = gi1;
                }
#pragma unroll
                for (int e = 0; e < N; ++e) xb[e] = xnext[e];
            }
            if (scaled) {
#pragma unroll
                for (int e = 0; e < N; ++e) q[e] = scale * q[e];                              // :443-445
            }
            T xhi[N], xlo[N], xlo2[N];
            tile(k, xhi);
            if (k > 0) tile(k - 1, xlo);
            for (int i = k - 1; i >= 0; --i) {                // :446-449 (two tiles ahead of the fma)
                tile(i > 0 ? i - 1 : 0, xlo2);
                const T c = c_s[i];
#pragma unroll
                for (int e = 0; e < N; ++e) { q[e] = dfma(c, xlo[e] - xhi[e], q[e]); xhi[e] = xlo[e]; xlo[e] = xlo2[e]; }
            }
        }
        T xn[N];
#pragma unroll
        for (int e = 0; e < N; ++e) {
            const bool real = owner && e0 + e < p.n;
            xn[e] = real ? dfma(p.t, q[e], x0v[e]) : (T)0;                                    // :124 (the padding of a ragged n stays +0)
            if (real) {
                diff |= !is_equal(xn[e], x0v[e]);                                             // :128
                mx_loc = fmax(mx_loc, (double)xn[e]);
                se_loc += exp((double)(xn[e] - mx_s[0]));                                     // relative to the CURRENT point's maximum
                const T dlt = xn[e] - cv[e];
                sq_loc = __builtin_fma((double)dlt, (double)dlt, sq_loc);
            }
        }
        if (owner) {
            if (p.store_tile) store16(reinterpret_cast<T *>(rb + p.new_off + toff), xn);
            if (p.store_d && !FIRST) {
#pragma unroll
                for (int e = 0; e < N; ++e) if (e0 + e < p.n) p.d[e0 + e] = q[e];
            }
        }
    }
    // per block: max, sum exp, sum of squares
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx_loc = fmax(mx_loc, __shfl_xor(mx_loc, off, 64));
    if (lane == 0) lds[wave] = mx_loc;
    __syncthreads();
    double bmx = lds[0];
    for (int w = 1; w < kWaves; ++w) bmx = fmax(bmx, lds[w]);
    __syncthreads();
    const double bse = block_sum(se_loc, lds);
    const double bsq = block_sum(sq_loc, lds);
    if (threadIdx.x == 0) { p.partials[blockIdx.x] = bmx; p.partials[gridDim.x + blockIdx.x] = bse; p.partials[2 * gridDim.x + blockIdx.x] = bsq; }
    block_raise_flag(diff, p.changed, &lds_flag);
}

// the trial point's scalars, f_new = max + log(sum exp) + lambda/2 |x - c|^2 (lse_finish_kernel's expression) and the decision
__global__ __launch_bounds__(kBlock) void lse_decide_kernel(const double *__restrict__ partials, int grid, double *__restrict__ pscal,
                                                            int cur_slot, int new_slot, double lambda, int cur_is_f32, DecideArgs dec) {
    __shared__ double lds[kWaves];
    double mx = -1.7976931348623157e308;
    for (int i = threadIdx.x; i < grid; i += kBlock) mx = fmax(mx, partials[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = mx;
    __syncthreads();
    double mx_new = lds[0];
    for (int w = 1; w < kWaves; ++w) mx_new = fmax(mx_new, lds[w]);
    __syncthreads();
    const double se_rel = reduce_partials_all(partials + grid, grid, lds);
    const double sq = reduce_partials_all(partials + 2 * grid, grid, lds);
    // (the exponentials were taken relative to the current point's maximum as the kernels see it, i.e. rounded to T)
    const double mx_cur = cur_is_f32 ? (double)(float)pscal[2 * cur_slot] : pscal[2 * cur_slot];
    const double se_new = se_rel * exp(mx_cur - mx_new);
    if (threadIdx.x == 0) {
        pscal[2 * new_slot] = mx_new; pscal[2 * new_slot + 1] = se_new;
        // no decrease can be claimed from a sum that left the range of exp (a trial point hundreds of units away): NaN fails :139
        const bool usable = se_rel > 0.0 && se_new > 0.0 && se_new < 1.7976931348623157e308;
        dec.result[0] = usable ? mx_new + log(se_new) + 0.5 * lambda * sq : __builtin_nan("");
    }
    __syncthreads();
    decide_body(dec, lds);
}

// the dots of the NEXT two-loop (post-push order: new pair = 0, old pair i = i + 1), point pass layout of the partials
template <typename T, int K>
__global__ __launch_bounds__(kBlock, (K <= 20 ? 2 : 1)) void lse_dots_kernel(LseParams<T> p) {
    constexpr int N = Vec16<T>::N;
    __shared__ T mx_s[kLseMaxK + 2];
    __shared__ double se_s[kLseMaxK + 2];
    __shared__ double wacc[kWaves][kFusedMaxK + 1][kGramValues];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kn = p.k_next;
    if (threadIdx.x <= K) {
        const int sl = p.slot[threadIdx.x];
        mx_s[threadIdx.x + 1] = (T)p.pscal[2 * sl];
        se_s[threadIdx.x + 1] = p.pscal[2 * sl + 1];
    }
    if (threadIdx.x == 0) { mx_s[0] = (T)p.pscal[2 * p.new_slot]; se_s[0] = p.pscal[2 * p.new_slot + 1]; }
    __syncthreads();
    const int64_t nvec = (p.n + N - 1) / N;
    const int64_t rows = (nvec + kRowOwn - 1) / kRowOwn;
    const uint32_t toff = (uint32_t)lane * 16u;
    TreeSum<kGramValues * (K + 1)> dots;
    dots.init();
    for (int64_t row = (int64_t)blockIdx.x * kWaves + wave; row < rows; row += (int64_t)gridDim.x * kWaves) {
        const int64_t v = row * kRowOwn - kRowLead + lane;
        const bool owner = v >= 0 && v < nvec && lane >= kRowLead && lane < kRowLead + kRowOwn;
        char *rb = reinterpret_cast<char *>(p.ring) + (uint64_t)row * p.rowbytes;
        const int64_t e0 = v * N;
        T cv[N], xn[N], xa[N], ga[N];
        load16_nt(reinterpret_cast<const T *>(rb + p.c_off + toff), cv);
        load16(reinterpret_cast<const T *>(rb + p.new_off + toff), xn);
        load16_nt(reinterpret_cast<const T *>(rb + p.soff[0] + toff), xa);
        T gn[N], sn[N], yn[N];
#pragma unroll
        for (int e = 0; e < N; ++e) {
            const bool real = owner && e0 + e < p.n;
            gn[e] = real ? lse_grad_elem<T>(xn[e], cv[e], mx_s[0], se_s[0], p.lambda) : (T)0;
            ga[e] = real ? lse_grad_elem<T>(xa[e], cv[e], mx_s[1], se_s[1], p.lambda) : (T)0;
            sn[e] = real ? xn[e] - xa[e] : (T)0;                                              // :145
            yn[e] = real ? gn[e] - ga[e] : (T)0;                                              // :478-480
        }
        {
            double t5[kGramValues] = {0, 0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < N; ++e) {
                const double sx = (double)sn[e], yx = (double)yn[e], gx = (double)gn[e];
                t5[0] = __builtin_fma(sx, gx, t5[0]); t5[1] = __builtin_fma(yx, gx, t5[1]); t5[2] = __builtin_fma(yx, yx, t5[2]);
                t5[3] = __builtin_fma(yx, sx, t5[3]); t5[4] = __builtin_fma(sx, yx, t5[4]);
            }
            dots.template push<0>(t5[0], lane); dots.template push<1>(t5[1], lane); dots.template push<2>(t5[2], lane);
            dots.template push<3>(t5[3], lane); dots.template push<4>(t5[4], lane);
        }
        T xb[N], xpre[N];
        load16_nt(reinterpret_cast<const T *>(rb + p.soff[1] + toff), xb);
        auto pair_dots = [&](auto ic) {                       // old pair i = point i - point i + 1 (xa / ga hold point i, xb point i + 1)
            constexpr int i = decltype(ic)::value;
            T gb[N];
            double t5[kGramValues] = {0, 0, 0, 0, 0};
            if (i < p.k) {                                    // (uniform; a pair beyond the history is zero: no tile, no exponentials, zeros into the sums)
            load16_nt(reinterpret_cast<const T *>(rb + p.soff[i + 2 <= K ? i + 2 : K] + toff), xpre);   // (one tile ahead of the exponentials)
#pragma unroll
            for (int e = 0; e < N; ++e) {
                const bool real = owner && e0 + e < p.n;
                gb[e] = real ? lse_grad_elem<T>(xb[e], cv[e], mx_s[i + 2], se_s[i + 2], p.lambda) : (T)0;
                const double sx = real ? (double)(xa[e] - xb[e]) : 0.0, yx = real ? (double)(ga[e] - gb[e]) : 0.0;
                t5[0] = __builtin_fma(sx, (double)gn[e], t5[0]); t5[1] = __builtin_fma(yx, (double)gn[e], t5[1]);
                t5[2] = __builtin_fma(yx, (double)yn[e], t5[2]); t5[3] = __builtin_fma(yx, (double)sn[e], t5[3]);
                t5[4] = __builtin_fma(sx, (double)yn[e], t5[4]);
            }
#pragma unroll
            for (int e = 0; e < N; ++e) { xa[e] = xb[e]; ga[e] = gb[e]; xb[e] = xpre[e]; }
            }
            constexpr int v0 = kGramValues * (i + 1);
            dots.template push<v0 + 0>(t5[0], lane); dots.template push<v0 + 1>(t5[1], lane); dots.template push<v0 + 2>(t5[2], lane);
            dots.template push<v0 + 3>(t5[3], lane); dots.template push<v0 + 4>(t5[4], lane);
        };
        static_for<K>(pair_dots);
        dots.finish_row(lane);
    }
    {
        double *wflat = &wacc[0][0][0];
        constexpr int kValues = kGramValues * (K + 1);
#pragma unroll
        for (int g = 0; g < TreeSum<kValues>::kGroups; ++g) {
            const int v = TreeSum<kValues>::value_of(g, lane);
            if (v < kValues) wflat[wave * (kGramValues * (kFusedMaxK + 1)) + v] = dots.acc[g];
        }
    }
    __syncthreads();
    for (int v = threadIdx.x; v < kGramValues * kn; v += kBlock) {
        const double *wflat = &wacc[0][0][0];
        constexpr int ws = kGramValues * (kFusedMaxK + 1);
        p.gram_partials[(int64_t)v * gridDim.x + blockIdx.x] = (wflat[v] + wflat[ws + v]) + (wflat[2 * ws + v] + wflat[3 * ws + v]);
    }
}

// Ragged n on the tile ring: a vector length that is not a multiple of the 16-byte vector gets a last vector padded with
// PHANTOM elements.  They are +0 in every point of the ring and stay +0: their stencil coefficients are all zero
// (rosen_coef), so their gradient is +0, their direction +-0 and their trial point fma(t, +-0, +0) = +0; they add exact
// zeros to every sum and never differ from the old point.  The contiguous arrays of the caller have n elements and no
// padding: whoever moves a vector between them and the ring touches the last vector element by element.
template <typename T> __device__ __forceinline__ void load_vec_tail(const T *lin, int64_t v, int64_t n, T (&t)[Vec16<T>::N]) {
    constexpr int N = Vec16<T>::N;
    if (v * N + N <= n) { load16(lin + v * N, t); return; }
#pragma unroll
    for (int j = 0; j < N; ++j) t[j] = v * N + j < n ? lin[v * N + j] : (T)0;
}
template <typename T> __device__ __forceinline__ void store_vec_tail(T *lin, int64_t v, int64_t n, const T (&t)[Vec16<T>::N]) {
    constexpr int N = Vec16<T>::N;
    if (v * N + N <= n) { store16(lin + v * N, t); return; }
#pragma unroll
    for (int j = 0; j < N; ++j) if (v * N + j < n) lin[v * N + j] = t[j];
}

// ring streams elementwise, tile positions included (the halo copies transform like their originals):
// a <- a - b (point ring -> pair ring, in place)
template <typename T>
__global__ __launch_bounds__(kBlock) void ring_diff_kernel(int64_t rows, T *__restrict__ a, const T *__restrict__ b, int64_t rowbytes) {
    constexpr int N = Vec16<T>::N;
    for (int64_t id = (int64_t)blockIdx.x * kBlock + threadIdx.x; id < rows * 64; id += (int64_t)gridDim.x * kBlock) {
        const int64_t row = id >> 6;
        const int pos = (int)(id & 63);
        T *pa = reinterpret_cast<T *>(reinterpret_cast<char *>(a) + row * rowbytes + pos * 16);
        const T *pb = reinterpret_cast<const T *>(reinterpret_cast<const char *>(b) + row * rowbytes + pos * 16);
        T va[N], vb[N];
        load16(pa, va); load16(pb, vb);
#pragma unroll
        for (int j = 0; j < N; ++j) va[j] = va[j] - vb[j];
        store16(pa, va);
    }
}

// the gradient tiles of a point from its point tiles (point ring; the passes do not write them): every tile position,
// halo copies included, evaluates rosen_grad_elem on its vector -- the bits the pass had in registers
template <typename T>
__global__ __launch_bounds__(kBlock) void ring_regrad_kernel(int64_t n, int64_t nvec, const T *__restrict__ xs, T *__restrict__ gs, int64_t rowbytes,
                                                             PointDecor<T> dec, int dec_on, int obj, T obj_lambda,
                                                             const T *__restrict__ cs = nullptr, const double *__restrict__ pscal2 = nullptr, double lse_lambda = 0) {
    constexpr int N = Vec16<T>::N;
    const int64_t rows = (nvec + kRowOwn - 1) / kRowOwn;
    for (int64_t id = (int64_t)blockIdx.x * kBlock + threadIdx.x; id < rows * 64; id += (int64_t)gridDim.x * kBlock) {
        const int64_t row = id >> 6;
        const int pos = (int)(id & 63);
        const int64_t v = row * kRowOwn - kRowLead + pos;
        if (v < 0 || v >= nvec) continue;
        T x[N], g[N];
        load16(reinterpret_cast<const T *>(reinterpret_cast<const char *>(xs) + row * rowbytes + pos * 16), x);
        if (obj == 2) {                                   // log-sum-exp: elementwise given the point's two scalars (lse_grad_elem)
            T cvv[N];
            load16(reinterpret_cast<const T *>(reinterpret_cast<const char *>(cs) + row * rowbytes + pos * 16), cvv);
            const T mxT = (T)pscal2[0];
            const double se = pscal2[1];
#pragma unroll
            for (int e = 0; e < N; ++e) g[e] = v * N + e < n ? lse_grad_elem<T>(x[e], cvv[e], mxT, se, lse_lambda) : (T)0;
            store16(reinterpret_cast<T *>(reinterpret_cast<char *>(gs) + row * rowbytes + pos * 16), g);
            continue;
        }
        const T xl = v > 0 ? hist_ptr<true>(xs, v - 1, rowbytes)[N - 1] : (T)0;
        const T xr = v + 1 < nvec ? hist_ptr<true>(xs, v + 1, rowbytes)[0] : (T)0;
#pragma unroll
        for (int e = 0; e < N; ++e) {           // (an element past n -- the padding of a ragged last vector -- has gradient +0)
            const T xp = e > 0 ? x[(e + N - 1) % N] : xl, xq = e + 1 < N ? x[(e + 1) % N] : xr;
            if (obj == 1) g[e] = qchain_grad_coef<T>(qchain_coef<T>(v * N + e, n, obj_lambda), xp, x[e], xq);
            else g[e] = v * N + e < n ? rosen_grad_elem<T>(v * N + e, n, xp, x[e], xq) : (T)0;
        }
        if (dec_on) {                                       // (the decorated gradient, as the DEC pass forms it)
#pragma unroll
            for (int e = 0; e < N; ++e) g[e] = decor_grad<T>(dec, g[e], x[e]);
        }
        store16(reinterpret_cast<T *>(reinterpret_cast<char *>(gs) + row * rowbytes + pos * 16), g);
    }
}

// contiguous vector <- stream a - stream b (delta_point / a pair of the point ring as a plain vector)
template <typename T>
__global__ __launch_bounds__(kBlock) void ring_gather_diff_kernel(int64_t n, int64_t nvec, const T *__restrict__ a, const T *__restrict__ b,
                                                                  T *__restrict__ lin, int64_t rowbytes) {
    constexpr int N = Vec16<T>::N;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * kBlock) {
        const int64_t row = v / kRowOwn;
        const int pos = (int)(v - row * kRowOwn) + kRowLead;
        T va[N], vb[N];
        load16(reinterpret_cast<const T *>(reinterpret_cast<const char *>(a) + row * rowbytes + pos * 16), va);
        load16(reinterpret_cast<const T *>(reinterpret_cast<const char *>(b) + row * rowbytes + pos * 16), vb);
#pragma unroll
        for (int j = 0; j < N; ++j) va[j] = va[j] - vb[j];
        store_vec_tail(lin, v, n, va);
    }
}

// ---------------------------------------------------------------------------- blocked ring <-> contiguous vectors
// One stream of the ring (all its tiles) from / to a contiguous vector of nvec 16-B vectors.  Thread = one
// tile position; the halo positions 0 and 63 take the neighbouring rows' edge vectors.
template <typename T>
__global__ __launch_bounds__(kBlock) void ring_scatter_kernel(int64_t n, int64_t nvec, const T *__restrict__ lin, T *__restrict__ stream, int64_t rowbytes) {
    constexpr int N = Vec16<T>::N;
    const int64_t rows = (nvec + kRowOwn - 1) / kRowOwn;
    for (int64_t id = (int64_t)blockIdx.x * kBlock + threadIdx.x; id < rows * 64; id += (int64_t)gridDim.x * kBlock) {
        const int64_t row = id >> 6;
        const int pos = (int)(id & 63);
        const int64_t v = row * kRowOwn - kRowLead + pos;
        if (v < 0 || v >= nvec) continue;
        T t[N];
        load_vec_tail(lin, v, n, t);
        store16(reinterpret_cast<T *>(reinterpret_cast<char *>(stream) + row * rowbytes + pos * 16), t);
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void ring_gather_kernel(int64_t n, int64_t nvec, const T *__restrict__ stream, T *__restrict__ lin, int64_t rowbytes) {
    constexpr int N = Vec16<T>::N;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * kBlock) {
        T t[N];
        load16(hist_ptr<true>(stream, v, rowbytes), t);
        store_vec_tail(lin, v, n, t);
    }
}

// does the contiguous vector still equal the stream it was gathered from?  (bitwise, NaN == NaN; one plain store
// per block that saw a difference)
template <typename T>
__global__ __launch_bounds__(kBlock) void ring_compare_kernel(int64_t n, int64_t nvec, const T *__restrict__ stream, const T *__restrict__ lin, int64_t rowbytes,
                                                              int32_t *__restrict__ differs) {
    constexpr int N = Vec16<T>::N;
    __shared__ int lds_flag;
    bool diff = false;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * kBlock) {
        T a[N], b[N];
        load16(hist_ptr<true>(stream, v, rowbytes), a);
        load_vec_tail(lin, v, n, b);                      // (the padding of a ragged last vector is +0 on both sides)
#pragma unroll
        for (int j = 0; j < N; ++j) diff |= !is_equal(a[j], b[j]);
    }
    block_raise_flag(diff, differs, &lds_flag);
}

// ============================================================================ post-gradient
// :480 delta_gradient = g - delta_gradient, fused with the partials of
// rho = dot(delta_point, delta_gradient) (:505).
template <typename T, bool VEC>
__global__ __launch_bounds__(kBlock) void delta_rho_kernel(int64_t n, const T *__restrict__ g, T *__restrict__ dg,
                                                           const T *__restrict__ dx, double *__restrict__ partials) {
    using L = Ld<T, VEC>;
    constexpr int N = L::N;
    __shared__ double lds[kWaves];
    double acc = 0;
    const int64_t nvec = n / N;
    const int64_t nthreads = (int64_t)gridDim.x * kBlock;
    for (int64_t base = (int64_t)blockIdx.x * kBlock * 2; base < nvec; base += nthreads * 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t vi = base + (int64_t)u * kBlock + threadIdx.x;
            if (vi >= nvec) continue;
            T gv[N], ov[N], xv[N];
            L::load(g + vi * N, gv);
            L::load(dg + vi * N, ov);
            L::load(dx + vi * N, xv);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                ov[j] = gv[j] - ov[j];
                acc = __builtin_fma((double)xv[j], (double)ov[j], acc);
            }
            L::store(dg + vi * N, ov);
        }
    }
    if constexpr (VEC) {
        const int64_t i = nvec * N + (int64_t)blockIdx.x * kBlock + threadIdx.x;
        if (i < n) {
            const T o = g[i] - dg[i];
            dg[i] = o;
            acc = __builtin_fma((double)dx[i], (double)o, acc);
        }
    }
    double r = block_sum(acc, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = r;
}

// The tail of an accepted step on the GENERAL path (callbacks, any objective) in one pass: after the gradient callback has
// written g_new into the optimizer's other gradient buffer,
//     delta_point    = x - x_old            (:145; x_old is what the first trial saved in delta_point, :118)
//     delta_gradient = g_new - g_old        (:478-480 without the copy of :478: g_old is simply the buffer that was current)
//     partials of rho = delta_point . delta_gradient   (:505)
// 4 reads + 2 writes per element instead of axpby (3) + copy (2) + delta_rho_kernel (4); same elementwise operations, so
// the same bits (x - x_old and g_new - g_old are single roundings either way).
template <typename T, bool VEC>
__global__ __launch_bounds__(kBlock) void accept_delta_rho_kernel(int64_t n, const T *__restrict__ x, T *__restrict__ dx,
                                                                  const T *__restrict__ g_new, const T *__restrict__ g_old,
                                                                  T *__restrict__ dg, double *__restrict__ partials) {
    using L = Ld<T, VEC>;
    constexpr int N = L::N;
    __shared__ double lds[kWaves];
    double acc = 0;
    const int64_t nvec = n / N;
    const int64_t nthreads = (int64_t)gridDim.x * kBlock;
    for (int64_t base = (int64_t)blockIdx.x * kBlock * 2; base < nvec; base += nthreads * 2) {
        T xv[2][N], ov[2][N], gn[2][N], go[2][N];
#pragma unroll
        for (int u = 0; u < 2; ++u) {                      // all loads of the two vectors first
            const int64_t vi = base + (int64_t)u * kBlock + threadIdx.x;
            if (vi >= nvec) continue;
            L::load(x + vi * N, xv[u]); L::load(dx + vi * N, ov[u]); L::load(g_new + vi * N, gn[u]); L::load(g_old + vi * N, go[u]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t vi = base + (int64_t)u * kBlock + threadIdx.x;
            if (vi >= nvec) continue;
            T sv[N], yv[N];
#pragma unroll
            for (int j = 0; j < N; ++j) {
                sv[j] = xv[u][j] - ov[u][j];
                yv[j] = gn[u][j] - go[u][j];
                acc = __builtin_fma((double)sv[j], (double)yv[j], acc);
            }
            L::store(dx + vi * N, sv);
            L::store(dg + vi * N, yv);
        }
    }
    if constexpr (VEC) {
        const int64_t i = nvec * N + (int64_t)blockIdx.x * kBlock + threadIdx.x;
        if (i < n) {
            const T sv = x[i] - dx[i], yv = g_new[i] - g_old[i];
            dx[i] = sv; dg[i] = yv;
            acc = __builtin_fma((double)sv, (double)yv, acc);
        }
    }
    const double r = block_sum(acc, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = r;
}

__global__ __launch_bounds__(kBlock) void finish_to_kernel(const double *__restrict__ partials, int count,
                                                           double *__restrict__ dst, int to_f32,
                                                           const int32_t *__restrict__ gate) {
    __shared__ double lds[kWaves];
    if (gate && *gate != 1) return;
    double r = reduce_partials_all(partials, count, lds);
    if (threadIdx.x == 0) dst[0] = to_f32 ? (double)(float)r : r;
}

static inline bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static SlotMap make_map(const dzo_lbfgs_s *o) {
    SlotMap mp;
    memset(&mp, 0, sizeof(mp));
    for (int i = 0; i < o->k; ++i) mp.slot[i] = (uint8_t)o->slot_of(i);
    return mp;
}

// ---------------------------------------------------------------------------- CHAIN driver
template <typename T> static int32_t direction_chain(dzo_lbfgs_s *o) {
    OptCore &c = o->core;
    hipStream_t s = c.stream;
    const int64_t n = c.n;
    const int k = o->k;
    const T *g = (const T *)c.g;
    T *d = (T *)o->d;
    const bool vec = al16(g);  // every other operand is library-allocated and aligned
    const int grid = stream_grid(n, (vec ? Vec16<T>::N : 1) * 2);
    double *bufA = o->link_partials, *bufB = o->link_partials + kMaxPartialBlocks;
    double *yyP = o->link_partials + 2 * kMaxPartialBlocks;
    {
        DZO_TIMED("lbfgs_chain_head", s);
        const T *s0 = o->s_slot<T>(o->slot_of(0));
        const T *y0 = o->y_slot<T>(o->slot_of(0));
        if (vec) hipLaunchKernelGGL((chain_head_kernel<T, true>), dim3(grid), dim3(kBlock), 0, s, n, s0, g, y0, bufA, k == 1 ? yyP : (double *)nullptr);
        else hipLaunchKernelGGL((chain_head_kernel<T, false>), dim3(grid), dim3(kBlock), 0, s, n, s0, g, y0, bufA, k == 1 ? yyP : (double *)nullptr);
    }
    auto launch = [&](const LinkParams &lp) {
        DZO_TIMED("lbfgs_chain_link", s);
        if (vec) hipLaunchKernelGGL((chain_link_kernel<T, true>), dim3(grid), dim3(kBlock), 0, s, lp);
        else hipLaunchKernelGGL((chain_link_kernel<T, false>), dim3(grid), dim3(kBlock), 0, s, lp);
    };
    double *prev = bufA, *next = bufB;
    for (int i = 0; i < k; ++i) {                         // :439-442
        LinkParams lp;
        memset(&lp, 0, sizeof(lp));
        lp.n = n;
        lp.in = (i == 0) ? (const void *)g : (const void *)d;
        lp.out = d;
        lp.v = o->y_slot<T>(o->slot_of(i));
        lp.w = (i < k - 1) ? (const void *)o->s_slot<T>(o->slot_of(i + 1)) : (const void *)o->y_slot<T>(o->slot_of(k - 1));
        lp.prev = prev; lp.prev_count = grid;
        lp.rho = o->rho + o->slot_of(i);
        lp.alpha = o->alpha + i;
        lp.second_loop = 0;
        if (i == k - 1) { lp.yy = yyP; lp.yy_count = grid; lp.rho0 = o->rho + o->slot_of(0); lp.scale_out = o->scale; }  // :443-445
        lp.dot_out = next;
        lp.yy_out = (i == 0 && k > 1) ? yyP : nullptr;
        launch(lp);
        std::swap(prev, next);
    }
    for (int i = k - 1; i >= 0; --i) {                    // :446-449
        LinkParams lp;
        memset(&lp, 0, sizeof(lp));
        lp.n = n;
        lp.in = d; lp.out = d;
        lp.v = o->s_slot<T>(o->slot_of(i));
        lp.w = (i > 0) ? (const void *)o->y_slot<T>(o->slot_of(i - 1)) : nullptr;
        lp.prev = prev; lp.prev_count = grid;
        lp.rho = o->rho + o->slot_of(i);
        lp.alpha = o->alpha + i;
        lp.coef_out = o->coef + i;
        lp.second_loop = 1;
        lp.dot_out = (i > 0) ? next : nullptr;
        launch(lp);
        std::swap(prev, next);
    }
    DZO_HIP(hipGetLastError());
    return DZO_OK;
}

// ---------------------------------------------------------------------------- GRAM driver
// blocks of `kernel` (kBlock threads, static LDS only) that fit on one CU at once; cached
static int resident_blocks(const void *kernel, size_t dyn_lds = 0) {
    static std::map<std::pair<const void *, size_t>, int> cache;
    static std::mutex mu;                      // handles may be driven from different host threads
    std::lock_guard<std::mutex> lk(mu);
    const auto key = std::make_pair(kernel, dyn_lds);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, kBlock, dyn_lds) != hipSuccess || nb < 1) nb = dyn_lds ? 1 : 4;
    cache[key] = nb;
    return nb;
}

static int tune(const char *name, int dflt) {
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}

static GramFinishParams gram_finish_params(dzo_lbfgs_s *o, int pivot, bool recurrence, const double *vals, bool rho_from_vals,
                                           const int32_t *gate) {
    GramFinishParams fp;
    fp.gate = gate;
    fp.k = o->k; fp.m1 = o->nslots; fp.pivot = pivot; fp.grid = 0; fp.do_recurrence = recurrence ? 1 : 0;
    fp.map = make_map(o); fp.partials = vals; fp.rho = o->rho;
    fp.rho_from_vals = rho_from_vals ? 1 : 0; fp.rho_to_f32 = o->core.dtype == DZO_F32 ? 1 : 0;
    fp.Gyy = o->Gyy; fp.Gsy = o->Gsy; fp.sg = o->sg; fp.yg = o->yg;
    fp.alpha = o->alpha; fp.coef = o->coef; fp.scale = o->scale;
    if (gate) { fp.alpha = o->alpha_sp; fp.coef = o->coef_sp; fp.scale = o->scale_sp; }   // next step's set
    fp.fast_div = tune("DZO_TUNE_FAST_DIV", 1) != 0 ? 1 : 0;
    return fp;
}

static int32_t gram_finish_launch(dzo_lbfgs_s *o, int pivot, bool recurrence, const double *vals, bool rho_from_vals = false,
                                  const int32_t *gate = nullptr) {
    hipStream_t s = o->core.stream;
    const GramFinishParams fp = gram_finish_params(o, pivot, recurrence, vals, rho_from_vals, gate);
    {
        DZO_TIMED("lbfgs_gram_finish", s);
        hipLaunchKernelGGL(gram_finish_kernel, dim3(1), dim3(kBlock), gram_finish_lds_bytes(o->k), s, fp);
    }
    DZO_HIP(hipGetLastError());
    return DZO_OK;
}

template <typename T> static int32_t gram_pass(dzo_lbfgs_s *o, int pivot, bool recurrence) {
    OptCore &c = o->core;
    hipStream_t s = c.stream;
    const int k = o->k;
    GramParams<T> gp;
    memset(&gp, 0, sizeof(gp));
    gp.n = c.n; gp.g = (const T *)c.g; gp.k = k;
    gp.sp = o->s_slot<T>(o->slot_of(pivot)); gp.yp = o->y_slot<T>(o->slot_of(pivot));
    for (int i = 0; i < k; ++i) { gp.s[i] = o->s_slot<T>(o->slot_of(i)); gp.y[i] = o->y_slot<T>(o->slot_of(i)); }
    gp.partials = o->gram_partials;
    gp.rowbytes = o->rowbytes;
    gp.debug = tune("DZO_TUNE_TP_DEBUG", 0) == 1;
    gp.peel = o->gram_peel;
    gp.fresh_plain = o->gram_fresh_plain;
    gp.pivot_first = (o->gram_skip0 && pivot == 0) ? 1 : 0;
    const bool post = o->post_pending;                    // (lbfgs_direction made sure: pivot 0, slabs, aligned operands, pivot served from registers)
    if (post) {
        gp.x = (const T *)c.x; gp.g_old = (const T *)o->post_g_old;
        gp.sp_out = o->s_slot<T>(o->slot_of(0)); gp.yp_out = o->y_slot<T>(o->slot_of(0));
    }
    const bool vec = al16(c.g);
    if (o->gram_variant == 1) {
        // lane-distributed accumulators: one launch shape for every k (the pair-per-wave
        // variant, 5*ceil(k/4) accumulators per lane and 8 concurrent streams per block, measured
        // 5 % slower at k = 20 and was removed)
        // vectors per thread: as many as still leave >= 4 blocks per CU (small n would otherwise
        // launch fewer blocks than there are CUs)
        int gu = o->gram_u;
        const int64_t vec_elems = (int64_t)kBlock * (vec ? Vec16<T>::N : 1);
        while (gu > 1 && (c.n + vec_elems * gu - 1) / (vec_elems * gu) < (int64_t)ctx().cus * 4) gu >>= 1;
        int64_t per_block = vec_elems * gu;
        int64_t blocks = (c.n + per_block - 1) / per_block;
        // grid = the blocks that are resident at once (occupancy x CUs): every block then walks
        // the same number of tiles (+-1) and there is no second, partially filled round of
        // blocks.  Measured at n = 1e7, k = 20: 494 us against 527 us with 8 blocks per CU.
        double *vals = o->gram_partials + (size_t)kGramValues * kMaxHistory * o->gram_grid * kWaves;
#define GL(UU)                                                                                                  \
    do {                                                                                                        \
        auto kern = o->blocked ? gram_pass_lanes_kernel<T, true, UU, true>                                      \
                    : post ? gram_pass_lanes_kernel<T, true, UU, false, true>                                   \
                    : vec ? gram_pass_lanes_kernel<T, true, UU> : gram_pass_lanes_kernel<T, false, UU>;         \
        int64_t cap = o->gram_grid;                                                                             \
        if (o->gram_bpc <= 0) { const int64_t res = (int64_t)ctx().cus * resident_blocks((const void *)kern); if (res < cap) cap = res; } \
        if (blocks > cap) blocks = cap;                                                                         \
        lgrid = (int)(blocks < 1 ? 1 : blocks);                                                                 \
        DZO_TIMED("lbfgs_gram_pass", s);                                                                        \
        hipLaunchKernelGGL(kern, dim3(lgrid), dim3(kBlock), 0, s, gp);                                          \
    } while (0)
        int lgrid = 1;
        if (gu == 1) GL(1); else if (gu == 2) GL(2); else GL(4);      // (8 vectors per lane spills: not offered)
#undef GL
        const int pcount = lgrid;
        const int nvals = kGramValues * k;
        const bool rho = o->rho_pending;
        if (post) {                                       // rho of the new pair is value 4 of pair 0 (s_p . y_p): gram_finish takes it from there
            o->post_pending = false;
            {
                DZO_TIMED("lbfgs_gram_reduce", s);
                hipLaunchKernelGGL(gram_reduce_kernel, dim3(nvals), dim3(kBlock), 0, s, o->gram_partials, pcount, vals, nvals, (const double *)nullptr, 0,
                                   (double *)nullptr, 0);
            }
            return gram_finish_launch(o, pivot, recurrence, vals, true);
        }
        if (o->fused_finish && (int64_t)nvals * pcount <= o->fused_finish_max) {
            // small problem: the scalar stage in one launch (see gram_reduce_finish_kernel)
            const GramFinishParams fp = gram_finish_params(o, pivot, recurrence, vals, false, nullptr);
            {
                DZO_TIMED("lbfgs_gram_reduce_finish", s);
                hipLaunchKernelGGL(gram_reduce_finish_kernel, dim3(1), dim3(kFusedFinishThreads), gram_finish_lds_bytes(k), s, fp,
                                   (const double *)o->gram_partials, pcount, vals, nvals, (const double *)c.partials(),
                                   rho ? o->rho_pending_count : 0, o->rho + o->rho_pending_slot, c.dtype == DZO_F32 ? 1 : 0);
            }
            o->rho_pending = false;
            DZO_HIP(hipGetLastError());
            return DZO_OK;
        }
        {
            DZO_TIMED("lbfgs_gram_reduce", s);
            hipLaunchKernelGGL(gram_reduce_kernel, dim3(nvals + (rho ? 1 : 0)), dim3(kBlock), 0, s, o->gram_partials, pcount,
                               vals, nvals, (const double *)c.partials(), o->rho_pending_count, o->rho + o->rho_pending_slot,
                               c.dtype == DZO_F32 ? 1 : 0);
            o->rho_pending = false;
        }
        return gram_finish_launch(o, pivot, recurrence, vals);
    }
    set_error("unknown Gram variant");
    return DZO_ERR_INVALID;
}

// alpha / coef / scale of the current history and gradient: from the dots the last single-pass step
// left behind (reduce + finish only), or from Gram passes over the history
template <typename T> static int32_t gram_scalars(dzo_lbfgs_s *o) {
    OptCore &c = o->core;
    const int k = o->k;
    if (o->spec_scalars) {                                // computed behind the previous step's decision
        std::swap(o->alpha, o->alpha_sp); std::swap(o->coef, o->coef_sp); std::swap(o->scale, o->scale_sp);
        o->spec_scalars = false;
        o->gram_ready = false; o->gram_rebuild = false; o->gram_stale = 0;
        return DZO_OK;
    }
    if (o->scalars_ready) return DZO_OK;
    if (o->gram_ready) {
        double *vals = o->gram_partials + (size_t)kGramValues * kMaxHistory * o->gram_grid * kWaves;
        {
            DZO_TIMED("lbfgs_gram_reduce", c.stream);
            hipLaunchKernelGGL(gram_reduce_kernel, dim3(kGramValues * k), dim3(kBlock), 0, c.stream, o->gram_partials,
                               o->gram_ready_grid, vals, kGramValues * k, (const double *)nullptr, 0, (double *)nullptr, 0);
        }
        DZO_TRY(gram_finish_launch(o, 0, true, vals, true));
        o->gram_ready = false;
    } else {
        if (o->gram_rebuild || o->gram_stale > 1) {
            // history installed from outside (or pushes without a direction in between): rebuild
            // every row/column of the caches, oldest pivot first, newest last
            for (int pv = k - 1; pv >= 1; --pv) DZO_TRY(gram_pass<T>(o, pv, false));
        }
        DZO_TRY(gram_pass<T>(o, 0, true));
    }
    o->gram_rebuild = false;
    o->gram_stale = 0;
    o->scalars_ready = true;
    return DZO_OK;
}

template <typename T> static int32_t direction_gram(dzo_lbfgs_s *o) {
    OptCore &c = o->core;
    hipStream_t s = c.stream;
    const int k = o->k;
    DZO_TRY(gram_scalars<T>(o));
    o->scalars_ready = false;          // (a later direction call may see a different gradient)
    CombineParams<T> cp;
    memset(&cp, 0, sizeof(cp));
    cp.n = c.n; cp.g = (const T *)c.g; cp.d = (T *)o->d; cp.k = k;
    for (int i = 0; i < k; ++i) { cp.s[i] = o->s_slot<T>(o->slot_of(i)); cp.y[i] = o->y_slot<T>(o->slot_of(i)); }
    cp.alpha = o->alpha; cp.coef = o->coef; cp.scale = o->scale;
    cp.rowbytes = o->rowbytes;
    cp.debug = tune("DZO_TUNE_TP_DEBUG", 0) == 2;
    cp.fresh_plain = o->combine_fresh_plain;
    const bool vec = al16(c.g);
    int u = o->combine_u;
    while (u > 1 && (c.n + (int64_t)kBlock * (vec ? Vec16<T>::N : 1) * u - 1) / ((int64_t)kBlock * (vec ? Vec16<T>::N : 1) * u) < (int64_t)ctx().cus * 4) u >>= 1;
    int64_t per_block = (int64_t)kBlock * (vec ? Vec16<T>::N : 1) * u;
    int64_t blocks = (c.n + per_block - 1) / per_block;
    {
        DZO_TIMED("lbfgs_combine", s);
#define CB(UU)                                                                                                   \
    do {                                                                                                         \
        auto kern = o->blocked ? combine_kernel<T, true, UU, true, true>                                         \
                    : (vec && o->combine_nts) ? combine_kernel<T, true, UU, true>                                \
                    : vec ? combine_kernel<T, true, UU, false> : combine_kernel<T, false, UU, false>;            \
        const int bpc = o->combine_blocks_per_cu > 0 ? o->combine_blocks_per_cu : resident_blocks((const void *)kern); \
        const int64_t cap = (int64_t)ctx().cus * bpc;                                                            \
        if (blocks > cap) blocks = cap;                                                                          \
        hipLaunchKernelGGL(kern, dim3((int)(blocks < 1 ? 1 : blocks)), dim3(kBlock), 0, s, cp);                 \
    } while (0)
        if (u == 1) CB(1); else if (u == 4) CB(4); else CB(2);
#undef CB
    }
    DZO_HIP(hipGetLastError());
    return DZO_OK;
}

static int32_t lbfgs_flush_rho(dzo_lbfgs_s *o);

// the deferred tail of the last accepted step, now (somebody needs delta_point / delta_gradient / rho or the buffers they
// are formed from before a Gram pass could form them on its way): accept_delta_rho_kernel, as the undeferred path runs it
static int32_t lbfgs_flush_post(dzo_lbfgs_s *o) {
    if (!o->post_pending) return DZO_OK;
    OptCore &c = o->core;
    o->post_pending = false;
    void *dx = o->s_slot_v(o->newest), *dg = o->y_slot_v(o->newest);
    const bool vec = al16(c.x) && al16(dx) && al16(c.g) && al16(o->post_g_old) && al16(dg);
    const int grid = stream_grid(c.n, (vec ? 16 / (int)dtype_size(c.dtype) : 1) * 2);
    {
        DZO_TIMED("lbfgs_accept_delta_rho", c.stream);
        if (vec) DZO_DISPATCH(c.dtype, hipLaunchKernelGGL((accept_delta_rho_kernel<T, true>), dim3(grid), dim3(kBlock), 0, c.stream, c.n, (const T *)c.x, (T *)dx,
                                                          (const T *)c.g, (const T *)o->post_g_old, (T *)dg, c.partials()));
        else DZO_DISPATCH(c.dtype, hipLaunchKernelGGL((accept_delta_rho_kernel<T, false>), dim3(grid), dim3(kBlock), 0, c.stream, c.n, (const T *)c.x, (T *)dx,
                                                      (const T *)c.g, (const T *)o->post_g_old, (T *)dg, c.partials()));
    }
    DZO_HIP(hipGetLastError());
    // rho of the newest pair: summed lazily like after any two-pass step (lbfgs_finish_push's rho_pending)
    if (o->mode == DZO_TWOLOOP_GRAM) { o->rho_pending = true; o->rho_pending_count = grid; o->rho_pending_slot = o->newest; }
    else {
        DZO_TIMED("lbfgs_rho_finish", c.stream);
        hipLaunchKernelGGL(finish_to_kernel, dim3(1), dim3(kBlock), 0, c.stream, c.partials(), grid, o->rho + o->newest, c.dtype == DZO_F32 ? 1 : 0, (const int32_t *)nullptr);
        DZO_HIP(hipGetLastError());
    }
    return DZO_OK;
}

static int32_t lbfgs_direction(dzo_lbfgs_s *o) {
    OptCore &c = o->core;
    // the deferred tail rides in the Gram pass only in its plain form: GRAM mode, one pass with pivot 0 served from registers
    if (o->post_pending && (o->mode != DZO_TWOLOOP_GRAM || o->k == 0 || o->blocked || o->gram_rebuild || o->gram_stale > 1 || !o->gram_skip0 ||
                            o->gram_variant != 1 || o->spec_scalars || o->scalars_ready || o->gram_ready ||
                            !(al16(c.x) && al16(c.g) && al16(o->post_g_old) && al16(o->s_slot_v(o->newest)) && al16(o->y_slot_v(o->newest)))))
        DZO_TRY(lbfgs_flush_post(o));
    if (o->k == 0) {                                      // :438 then the :443 guard
        DZO_HIP(hipMemcpyAsync(o->d, c.g, (size_t)c.n * dtype_size(c.dtype), hipMemcpyDeviceToDevice, c.stream));
        return DZO_OK;
    }
    if (o->mode == DZO_TWOLOOP_CHAIN) {
        DZO_TRY(lbfgs_flush_rho(o));
        DZO_DISPATCH(c.dtype, return direction_chain<T>(o));
    }
    DZO_DISPATCH(c.dtype, return direction_gram<T>(o));
    return DZO_OK;
}

static int32_t lbfgs_finish_push(dzo_lbfgs_s *o, int grid, bool rho_done, bool rho_final = false, bool tiles_written = false);

static int32_t lbfgs_post_gradient(dzo_lbfgs_s *o) {
    OptCore &c = o->core;
    hipStream_t s = c.stream;
    const bool vec = al16(c.g);
    const int grid = stream_grid(c.n, (vec ? 16 / (int)dtype_size(c.dtype) : 1) * 2);
    {
        DZO_TIMED("lbfgs_delta_rho", s);
        if (c.dtype == DZO_F64) {
            if (vec) hipLaunchKernelGGL((delta_rho_kernel<double, true>), dim3(grid), dim3(kBlock), 0, s, c.n, (const double *)c.g, (double *)c.dg, (const double *)c.dx, c.partials());
            else hipLaunchKernelGGL((delta_rho_kernel<double, false>), dim3(grid), dim3(kBlock), 0, s, c.n, (const double *)c.g, (double *)c.dg, (const double *)c.dx, c.partials());
        } else {
            if (vec) hipLaunchKernelGGL((delta_rho_kernel<float, true>), dim3(grid), dim3(kBlock), 0, s, c.n, (const float *)c.g, (float *)c.dg, (const float *)c.dx, c.partials());
            else hipLaunchKernelGGL((delta_rho_kernel<float, false>), dim3(grid), dim3(kBlock), 0, s, c.n, (const float *)c.g, (float *)c.dg, (const float *)c.dx, c.partials());
        }
    }
    DZO_HIP(hipGetLastError());
    return lbfgs_finish_push(o, grid, false);
}

// :505 rho of the pair being pushed (fixed-order sum of the partials), then the ring rotation
// rho of the newest pair is summed lazily: in GRAM mode the next direction's gram_reduce launch
// carries it; anything else that needs rho on the device or host flushes it first.
static int32_t lbfgs_flush_rho(dzo_lbfgs_s *o) {
    DZO_TRY(lbfgs_flush_post(o));
    if (o->gram_ready) {
        // rho of the newest pair still sits in the single pass's dots (value 4 of pair 0 = s_p.y_p).
        // Sum it exactly as the coming gram_reduce launch will (same kernel, same order), so that
        // reading rho_history never changes what the optimizer computes next.
        OptCore &c = o->core;
        double *tmp = o->gram_partials + (size_t)kGramValues * kMaxHistory * o->gram_grid * kWaves + kGramValues * kMaxHistory - 1;   // last, unused slot of the reduced values
        hipLaunchKernelGGL(gram_reduce_kernel, dim3(1), dim3(kBlock), 0, c.stream,
                           o->gram_partials + (size_t)4 * o->gram_ready_grid, o->gram_ready_grid, tmp, 1,
                           (const double *)nullptr, 0, (double *)nullptr, 0);
        hipLaunchKernelGGL(finish_to_kernel, dim3(1), dim3(kBlock), 0, c.stream, (const double *)tmp, 1, o->rho + o->newest,
                           c.dtype == DZO_F32 ? 1 : 0, (const int32_t *)nullptr);
        DZO_HIP(hipGetLastError());
    }
    if (!o->rho_pending) return DZO_OK;
    OptCore &c = o->core;
    DZO_TIMED("lbfgs_rho_finish", c.stream);
    hipLaunchKernelGGL(finish_to_kernel, dim3(1), dim3(kBlock), 0, c.stream, c.partials(), o->rho_pending_count,
                       o->rho + o->rho_pending_slot, c.dtype == DZO_F32 ? 1 : 0, (const int32_t *)nullptr);
    DZO_HIP(hipGetLastError());
    o->rho_pending = false;
    return DZO_OK;
}

static int32_t lbfgs_rho_finish(dzo_lbfgs_s *o, int grid, const int32_t *gate) {
    OptCore &c = o->core;
    DZO_TIMED("lbfgs_rho_finish", c.stream);
    // stored by slot, rounded to T like the reference's dot
    hipLaunchKernelGGL(finish_to_kernel, dim3(1), dim3(kBlock), 0, c.stream, c.partials(), grid, o->rho + o->spare(),
                       c.dtype == DZO_F32 ? 1 : 0, gate);
    DZO_HIP(hipGetLastError());
    return DZO_OK;
}

// Tail of an accepted step, enqueued BEFORE the host knows the trial's outcome (gated on the
// device-side decision): :145 + :478-480 + :505 for the built-in chained Rosenbrock objective.
static int32_t lbfgs_speculative_tail(void *self, const int32_t *gate) {
    dzo_lbfgs_s *o = static_cast<dzo_lbfgs_s *>(self);
    OptCore &c = o->core;
    int grid = 0;
    DZO_TRY(problem_fused_post_async(c.problem, c.stream, c.x, c.dx, c.g, c.dg, c.partials(), &grid, gate));
    o->tail_grid = grid;
    if (o->mode == DZO_TWOLOOP_GRAM) return DZO_OK;      // rho rides on the next gram_reduce launch
    return lbfgs_rho_finish(o, grid, gate);
}

// ---------------------------------------------------------------------------- blocked ring, host side
template <typename T> static inline PointDecor<T> point_decor(const RingDecor &r) {
    PointDecor<T> d;
    // :247 lambda + lambda in T (as problem_grad_async)
    d.two_lambda = sizeof(T) == 4 ? (T)((float)r.l2 + (float)r.l2) : (T)(r.l2 + r.l2);
    d.l2_on = r.l2 != 0.0 ? 1 : 0;
    d.bg_lo = r.bg_on ? (T)r.bg_lo : -std::numeric_limits<T>::infinity(); d.bg_hi = r.bg_on ? (T)r.bg_hi : std::numeric_limits<T>::infinity();
    d.cons_lo = r.cons_on ? (T)r.cons_lo : -std::numeric_limits<T>::infinity(); d.cons_hi = r.cons_on ? (T)r.cons_hi : std::numeric_limits<T>::infinity();
    return d;
}
static inline RingDecor ring_decor_of(const dzo_problem_s *p) {
    RingDecor r;
    if (p) { r.l2 = p->l2; r.bg_on = p->bg_on; r.bg_lo = p->bg_lo; r.bg_hi = p->bg_hi; r.cons_on = p->cons_on; r.cons_lo = p->cons_lo; r.cons_hi = p->cons_hi; }
    return r;
}
// the chained objectives the point pass serves (ChainObj): -1 = none of them
static inline int ring_obj_of(const dzo_problem_s *p) {
    if (!p) return -1;
    return p->kind == DZO_PROBLEM_ROSENBROCK_CHAIN ? 0 : (p->kind == DZO_PROBLEM_QUADRATIC_CHAIN ? 1 : (p->kind == DZO_PROBLEM_LSE ? 2 : -1));
}
// vectors of 16 bytes a ring stream holds: the last one is padded with phantom elements when n is ragged (see load_vec_tail)
template <typename T> static inline int64_t ring_nvec(const dzo_lbfgs_s *o) { return (o->core.n + Vec16<T>::N - 1) / Vec16<T>::N; }
static inline bool ring_ragged(const dzo_lbfgs_s *o) { return o->core.n % (16 / (int64_t)dtype_size(o->core.dtype)) != 0; }
template <typename T> static void ring_scatter(dzo_lbfgs_s *o, const void *lin, void *stream) {
    const int grid = stream_grid(o->ring_rows * 64, 1);
    hipLaunchKernelGGL(ring_scatter_kernel<T>, dim3(grid), dim3(kBlock), 0, o->core.stream, o->core.n, ring_nvec<T>(o), (const T *)lin, (T *)stream, o->rowbytes);
}
template <typename T> static void ring_gather(dzo_lbfgs_s *o, const void *stream, void *lin) {
    const int64_t nvec = ring_nvec<T>(o);
    const int grid = stream_grid(nvec, 1);
    hipLaunchKernelGGL(ring_gather_kernel<T>, dim3(grid), dim3(kBlock), 0, o->core.stream, o->core.n, nvec, (const T *)stream, (T *)lin, o->rowbytes);
}

template <typename T> static void ring_gather_diff(dzo_lbfgs_s *o, const void *a, const void *b, void *lin) {
    const int64_t nvec = ring_nvec<T>(o);
    const int grid = stream_grid(nvec, 1);
    hipLaunchKernelGGL(ring_gather_diff_kernel<T>, dim3(grid), dim3(kBlock), 0, o->core.stream, o->core.n, nvec, (const T *)a, (const T *)b, (T *)lin, o->rowbytes);
}
template <typename T> static void ring_compare(dzo_lbfgs_s *o, const void *stream, const void *lin) {
    const int64_t nvec = ring_nvec<T>(o);
    const int grid = stream_grid(nvec, 1);
    hipLaunchKernelGGL(ring_compare_kernel<T>, dim3(grid), dim3(kBlock), 0, o->core.stream, o->core.n, nvec, (const T *)stream, (const T *)lin, o->rowbytes, o->xg_differs);
}
template <typename T> static void ring_diff(dzo_lbfgs_s *o, void *a, const void *b) {
    const int grid = stream_grid(o->ring_rows * 64, 1);
    hipLaunchKernelGGL(ring_diff_kernel<T>, dim3(grid), dim3(kBlock), 0, o->core.stream, o->ring_rows, (T *)a, (const T *)b, o->rowbytes);
}

// point ring: make sure the gradient tiles of `slot` hold the gradient of the point in its point tiles
static int32_t lbfgs_ensure_g(dzo_lbfgs_s *o, int slot) {
    if (!o->points || (o->g_valid >> slot & 1u)) return DZO_OK;
    OptCore &c = o->core;
    DZO_TIMED("lbfgs_ring_regrad", c.stream);
    const int grid = stream_grid(o->ring_rows * 64, 1);
    DZO_DISPATCH(c.dtype, hipLaunchKernelGGL(ring_regrad_kernel<T>, dim3(grid), dim3(kBlock), 0, c.stream, c.n, ring_nvec<T>(o),
                                             (const T *)o->s_slot_v(slot), (T *)o->y_slot_v(slot), o->rowbytes, point_decor<T>(o->ring_dec), o->ring_dec.any() ? 1 : 0,
                                             o->ring_obj, (T)o->ring_obj_lambda, (const T *)(o->ring_obj == 2 ? o->s_slot_v(o->nslots) : nullptr),
                                             (const double *)(o->pscal ? o->pscal + 2 * slot : nullptr), o->ring_obj_lambda));
    DZO_HIP(hipGetLastError());
    o->g_valid |= 1u << slot;
    return DZO_OK;
}

// point ring: current_point / current_gradient into the caller's arrays (the optimizer aliases them, :393)
// hand_out = false (dzo_lbfgs_read): the arrays are filled for a copy to the host that the library makes itself; nobody
// gets a pointer, so the next step need not check them against the ring (lbfgs_adopt_host_writes: a regrad, two compare
// passes and a host round trip -- what a monitoring read of current_point used to cost the step behind it).
static int32_t lbfgs_points_settle(dzo_lbfgs_s *o, bool hand_out = true) {
    if (!o->points) return DZO_OK;
    if (hand_out) o->xg_host_may_write = true;            // (whoever is handed the arrays may write into them)
    if (!o->xg_lin_stale) return DZO_OK;
    DZO_TRY(lbfgs_ensure_g(o, o->newest));
    DZO_TIMED("lbfgs_ring_gather", o->core.stream);
    DZO_DISPATCH(o->core.dtype, (ring_gather<T>(o, o->s_slot_v(o->newest), o->x_user), ring_gather<T>(o, o->y_slot_v(o->newest), o->g_user)));
    DZO_HIP(hipGetLastError());
    o->xg_lin_stale = false;
    return DZO_OK;
}

// delta_point / delta_gradient as contiguous vectors (blocked ring: gathered from the newest pair when a
// single-pass step pushed it without materialising them)
static int32_t lbfgs_refresh_lin(dzo_lbfgs_s *o) {
    if (!o->blocked || !o->lin_stale) return DZO_OK;
    if (o->points) {                                      // pair 0 = point 0 - point 1
        if (o->k < 1) { o->lin_stale = false; return DZO_OK; }
        const int a = o->slot_of(0), b = o->slot_of(1);
        DZO_TRY(lbfgs_ensure_g(o, a)); DZO_TRY(lbfgs_ensure_g(o, b));
        DZO_TIMED("lbfgs_ring_gather", o->core.stream);
        DZO_DISPATCH(o->core.dtype, (ring_gather_diff<T>(o, o->s_slot_v(a), o->s_slot_v(b), o->dx_lin),
                                     ring_gather_diff<T>(o, o->y_slot_v(a), o->y_slot_v(b), o->dg_lin)));
        DZO_HIP(hipGetLastError());
        o->lin_stale = false;
        return DZO_OK;
    }
    DZO_TIMED("lbfgs_ring_gather", o->core.stream);
    DZO_DISPATCH(o->core.dtype, (ring_gather<T>(o, o->s_slot_v(o->newest), o->dx_lin), ring_gather<T>(o, o->y_slot_v(o->newest), o->dg_lin)));
    DZO_HIP(hipGetLastError());
    o->lin_stale = false;
    return DZO_OK;
}

static void lbfgs_mark_unsettled(dzo_lbfgs_s *o);
static int32_t lbfgs_materialize_d(dzo_lbfgs_s *o);
static int32_t lbfgs_unblock(dzo_lbfgs_s *o);

// Point ring -> pair ring, in place: the caller's arrays receive point 0, then slot_of(i) <- point i - point i+1
// from the newest pair to the oldest (each subtraction reads two slots no earlier one has touched).  The pair
// ring keeps slots, rho, Gram caches and the scalars computed behind the last decision: the pairs are the same
// numbers.  One way only (pairs cannot be turned back into points).
static int32_t lbfgs_leave_points(dzo_lbfgs_s *o) {
    if (!o->points) return DZO_OK;
    std::lock_guard<std::recursive_mutex> lk(o->mu);
    OptCore &c = o->core;
    DZO_TRY(lbfgs_materialize_d(o));                      // step_direction, delta_point / delta_gradient and the caller's
    DZO_TRY(lbfgs_points_settle(o));                      // arrays while the points still exist
    DZO_TRY(lbfgs_refresh_lin(o));
    for (int i = 0; i <= o->k; ++i) DZO_TRY(lbfgs_ensure_g(o, o->slot_of(i)));   // the gradients of all k + 1 points
    {
        DZO_TIMED("lbfgs_ring_to_pairs", c.stream);
        for (int i = 0; i < o->k; ++i) {
            const int a = o->slot_of(i), b = o->slot_of(i + 1);
            DZO_DISPATCH(c.dtype, (ring_diff<T>(o, o->s_slot_v(a), o->s_slot_v(b)), ring_diff<T>(o, o->y_slot_v(a), o->y_slot_v(b))));
        }
        if (o->k == 0) {                                  // an empty pair ring is a zeroed one (:366-374)
            DZO_HIP(hipMemsetAsync(o->S, 0, o->ring_bytes, c.stream));
        }
        DZO_HIP(hipGetLastError());
    }
    o->points = false;
    lbfgs_mark_unsettled(o);
    // the pair kernels of the tile ring read whole 16-byte vectors of the caller's contiguous arrays: a ragged n continues
    // on the slabs (the two-pass kernels with their element tails)
    if (ring_ragged(o)) return lbfgs_unblock(o);
    return DZO_OK;
}

// Leave the blocked layout for good (CHAIN mode walks the history as plain vectors): every live pair is
// gathered into a slab ring, the blocked ring is freed.
static int32_t lbfgs_unblock(dzo_lbfgs_s *o) {
    if (!o->blocked) return DZO_OK;
    DZO_TRY(lbfgs_leave_points(o));
    if (!o->blocked) return DZO_OK;                         // (a ragged point ring: leaving the points already ended here)
    OptCore &c = o->core;
    DZO_TRY(lbfgs_refresh_lin(o));
    const size_t es = dtype_size(c.dtype);
    const int m1 = o->nslots;                               // (the slab ring keeps the slot numbering)
    const size_t slab = (size_t)m1 * (size_t)o->stride * es;
    void *ring = nullptr;
    hipError_t e = hipMalloc(&ring, 2 * slab);
    if (e != hipSuccess) { set_error("out of device memory converting the history ring (%zu bytes)", 2 * slab); (void)hipGetLastError(); return DZO_ERR_NOMEM; }
    DZO_HIP(hipMemsetAsync(ring, 0, 2 * slab, c.stream));
    void *Yb = (char *)ring + (size_t)o->stride * es;
    const int64_t ps = 2 * o->stride;
    for (int i = 0; i < o->k; ++i) {
        const int slot = o->slot_of(i);
        DZO_DISPATCH(c.dtype, (ring_gather<T>(o, o->s_slot_v(slot), (char *)ring + (size_t)slot * ps * es),
                               ring_gather<T>(o, o->y_slot_v(slot), (char *)Yb + (size_t)slot * ps * es)));
    }
    DZO_HIP(hipGetLastError());
    // the deltas of the last step live in the spare slots of a slab ring
    const int sp = o->spare();
    DZO_HIP(hipMemcpyAsync((char *)ring + (size_t)sp * ps * es, o->dx_lin, (size_t)c.n * es, hipMemcpyDeviceToDevice, c.stream));
    DZO_HIP(hipMemcpyAsync((char *)Yb + (size_t)sp * ps * es, o->dg_lin, (size_t)c.n * es, hipMemcpyDeviceToDevice, c.stream));
    DZO_HIP(hipStreamSynchronize(c.stream));
    (void)hipFree(o->S);
    o->S = ring; o->Y = Yb; o->interleaved = true; o->pair_stride = ps;
    o->blocked = false;
    // (after a push delta_point / delta_gradient are the newest pair; before the first step the zero-filled spare)
    if (c.iteration_count > 0 && o->k > 0) { c.dx = o->s_slot_v(o->newest); c.dg = o->y_slot_v(o->newest); }
    else o->refresh_delta_ptrs();
    return DZO_OK;
}

static int32_t lbfgs_finish_push(dzo_lbfgs_s *o, int grid, bool rho_done, bool rho_final, bool tiles_written) {
    OptCore &c = o->core;
    const int sp = o->spare();
    if (o->blocked) {
        if (tiles_written) {
            o->lin_stale = true;                          // (the single pass wrote the pair's tiles, not dx_lin / dg_lin)
        } else {
            // a two-pass step leaves delta_point / delta_gradient in the contiguous vectors: copy them into the
            // spare slot's tiles (4 n T; only after a rejected first trial, with callbacks, ...)
            DZO_TIMED("lbfgs_ring_scatter", c.stream);
            DZO_DISPATCH(c.dtype, (ring_scatter<T>(o, o->dx_lin, o->s_slot_v(sp)), ring_scatter<T>(o, o->dg_lin, o->y_slot_v(sp))));
            DZO_HIP(hipGetLastError());
            o->lin_stale = false;
        }
    }
    if (o->reset_on_push) {                               // legacy :609  _history_count[] = 0
        o->k = 0;
        o->reset_on_push = false;
        o->history_resets += 1;
    }
    if (rho_final) {
        o->rho_pending = false;                           // rho[spare] already holds s.y
    } else if (o->mode == DZO_TWOLOOP_GRAM) {
        // defer the final sum of rho to the next direction's gram_reduce launch
        o->rho_pending = true;
        o->rho_pending_count = rho_done ? o->tail_grid : grid;
        o->rho_pending_slot = sp;
    } else if (!rho_done) {
        DZO_TRY(lbfgs_rho_finish(o, grid, nullptr));
    }
    // :482-496  pushfirst!: the spare slots (= delta_point, delta_gradient) become pair 0
    o->newest = sp;
    if (o->k < o->m) o->k += 1;
    if (o->n_alpha < o->m) o->n_alpha += 1;               // :498-500
    o->gram_stale += 1;
    // delta_point / delta_gradient keep pointing at the pushed pair until the next search
    // begins, exactly as the reference's fields hold the last step's deltas
    c.iteration_count += 1;                               // :507
    return DZO_OK;
}

// scratch for the safeguards' dot products: the upper half of the workspace partials (the lower
// half may still hold the pending rho partials); the result lands in result()[0]
static int32_t lbfgs_aux_dot(dzo_lbfgs_s *o, const void *a, const void *b, double *out) {
    OptCore &c = o->core;
    double v = 0;
    DZO_TRY(dot_blocking(c.stream, c.n, c.dtype, a, b, c.partials() + kMaxPartialBlocks, c.host, &v));
    *out = round_to_dtype(c.dtype, v);
    return DZO_OK;
}

// d = -(last_step_length / ||g||) g   (legacy/DZOptimization.jl:594-596, :688-690)
static int32_t lbfgs_steepest(dzo_lbfgs_s *o) {
    OptCore &c = o->core;
    double ss = 0;
    DZO_TRY(lbfgs_aux_dot(o, c.g, c.g, &ss));
    const double inv = c.dtype == DZO_F32 ? (double)(1.0f / sqrtf((float)ss)) : 1.0 / sqrt(ss);   // Kernels.jl:141
    const double sc = round_to_dtype(c.dtype, -o->last_step_length * inv);
    DZO_DISPATCH(c.dtype, launch_scal_oop<T>(c.stream, c.n, (T *)o->d, (T)sc, (const T *)c.g));
    DZO_HIP(hipGetLastError());
    return DZO_OK;
}

static int32_t lbfgs_eval_at(OptCore &c, void *x, double *f) {
    if (c.objective) {
        DZO_HIP(hipStreamSynchronize(c.stream));
        *f = round_to_dtype(c.dtype, c.objective(c.cb_ctx, x));
        return DZO_OK;
    }
    DZO_TRY(problem_eval_async(c.problem, c.stream, x, c.result()));
    DZO_HIP(hipMemcpyAsync(c.host, c.result(), sizeof(double), hipMemcpyDeviceToHost, c.stream));
    DZO_HIP(hipStreamSynchronize(c.stream));
    *f = round_to_dtype(c.dtype, c.host[0]);
    return DZO_OK;
}

static int32_t lbfgs_grad_at(OptCore &c, void *g, void *x) {
    if (c.gradient) {
        DZO_HIP(hipStreamSynchronize(c.stream));
        c.gradient(c.cb_ctx, g, x);
        return DZO_OK;
    }
    return problem_grad_async(c.problem, c.stream, g, x);
}

// Strong-Wolfe search by bisection / doubling on the LineSearchEvaluator quotients
// (src/DZOptimization.jl:65-92): improvement_ratio >= c1, |slope_ratio| <= c2.  Host-driven:
// every evaluation ends in a scalar read-back, like the reference's evaluator call.  On
// success x, f, df, dx, g, dg are all final (the trial gradient is reused).
static int32_t lbfgs_wolfe_search(dzo_lbfgs_s *o, bool *accepted) {
    OptCore &c = o->core;
    hipStream_t s = c.stream;
    const int32_t dt = c.dtype;
    const size_t bytes = (size_t)c.n * dtype_size(dt);
    *accepted = false;
    if (!o->xt) {
        const size_t padded = (size_t)((c.n + 63) / 64 * 64) * dtype_size(dt);
        DZO_HIP(hipMalloc(&o->xt, padded));
        DZO_HIP(hipMalloc(&o->gt, padded));
    }
    c.last_trials = 0;
    double overlap = 0;
    DZO_TRY(lbfgs_aux_dot(o, c.g, o->d, &overlap));                  // the evaluator's `overlap` = g.d
    if (!(overlap < 0.0)) { c.is_stuck = true; return DZO_OK; }      // not a descent direction (or NaN)
    const double tmax = dt == DZO_F32 ? 3.4028234663852886e38 : 1.7976931348623157e308;
    double t = 1.0, lo = 0.0, hi = -1.0;                             // hi < 0: no upper bound yet
    for (int32_t it = 0; it < o->wolfe_max_evals; ++it) {
        DZO_DISPATCH(dt, launch_axpy_oop<T>(s, c.n, (T *)o->xt, (T)t, (const T *)o->d, (const T *)c.x));   // :69-70
        DZO_HIP(hipGetLastError());
        bool feasible = true;
        if (c.constraint) {                                          // :71-79
            DZO_HIP(hipStreamSynchronize(s));
            feasible = c.constraint(c.cb_ctx, o->xt) != 0;
        } else if (c.box_on) {
            DZO_TRY(box_clamp_async(s, c.n, dt, o->xt, c.box_lo, c.box_hi));
        }
        double f_t = tmax, ir = -tmax, sr = tmax;
        if (feasible) {
            DZO_TRY(lbfgs_eval_at(c, o->xt, &f_t));                  // :80-81
            ir = round_to_dtype(dt, round_to_dtype(dt, f_t - c.f) / round_to_dtype(dt, t * overlap));   // :84
            DZO_TRY(lbfgs_grad_at(c, o->gt, o->xt));                 // :87
            double gd = 0;
            DZO_TRY(lbfgs_aux_dot(o, o->gt, o->d, &gd));
            sr = round_to_dtype(dt, gd / overlap);                   // :88-89
        }
        c.last_trials += 1;
        if (!(ir >= o->wolfe_c1) || !(f_t < c.f)) hi = t;            // Armijo fails (NaN counts as failure)
        else if (sr > o->wolfe_c2) lo = t;                           // still descending steeply
        else if (sr < -o->wolfe_c2) hi = t;                          // overshot the minimiser
        else {
            DZO_HIP(hipMemcpyAsync(c.dx, o->xt, bytes, hipMemcpyDeviceToDevice, s));
            DZO_DISPATCH(dt, launch_axpby<T>(s, c.n, (T)-1, (const T *)c.x, (T)1, (T *)c.dx));     // dx = x_new - x_old
            DZO_HIP(hipMemcpyAsync(c.x, o->xt, bytes, hipMemcpyDeviceToDevice, s));
            c.df = round_to_dtype(dt, f_t - c.f);
            c.f = f_t;
            DZO_HIP(hipMemcpyAsync(c.dg, o->gt, bytes, hipMemcpyDeviceToDevice, s));
            DZO_DISPATCH(dt, launch_axpby<T>(s, c.n, (T)-1, (const T *)c.g, (T)1, (T *)c.dg));     // dg = g_new - g_old
            DZO_HIP(hipMemcpyAsync(c.g, o->gt, bytes, hipMemcpyDeviceToDevice, s));
            DZO_HIP(hipGetLastError());
            *accepted = true;
            return DZO_OK;
        }
        const double t_next = round_to_dtype(dt, hi < 0.0 ? t + t : (lo + hi) * 0.5);
        if (t_next == lo || t_next == hi || !(t_next > 0.0)) break;  // interval exhausted
        t = t_next;
    }
    c.is_stuck = true;
    return DZO_OK;
}

static bool lbfgs_ensure_twins(dzo_lbfgs_s *o);
static void lbfgs_mark_unsettled(dzo_lbfgs_s *o);
// one line search along o->d followed, when it succeeds, by the post-gradient phase and the push
static int32_t lbfgs_search_and_post(dzo_lbfgs_s *o, int trials_rejected = 0) {
    OptCore &c = o->core;
    DZO_TRY(lbfgs_flush_post(o));                         // (a trial overwrites x: a deferred tail that no Gram pass consumed is formed first)
    o->refresh_delta_ptrs();                              // deltas move to the spare slots
    o->scalars_ready = false; o->gram_ready = false;      // x, g and the history are about to change
    o->spec_scalars = false;
    const bool safeguards = o->descent_check || o->sd_fallback;
    if (o->line_search == 1) {
        bool accepted = false;
        DZO_TRY(lbfgs_wolfe_search(o, &accepted));
        if (!accepted) return DZO_OK;
        if (safeguards) {
            double ss = 0;
            DZO_TRY(lbfgs_aux_dot(o, c.dx, c.dx, &ss));
            o->last_step_length = c.dtype == DZO_F32 ? (double)sqrtf((float)ss) : sqrt(ss);
        }
        double r = 0;
        DZO_TRY(lbfgs_aux_dot(o, c.dx, c.dg, &r));                   // :505
        DZO_HIP(hipMemcpyAsync(o->rho + o->spare(), &r, sizeof(double), hipMemcpyHostToDevice, c.stream));
        DZO_HIP(hipStreamSynchronize(c.stream));
        return lbfgs_finish_push(o, 0, true, true);
    }
    const bool fused = o->fused_post && !c.objective && !c.gradient && !c.constraint &&
                       problem_has_fused_post(c.problem, c.x, c.dx, c.g, c.dg);
    // General path (callbacks, any objective without a fused tail): the accepted step's tail as ONE pass
    // (accept_delta_rho_kernel) with the gradient callback writing into the other gradient buffer -- no axpby, no copy of
    // g_old (DZO_TUNE_GENERIC_POST=0: the separate kernels, :145 / :478 / :480 / :505 one launch each)
    const bool one_pass_tail = !fused && tune("DZO_TUNE_GENERIC_POST", 1) != 0 && lbfgs_ensure_twins(o);
    c.defer_delta = fused || one_pass_tail;
    c.speculative_tail = (fused && o->speculate) ? lbfgs_speculative_tail : nullptr;
    c.speculative_self = o;
    int32_t rc = core_backtracking_step(c, 1.0, o->d, trials_rejected);    // :473
    const bool speculated = c.speculative_tail != nullptr;
    c.defer_delta = false;
    c.speculative_tail = nullptr;
    DZO_TRY(rc);
    if (c.is_stuck) {                                     // :474-476
        // delta_point holds x_old (:118, the first trial's backup); delta_gradient is still the
        // previous step's, which lives in the newest pair of the ring
        if (o->blocked) {
            if (o->k > 0) {                               // (the search used dg_lin as its scratch)
                DZO_DISPATCH(c.dtype, ring_gather<T>(o, o->y_slot_v(o->newest), o->dg_lin));
                DZO_HIP(hipGetLastError());
            }
            o->lin_stale = false;
        } else if (o->k > 0) {
            c.dg = o->y_slot_v(o->newest);
        }
        return DZO_OK;
    }
    int32_t done = -1;
    if (fused && speculated) {
        done = lbfgs_finish_push(o, 0, true);             // the gated tail already ran
    } else if (fused) {
        // :145 + :478-480 + partials of :505 in one pass
        int grid = 0;
        DZO_TRY(problem_fused_post_async(c.problem, c.stream, c.x, c.dx, c.g, c.dg, c.partials(), &grid, nullptr));
        done = lbfgs_finish_push(o, grid, false);
    } else if (one_pass_tail) {
        void *g_old = c.g;
        c.g = o->g_twin;                                  // :479 gradient!(g, x) into the other buffer (the caller's array is settled when somebody looks)
        int32_t rcg = core_gradient(c);
        if (rcg != DZO_OK) { c.g = g_old; return rcg; }
        o->g_twin = g_old;
        if (!o->blocked && o->mode == DZO_TWOLOOP_GRAM && !safeguards && tune("DZO_TUNE_GENERIC_POST", 1) >= 2) {
            // deferred: the next step's Gram pass forms the pair on its way (gram_pass_lanes_kernel<POST>); the getters and
            // everything else that needs it earlier call lbfgs_flush_post
            lbfgs_mark_unsettled(o);
            done = lbfgs_finish_push(o, 0, true, true);
            o->post_pending = true; o->post_g_old = g_old;
            DZO_TRY(done);
            return DZO_OK;
        }
        const bool vec = al16(c.x) && al16(c.dx) && al16(c.g) && al16(g_old) && al16(c.dg);
        const int grid = stream_grid(c.n, (vec ? 16 / (int)dtype_size(c.dtype) : 1) * 2);
        {
            DZO_TIMED("lbfgs_accept_delta_rho", c.stream);
            if (vec) DZO_DISPATCH(c.dtype, hipLaunchKernelGGL((accept_delta_rho_kernel<T, true>), dim3(grid), dim3(kBlock), 0, c.stream, c.n, (const T *)c.x, (T *)c.dx,
                                                              (const T *)c.g, (const T *)g_old, (T *)c.dg, c.partials()));
            else DZO_DISPATCH(c.dtype, hipLaunchKernelGGL((accept_delta_rho_kernel<T, false>), dim3(grid), dim3(kBlock), 0, c.stream, c.n, (const T *)c.x, (T *)c.dx,
                                                          (const T *)c.g, (const T *)g_old, (T *)c.dg, c.partials()));
        }
        DZO_HIP(hipGetLastError());
        lbfgs_mark_unsettled(o);
        done = lbfgs_finish_push(o, grid, false);
    } else {
        DZO_HIP(hipMemcpyAsync(c.dg, c.g, (size_t)c.n * dtype_size(c.dtype), hipMemcpyDeviceToDevice, c.stream)); // :478
        DZO_TRY(core_gradient(c));                        // :479
        done = lbfgs_post_gradient(o);                    // :480-507
    }
    DZO_TRY(done);
    if (safeguards) {                                     // legacy :625-627 (delta_point stays valid after the push)
        double ss = 0;
        DZO_TRY(lbfgs_aux_dot(o, c.dx, c.dx, &ss));
        o->last_step_length = c.dtype == DZO_F32 ? (double)sqrtf((float)ss) : sqrt(ss);
    }
    return DZO_OK;
}

static inline bool al16v(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// the twin buffers of x and g (the pair-ring pass writes its trial point / gradient there; the general path's gradient
// callback writes g_new there): one slab, allocated on first use; false when the allocation fails
static bool lbfgs_ensure_twins(dzo_lbfgs_s *o) {
    if (o->twin_slab) return true;
    OptCore &c = o->core;
    const size_t padded = (size_t)((c.n + 63) / 64 * 64) * dtype_size(c.dtype);
    // the two vectors an odd number of KiB apart and off the 2-MiB grid the allocator hands out: x, g, d and the twins are
    // accessed at the same element offset at the same time, and equal offsets into equally aligned buffers hit the same
    // HBM channel
    const size_t slot = ((padded + 1023) / 1024 | 1) * 1024;
    if (hipMalloc(&o->twin_slab, 2 * slot + 16 * 1024) != hipSuccess) { (void)hipGetLastError(); o->twin_slab = nullptr; return false; }
    o->x_twin = (char *)o->twin_slab + 5 * 1024;
    o->g_twin = (char *)o->x_twin + slot;
    return true;
}

// can this step run as one pass over the history?  (built-in chained Rosenbrock, plain options)
static bool single_pass_ok(dzo_lbfgs_s *o) {
    OptCore &c = o->core;
    if (!o->single_pass || !o->blocked || o->mode != DZO_TWOLOOP_GRAM || o->line_search != 0 || o->descent_check || o->sd_fallback) return false;
    if (c.objective || c.gradient || c.constraint || c.box_on || !o->speculate || !o->fused_post) return false;
    if (c.iteration_count == 0 || o->k < 1 || o->k > kPairMaxK || o->m > kPairMaxK) return false;   // (m: the pass also forms the dots of pair k + 1)
    const int vecn = 16 / (int)dtype_size(c.dtype);
    // (a ragged n never gets here: its tile ring exists as a POINT ring only -- lbfgs_leave_points hands it to the slabs)
    if (c.n < 4 * vecn || (uint64_t)c.n * dtype_size(c.dtype) >= (1ull << 32)) return false;   // 32-bit byte offsets
    o->refresh_delta_ptrs();
    if (!problem_has_fused_post(c.problem, c.x, c.dx, c.g, c.dg) || !al16v(o->d)) return false;
    // the twin buffers of x and g: allocated here, before the step touches anything; a failed allocation
    // only switches this optimizer to the two-pass kernels
    if (!lbfgs_ensure_twins(o)) { o->single_pass = false; return false; }
    return true;
}

// the point ring serves exactly the optimizers the single-pass step serves (and, unlike it, the first step)
static bool points_ok(dzo_lbfgs_s *o) {
    OptCore &c = o->core;
    if (!o->points || !o->single_pass || !o->blocked || o->mode != DZO_TWOLOOP_GRAM || o->line_search != 0 || o->descent_check || o->sd_fallback) return false;
    if (c.objective || c.gradient || c.constraint || !o->speculate || !o->fused_post || !c.problem) return false;
    // (the decorators of legacy :219-296 ride on the pass: its DEC instantiations, under the set the ring was stored with)
    if (ring_obj_of(c.problem) != o->ring_obj || !(ring_decor_of(c.problem) == o->ring_dec)) return false;
    if (o->ring_obj >= 1 && (o->ring_dec.any() || c.problem->lambda != o->ring_obj_lambda)) return false;   // (no DEC instantiations of those objectives)
    if (o->ring_obj == 2 && c.problem->c != o->lse_c) return false;
    if (o->k > point_max_k(c.dtype) || o->m > point_max_k(c.dtype) || !al16v(o->d)) return false;
    if (o->k > 0 && !o->spec_scalars) return false;       // (the scalars come from the previous pass; anything else goes through Gram passes)
    return true;
}

// ---- aliasing of the caller's arrays (:393, :395) with twin buffers
static int32_t lbfgs_settle_entry(void *h);

static void lbfgs_mark_unsettled(dzo_lbfgs_s *o) {
    // (point ring, arrays filled for a read-only copy: the handle stays registered, so that the next dzo_synchronize /
    // dzo_memcpy_* -- after which the host may write into them -- still marks them as handed out)
    const bool dirty = o->core.x != o->x_user || o->core.g != o->g_user || (o->points && (o->xg_lin_stale || !o->xg_host_may_write));
    if (dirty && !o->unsettled) { unsettled_add(o, lbfgs_settle_entry); o->unsettled = true; }
    if (!dirty && o->unsettled) { unsettled_remove(o); o->unsettled = false; }
}

// current_point / current_gradient back into the arrays the optimizer aliases (a device copy each, only
// when they currently live in the twins)
static int32_t lbfgs_settle(dzo_lbfgs_s *o, bool hand_out = true) {
    std::lock_guard<std::recursive_mutex> lk(o->mu);
    OptCore &c = o->core;
    DZO_TRY(lbfgs_flush_post(o));                         // (the copy below may overwrite g_old: the caller's array may be that buffer)
    if (o->points) {
        DZO_TRY(lbfgs_points_settle(o, hand_out));
        lbfgs_mark_unsettled(o);
        return DZO_OK;
    }
    const size_t bytes = (size_t)c.n * dtype_size(c.dtype);
    if (c.x != o->x_user) {
        DZO_HIP(hipMemcpyAsync(o->x_user, c.x, bytes, hipMemcpyDeviceToDevice, c.stream));
        o->x_twin = c.x; c.x = o->x_user;
    }
    if (c.g != o->g_user) {
        DZO_HIP(hipMemcpyAsync(o->g_user, c.g, bytes, hipMemcpyDeviceToDevice, c.stream));
        o->g_twin = c.g; c.g = o->g_user;
    }
    lbfgs_mark_unsettled(o);
    return DZO_OK;
}

// called by dzo_synchronize / dzo_memcpy_* (the host is about to look at device memory)
static int32_t lbfgs_settle_entry(void *h) {
    dzo_lbfgs_s *o = static_cast<dzo_lbfgs_s *>(h);
    DeviceScope scope(o->device);
    DZO_TRY(lbfgs_settle(o));
    DZO_HIP(hipStreamSynchronize(o->core.stream));
    return DZO_OK;
}

template <typename T> static int32_t lbfgs_step_single_pass(dzo_lbfgs_s *o) {
    OptCore &c = o->core;
    hipStream_t s = c.stream;
    constexpr int N = Vec16<T>::N;
    const int k = o->k;
    const int64_t nvec = c.n / N;
    const int64_t rows = (nvec + kRowOwn - 1) / kRowOwn;
    DZO_TRY(gram_scalars<T>(o));                          // alpha / coef / scale of THIS step (:438-448 on scalars)
    o->scalars_ready = false;
    o->refresh_delta_ptrs();                              // deltas move to the spare slots
    FusedParams<T> fp;
    memset(&fp, 0, sizeof(fp));
    fp.n = c.n; fp.k = k; fp.k_next = k < o->m ? k + 1 : o->m; fp.t = (T)1;
    fp.x = (const T *)c.x; fp.g = (const T *)c.g; fp.x_out = (T *)o->x_twin; fp.g_out = (T *)o->g_twin;
    fp.d = (T *)o->d;
    fp.ring = (T *)o->S; fp.rowbytes = (uint32_t)o->rowbytes;
    fp.new_off = (uint32_t)((uint64_t)(2 * o->spare()) * (uint64_t)o->tile_stride); fp.ystride = (uint32_t)o->tile_stride;
    fp.alpha = o->alpha; fp.coef = o->coef; fp.scale = o->scale;
    for (int i = 0; i < kFusedMaxK; ++i) {                // entries >= k: any valid vector (zero coefficient)
        const int slot = o->slot_of(i < k ? i : 0);
        fp.soff[i] = (uint32_t)((uint64_t)(2 * slot) * (uint64_t)o->tile_stride);
    }
    fp.gram_partials = o->gram_partials;
    fp.obj_partials = c.problem->scratch;
    fp.changed = c.flag();
    fp.debug_skip = tune("DZO_TUNE_SP_DEBUG", 0);
    // register footprint follows the history length: pick the smallest instantiation that holds m pairs
    void (*kern)(FusedParams<T>) = o->m <= 8 ? lbfgs_single_pass_kernel<T, 8>
                                   : o->m <= 16 ? lbfgs_single_pass_kernel<T, 16>
                                   : lbfgs_single_pass_kernel<T, 20>;
    if constexpr (std::is_same<T, double>::value) {
        if ((fp.debug_skip & 256) && o->m > 16) kern = lbfgs_single_pass_kernel<double, 20, true>;
    }
    int64_t blocks = (rows + kWaves - 1) / kWaves;
    const int64_t res = (int64_t)ctx().cus * resident_blocks((const void *)kern);
    if (blocks > res) blocks = res;
    if (blocks > (int64_t)o->gram_grid * kWaves) blocks = (int64_t)o->gram_grid * kWaves;
    if (blocks > kMaxPartialBlocks) blocks = kMaxPartialBlocks;              // two objective partials per block in the problem scratch
    const int grid = (int)(blocks < 1 ? 1 : blocks);
    // The pass runs at t = 1 and, when that trial is rejected while the objective at t/2 (which rode along) is a
    // decrease, once more at t = 1/2 -- the loop of take_backtracking_step! (:121-152) on the same kernel, x and g
    // untouched in between.  Deeper halvings continue on the cheap trial kernels.
    const bool retry_pass = tune("DZO_TUNE_SP_RETRY", 1) != 0;
    DZO_REQUIRE(fused_params_ok<T>(fp, true), DZO_ERR_STATE, "single-pass step: a null operand (ring %p, x %p, g %p, twins %p %p, d %p)",
                (void *)fp.ring, (const void *)fp.x, (const void *)fp.g, (void *)fp.x_out, (void *)fp.g_out, (void *)fp.d);
    double t = 1.0;
    for (int attempt = 0;; ++attempt) {
    fp.t = (T)t; fp.t_half = (T)round_to_dtype(c.dtype, t * 0.5);
    if (!c.flag_armed) DZO_HIP(hipMemsetAsync(c.flag(), 0, sizeof(int32_t), s));
    c.flag_armed = false;
    {
        DZO_TIMED(attempt == 0 ? "lbfgs_single_pass" : "lbfgs_single_pass_retry", s);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), 0, s, fp);
    }
    {
        // :128 / :139 on the device (outcome to pinned host memory) as one block of the launch that also sums the
        // Gram partials; then, enqueued BEFORE the host knows the outcome and gated on that decision, the scalar
        // part of the NEXT two-loop (recurrence on the post-push history), so the GPU has work while the host
        // round-trips and the next step starts with its scalars ready.
        const int sv_newest = o->newest, sv_k = o->k;
        o->newest = o->spare(); o->k = fp.k_next;         // as lbfgs_finish_push will leave them
        double *vals = o->gram_partials + (size_t)kGramValues * kMaxHistory * o->gram_grid * kWaves;
        {
            DZO_TIMED("lbfgs_gram_reduce", s);
            const int nvals = kGramValues * o->k;
            DecideArgs da = decide_args(c, fp.obj_partials, grid, 1.0);
            da.partials2 = fp.obj_partials + grid;        // f(x + t/2 d)
            hipLaunchKernelGGL(gram_reduce_decide_kernel, dim3(nvals + 1), dim3(kBlock), 0, s, (const double *)o->gram_partials, grid, vals,
                               nvals, da);
            c.flag_armed = true;
        }
        DZO_HIP(hipGetLastError());
        int32_t rc = gram_finish_launch(o, 0, true, vals, true, c.status());
        o->newest = sv_newest; o->k = sv_k;
        DZO_TRY(rc);
    }
    DZO_TRY(core_wait_decision(c));
    // the host follows the DEVICE's decision (the gated kernels already acted on it): 0 reject, 1 accept, 2 stuck
    const int32_t status = reinterpret_cast<int32_t *>(c.host + 3)[0];
    if (attempt == 0) { c.last_trials = 0; o->single_pass_steps += 1; }
    if (status == 2) {                                    // :128-131 (x_new == x_old everywhere; x and g were never written)
        c.is_stuck = true;
        // the fields as take_backtracking_step! leaves them: delta_point = x_old (:118), delta_gradient
        // still the previous step's (= the newest pair of the ring; the pass wrote only spare slots)
        // (blocked ring: delta_gradient is gathered from the newest pair's tiles unless dg_lin still holds it)
        if (o->lin_stale && o->k > 0) { ring_gather<T>(o, o->y_slot_v(o->newest), o->dg_lin); o->lin_stale = false; }
        DZO_HIP(hipMemcpyAsync(c.dx, c.x, (size_t)c.n * sizeof(T), hipMemcpyDeviceToDevice, s));
        DZO_HIP(hipGetLastError());
        return DZO_OK;
    }
    c.last_trials += 1;
    const double f_new = round_to_dtype(c.dtype, c.host[0]);
    if (status == 1) {                                    // :139-146 (f_new < f), and the kernel already did :478-480
        c.df = round_to_dtype(c.dtype, f_new - c.f);
        c.f = f_new;
        DZO_TRY(lbfgs_finish_push(o, 0, true, true, true));   // rho of the new pair was set by the gated gram_finish; the pass wrote its tiles
        o->spec_scalars = true;                           // alpha_sp / coef_sp / scale_sp hold the next step's scalars
        o->gram_ready = false;
        o->gram_stale = 0;
        // the trial point and its gradient become the current ones: swap the buffers' roles
        std::swap(c.x, o->x_twin);
        std::swap(c.g, o->g_twin);
        lbfgs_mark_unsettled(o);
        return DZO_OK;
    }
    // rejected: x and g are untouched
    if (attempt == 0) o->single_pass_rejections += 1;
    const double f_half = round_to_dtype(c.dtype, c.host[1]);
    if (attempt == 0 && retry_pass && f_half < c.f) {     // :152, and the trial at t/2 will pass :139
        t = round_to_dtype(c.dtype, t * 0.5);
        o->single_pass_retries += 1;
        continue;
    }
    // the reference's loop continues on the two-pass kernels (whose first trial saves x_old in delta_point, :118):
    // after this trial, and after the one at t/2 as well when its objective is already known to be no decrease
    if (attempt == 0 && retry_pass) { c.last_trials += 1; return lbfgs_search_and_post(o, 2); }
    return lbfgs_search_and_post(o, attempt + 1);
    }
}

// step! on the point ring: every trial of the step is one lbfgs_point_pass_kernel launch (t = 1, then the halvings
// of :152 -- skipping a t/2 whose objective, carried by the previous pass, is already known to be no decrease),
// x and g (= point 0) untouched until a trial is accepted, which makes the spare slot point 0.
// grid of a point pass: the resident blocks, bounded by the partial-sum buffers
template <typename T> static int points_grid(dzo_lbfgs_s *o, void (*kern)(FusedParams<T>), size_t dyn_lds = 0) {
    const int64_t nvec = ring_nvec<T>(o);
    const int64_t rows = (nvec + kRowOwn - 1) / kRowOwn;
    int64_t blocks = (rows + kWaves - 1) / kWaves;
    const int64_t res = (int64_t)ctx().cus * resident_blocks((const void *)kern, dyn_lds);
    if (blocks > res) blocks = res;
    if (blocks > (int64_t)o->gram_grid * kWaves) blocks = (int64_t)o->gram_grid * kWaves;
    if (blocks > kMaxPartialBlocks / 2) blocks = kMaxPartialBlocks / 2;     // up to four partial sums per block in the problem scratch (2 kMaxPartialBlocks doubles)
    return (int)(blocks < 1 ? 1 : blocks);
}

// one register set per wave (two waves per SIMD)?  DZO_TUNE_POINT_SETS=1 (the default), where the instantiation fits 256 registers
template <typename T> static bool point_one_set(const dzo_lbfgs_s *o) { return o->point_sets == 1 && o->m <= 20 && (sizeof(T) == 8 || o->m <= 12); }

// the instantiation of the point pass for this optimizer: the smallest K that holds m pairs; one or two register sets
// (DZO_TUNE_POINT_SETS; see the kernel)
template <typename T> static void (*point_pass_kernel_sel(dzo_lbfgs_s *o))(FusedParams<T>) {
    if (o->ring_obj == 1) {
        // the chained quadratic (ChainObj<T, 1>): the same ladder of history lengths as the decorated pass
        if (point_one_set<T>(o)) {
            if constexpr (sizeof(T) == 8) {
                return o->m <= 8 ? lbfgs_point_pass_kernel<T, 8, false, 1, false, 1>
                       : o->m <= 12 ? lbfgs_point_pass_kernel<T, 12, false, 1, false, 1>
                       : o->m <= 16 ? lbfgs_point_pass_kernel<T, 16, false, 1, false, 1>
                       : lbfgs_point_pass_kernel<T, 20, false, 1, false, 1>;
            } else {
                return o->m <= 8 ? lbfgs_point_pass_kernel<T, 8, false, 1, false, 1> : lbfgs_point_pass_kernel<T, 12, false, 1, false, 1>;
            }
        }
        if constexpr (sizeof(T) == 8) { if (o->m > 20) return lbfgs_point_pass_kernel<T, 24, false, 2, false, 1>; }
        return o->m <= 12 ? lbfgs_point_pass_kernel<T, 12, false, 2, false, 1>
               : o->m <= 16 ? lbfgs_point_pass_kernel<T, 16, false, 2, false, 1>
               : lbfgs_point_pass_kernel<T, 20, false, 2, false, 1>;
    }
    if (o->ring_dec.any()) {
        // the decorated pass (DEC): fewer instantiations, the next larger K serves the history lengths in between
        if (point_one_set<T>(o)) {
            if constexpr (sizeof(T) == 8) {
                return o->m <= 8 ? lbfgs_point_pass_kernel<T, 8, false, 1, true>
                       : o->m <= 12 ? lbfgs_point_pass_kernel<T, 12, false, 1, true>
                       : o->m <= 16 ? lbfgs_point_pass_kernel<T, 16, false, 1, true>
                       : lbfgs_point_pass_kernel<T, 20, false, 1, true>;
            } else {
                return o->m <= 8 ? lbfgs_point_pass_kernel<T, 8, false, 1, true> : lbfgs_point_pass_kernel<T, 12, false, 1, true>;
            }
        }
        if constexpr (sizeof(T) == 8) { if (o->m > 20) return lbfgs_point_pass_kernel<T, 24, false, 2, true>; }
        return o->m <= 12 ? lbfgs_point_pass_kernel<T, 12, false, 2, true>
               : o->m <= 16 ? lbfgs_point_pass_kernel<T, 16, false, 2, true>
               : lbfgs_point_pass_kernel<T, 20, false, 2, true>;
    }
    if (point_one_set<T>(o)) {
        if constexpr (sizeof(T) == 8) {
            // (K = 10: m = 10 is the history length most L-BFGS users ask for; on the K = 12 instantiation it paid for two
            // masked pairs -- 211 us per pass at n = 1e7 where m = 12 takes 218)
            return o->m <= 6 ? lbfgs_point_pass_kernel<T, 6, false, 1>
                   : o->m <= 8 ? lbfgs_point_pass_kernel<T, 8, false, 1>
                   : o->m <= 10 ? lbfgs_point_pass_kernel<T, 10, false, 1>
                   : o->m <= 12 ? lbfgs_point_pass_kernel<T, 12, false, 1>
                   : o->m <= 14 ? lbfgs_point_pass_kernel<T, 14, false, 1>
                   : o->m <= 16 ? lbfgs_point_pass_kernel<T, 16, false, 1>
                   : o->m <= 18 ? lbfgs_point_pass_kernel<T, 18, false, 1>
                   : lbfgs_point_pass_kernel<T, 20, false, 1>;
        } else {                                             // (fp32, K > 12: the fp64 copies for the dots do not fit 256 registers)
            return o->m <= 6 ? lbfgs_point_pass_kernel<T, 6, false, 1>
                   : o->m <= 8 ? lbfgs_point_pass_kernel<T, 8, false, 1>
                   : o->m <= 10 ? lbfgs_point_pass_kernel<T, 10, false, 1> : lbfgs_point_pass_kernel<T, 12, false, 1>;
        }
    }
    // (m <= 24: point_max_k; K = 22 for m = 21, 22: on K = 24 they paid for two or three masked pairs)
    if constexpr (sizeof(T) == 8) { if (o->m > 20) return o->m <= 22 ? lbfgs_point_pass_kernel<T, 22, false, 2> : lbfgs_point_pass_kernel<T, 24, false, 2>; }
    return o->m <= 8 ? lbfgs_point_pass_kernel<T, 8, false, 2>
           : o->m <= 12 ? lbfgs_point_pass_kernel<T, 12, false, 2>
           : o->m <= 16 ? lbfgs_point_pass_kernel<T, 16, false, 2>
           : lbfgs_point_pass_kernel<T, 20, false, 2>;
}
// first: the first step's kernel
template <typename T> static void (*point_pass_kernel_for(dzo_lbfgs_s *o, bool first))(FusedParams<T>) {
    if (first) return o->ring_obj == 1 ? lbfgs_point_pass_kernel<T, 8, true, 2, false, 1>
                      : o->ring_dec.any() ? lbfgs_point_pass_kernel<T, 8, true, 2, true> : lbfgs_point_pass_kernel<T, 8, true>;
    return point_pass_kernel_sel<T>(o);
}

// step_direction of the last step, on demand: the same pass once more over the view of the ring that step started
// from (its points are all still there: the push only rotated the ring, and the slot the next pass will overwrite is
// that view's oldest point), with the scalars that step used, writing d and nothing else.
template <typename T> static int32_t lbfgs_materialize_d_t(dzo_lbfgs_s *o) {
    OptCore &c = o->core;
    hipStream_t s = c.stream;
    const int k = o->dview_k;
    FusedParams<T> fp;
    memset(&fp, 0, sizeof(fp));
    fp.n = c.n; fp.k = k; fp.k_next = 1; fp.t = (T)1; fp.t_half = (T)0.5;
    fp.d = (T *)o->d; fp.store_d = 1;
    fp.ring = (T *)o->S; fp.rowbytes = (uint32_t)o->rowbytes;
    fp.new_off = (uint32_t)((uint64_t)(2 * o->spare()) * (uint64_t)o->tile_stride); fp.ystride = (uint32_t)o->tile_stride;   // (never written: no tile stores in this mode)
    fp.alpha = o->alpha; fp.coef = o->coef; fp.scale = o->scale;
    auto old_slot = [&](int j) { return ((o->dview_newest - j) % o->nslots + o->nslots) % o->nslots; };
    for (int j = 0; j <= kFusedMaxK; ++j) fp.soff[j] = (uint32_t)((uint64_t)(2 * old_slot(j < k ? j : k)) * (uint64_t)o->tile_stride);
    fp.gram_partials = o->gram_partials;                  // (consumed by the step's gated gram_finish; scratch here)
    fp.obj_partials = c.problem->scratch;
    fp.changed = c.flag();
    fp.debug_skip = 1 | 64;                               // no pair dots, no tile stores (and with them no halo copies)
    fp.dec = point_decor<T>(o->ring_dec);
    fp.obj_lambda = (T)o->ring_obj_lambda;
    fp.stage_rows = 1;
    void (*kern)(FusedParams<T>) = point_pass_kernel_for<T>(o, false);
    const int grid = points_grid<T>(o, kern);
    DZO_REQUIRE(fused_params_ok<T>(fp, false), DZO_ERR_STATE, "direction on demand: a null operand (ring %p, d %p)", (void *)fp.ring, (void *)fp.d);
    {
        DZO_TIMED("lbfgs_direction_on_demand", s);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), 0, s, fp);
    }
    DZO_HIP(hipMemsetAsync(c.flag(), 0, sizeof(int32_t), s));                 // (the pass raised the change flag)
    c.flag_armed = true;
    DZO_HIP(hipGetLastError());
    o->d_stale = false;
    return DZO_OK;
}
template <typename T> static int32_t lbfgs_materialize_d_lse(dzo_lbfgs_s *o);
static int32_t lbfgs_materialize_d(dzo_lbfgs_s *o) {
    if (!o->points || !o->d_stale) return DZO_OK;
    std::lock_guard<std::recursive_mutex> lk(o->mu);
    if (o->ring_obj == 2) { DZO_DISPATCH(o->core.dtype, return lbfgs_materialize_d_lse<T>(o)); }
    DZO_DISPATCH(o->core.dtype, return lbfgs_materialize_d_t<T>(o));
}

template <typename T> static int32_t lbfgs_step_points(dzo_lbfgs_s *o) {
    OptCore &c = o->core;
    hipStream_t s = c.stream;
    const int k = o->k;
    if (k > 0) {
        DZO_TRY(gram_scalars<T>(o));                      // alpha / coef / scale of THIS step (computed behind the last decision)
        o->scalars_ready = false;
    }                                                     // (:463: the first step walks along the step_direction the constructor left)
    FusedParams<T> fp;
    memset(&fp, 0, sizeof(fp));
    fp.n = c.n; fp.k = k; fp.k_next = k < o->m ? k + 1 : o->m;
    fp.d = (T *)o->d;
    fp.ring = (T *)o->S; fp.rowbytes = (uint32_t)o->rowbytes;
    fp.new_off = (uint32_t)((uint64_t)(2 * o->spare()) * (uint64_t)o->tile_stride); fp.ystride = (uint32_t)o->tile_stride;
    fp.alpha = o->alpha; fp.coef = o->coef; fp.scale = o->scale;
    for (int j = 0; j <= kFusedMaxK; ++j)                 // points beyond k: point k again (those pairs are zero)
        fp.soff[j] = (uint32_t)((uint64_t)(2 * o->slot_of(j < k ? j : k)) * (uint64_t)o->tile_stride);
    fp.gram_partials = o->gram_partials;
    fp.obj_partials = c.problem->scratch;
    fp.changed = c.flag();
    fp.debug_skip = tune("DZO_TUNE_SP_DEBUG", 0);
    fp.dec = point_decor<T>(o->ring_dec);
    fp.obj_lambda = (T)o->ring_obj_lambda;
    void (*kern)(FusedParams<T>) = point_pass_kernel_for<T>(o, k == 0);
    fp.store_d = o->lazy_d ? 0 : 1;
    {
        // tile-major ring: plain stores while the two streams fit the 256-MiB Infinity Cache (695 vs 735 us at
        // n = 1e7); stream-major ring: non-temporal (655-672 vs 659-682 us over four interleaved rounds)
        const int64_t cache_mb = tune("DZO_TUNE_POINT_PLAIN_MB", o->tile_stride == kTileBytes ? 200 : 0);
        fp.nt_tiles = 2 * (int64_t)c.n * (int64_t)sizeof(T) > (cache_mb << 20) ? 1 : 0;
    }
    // rows of new tiles a wave collects in LDS before it writes them (2 KiB per row and wave; the whole 160-KiB LDS
    // of a CU is this one block's)
    // (two blocks per CU with one register set per wave: half the LDS each)
    const bool one_set = point_one_set<T>(o) && k > 0;
    fp.prio = one_set ? (tune("DZO_TUNE_POINT_PRIO", 1) != 0 ? 1 : 0) : 0;   // (two waves per SIMD only; see the kernel)
    fp.leftover_even = tune("DZO_TUNE_POINT_LEFTOVER_EVEN", 1) != 0 ? 1 : 0;
    const int stage_tiles = (k == 0 || DZO_PP_REGRAD == 0) ? 2 : 1;      // tiles staged per row: the point, and its gradient where the kernel writes it
    const int stage_max = (one_set ? 72 : 144) / (4 * stage_tiles);        // KiB of LDS per block / (waves x KiB per staged row)
    fp.stage_rows = tune("DZO_TUNE_POINT_STAGE_ROWS", 16);
    if (fp.stage_rows < 1) fp.stage_rows = 1;
    if (fp.stage_rows > stage_max) fp.stage_rows = stage_max;
    size_t stage_bytes = (size_t)kWaves * fp.stage_rows * stage_tiles * kTileBytes;
    if (o->stage_kern != (const void *)kern || o->stage_bytes != stage_bytes) {   // (once per kernel and size)
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)stage_bytes) != hipSuccess) {
            (void)hipGetLastError();                      // a device that does not grant it: stay within the default 64 KiB
            o->stage_small = true;
        }
        o->stage_kern = (const void *)kern; o->stage_bytes = stage_bytes;
    }
    if (o->stage_small && fp.stage_rows > 14 / stage_tiles) {
        fp.stage_rows = 14 / stage_tiles;
        stage_bytes = (size_t)kWaves * fp.stage_rows * stage_tiles * kTileBytes;
    }
    // the blocks resident at once WITH the dynamic LDS of this launch: asked without it, the query answers 3 per CU for the
    // K = 8 instantiation (149 registers) while 64 KiB of staging let two in, and the third block of every CU ran as a
    // second round behind the others (n = 1e7, m = 5: 172 -> 144 us per pass, 4400 -> 5500 step!()/s)
    const int grid = points_grid<T>(o, kern, stage_bytes);
    const int pgrid = grid;                               // columns of per-block partial sums
    DZO_REQUIRE(fused_params_ok<T>(fp, false), DZO_ERR_STATE, "point pass: a null operand (ring %p, d %p, scalars %p %p %p)",
                (void *)fp.ring, (void *)fp.d, (const void *)fp.alpha, (const void *)fp.coef, (const void *)fp.scale);
    o->d_stale = false;                                   // (whatever was pending belonged to the previous step)
    const int view_k = k, view_newest = o->newest;
    auto direction_pending = [&]() { if (k > 0 && o->lazy_d) { o->d_stale = true; o->dview_k = view_k; o->dview_newest = view_newest; } };
    c.last_trials = 0;
    o->single_pass_steps += 1;
    int64_t halvings = 0;
    double t = 1.0;
    for (int attempt = 0;; ++attempt) {
        fp.t = (T)t; fp.t_half = (T)round_to_dtype(c.dtype, t * 0.5);
        if (!c.flag_armed) DZO_HIP(hipMemsetAsync(c.flag(), 0, sizeof(int32_t), s));
        c.flag_armed = false;
        {
            DZO_TIMED(attempt == 0 ? "lbfgs_single_pass" : "lbfgs_single_pass_retry", s);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), stage_bytes, s, fp);
        }
        {
            // decision + Gram reduction in one launch, then the gated recurrence for the next step (as
            // lbfgs_step_single_pass)
            const int sv_newest = o->newest, sv_k = o->k;
            o->newest = o->spare(); o->k = fp.k_next;
            double *vals = o->gram_partials + (size_t)kGramValues * kMaxHistory * o->gram_grid * kWaves;
            {
                DZO_TIMED("lbfgs_gram_reduce", s);
                const int nvals = kGramValues * o->k;
                DecideArgs da = decide_args(c, fp.obj_partials, pgrid, 1.0);
                da.partials2 = fp.obj_partials + pgrid;   // f(x + t/2 d)
                if (o->ring_dec.l2 != 0.0) {              // :232  + lambda * norm2(x), both candidates
                    da.l2_partials = fp.obj_partials + 2 * pgrid; da.l2_partials2 = fp.obj_partials + 3 * pgrid;
                    da.l2_lambda = o->ring_dec.l2;
                }
                hipLaunchKernelGGL(gram_reduce_decide_kernel, dim3(nvals + 1), dim3(kBlock), 0, s, (const double *)o->gram_partials, pgrid, vals,
                                   nvals, da);
                c.flag_armed = true;
            }
            DZO_HIP(hipGetLastError());
            int32_t rc = gram_finish_launch(o, 0, true, vals, true, c.status());
            o->newest = sv_newest; o->k = sv_k;
            DZO_TRY(rc);
        }
        DZO_TRY(core_wait_decision(c));
        const int32_t status = reinterpret_cast<int32_t *>(c.host + 3)[0];
        if (status == 2) {                                // :128-131 (x_new == x_old everywhere)
            c.is_stuck = true;
            direction_pending();
            // delta_point = x_old (:118), delta_gradient still the previous step's
            if (o->k > 0) {
                DZO_TRY(lbfgs_ensure_g(o, o->slot_of(0))); DZO_TRY(lbfgs_ensure_g(o, o->slot_of(1)));
                ring_gather_diff<T>(o, o->y_slot_v(o->slot_of(0)), o->y_slot_v(o->slot_of(1)), o->dg_lin);
            }
            ring_gather<T>(o, o->s_slot_v(o->newest), o->dx_lin);
            DZO_HIP(hipGetLastError());
            o->lin_stale = false;
            return DZO_OK;
        }
        c.last_trials += 1;
        const double f_new = round_to_dtype(c.dtype, c.host[0]);
        if (status == 1) {                                // :139-146, and the pass already did :478-480
            c.df = round_to_dtype(c.dtype, f_new - c.f);
            c.f = f_new;
            {
                // the spare slot becomes point 0: its gradient tiles are the new gradient only if this launch wrote them
                // (the first step's kernel and builds that stream the gradients do; the recomputing pass does not)
                const bool wrote_g = (k == 0) || DZO_PP_REGRAD == 0;
                if (wrote_g) o->g_valid |= 1u << o->spare(); else o->g_valid &= ~(1u << o->spare());
            }
            DZO_TRY(lbfgs_finish_push(o, 0, true, true, true));   // rho by the gated gram_finish; the pass wrote the new point's tiles
            o->spec_scalars = true;
            o->gram_ready = false;
            o->gram_stale = 0;
            o->xg_lin_stale = true;
            direction_pending();
            lbfgs_mark_unsettled(o);
            return DZO_OK;
        }
        // rejected (:151-152)
        if (attempt == 0) o->single_pass_rejections += 1;
        auto halve = [&]() -> bool {                      // false: the halving limit ended the search
            t = round_to_dtype(c.dtype, t * 0.5);
            return !(c.max_halvings > 0 && ++halvings >= c.max_halvings);
        };
        bool go_on = halve();
        const double f_half = round_to_dtype(c.dtype, c.host[1]);
        if (go_on && !(f_half < c.f)) {                   // the trial at t/2 would fail :139 as well: count it and move on
            // (its change flag: x + (t/2) d == x only when x + t d == x, which the pass has just ruled out, up to the
            // last halvings before the limit -- there the next pass reports status 2 itself)
            c.last_trials += 1;
            go_on = halve();
        }
        if (!go_on) {                                     // build-added escape from the NaN loop
            c.is_stuck = true;
            direction_pending();
            ring_gather<T>(o, o->s_slot_v(o->newest), o->dx_lin);
            if (o->k > 0) {
                DZO_TRY(lbfgs_ensure_g(o, o->slot_of(0))); DZO_TRY(lbfgs_ensure_g(o, o->slot_of(1)));
                ring_gather_diff<T>(o, o->y_slot_v(o->slot_of(0)), o->y_slot_v(o->slot_of(1)), o->dg_lin);
            }
            DZO_HIP(hipGetLastError());
            o->lin_stale = false;
            return DZO_OK;
        }
        o->single_pass_retries += 1;
    }
}

// ---- step! on the point ring for the log-sum-exp objective (lse_trial_kernel / lse_decide_kernel / lse_dots_kernel)
template <typename T> static void lse_fill_params(dzo_lbfgs_s *o, LseParams<T> &lp, int k, int view_newest) {
    OptCore &c = o->core;
    memset(&lp, 0, sizeof(lp));
    auto slot_at = [&](int j) { return ((view_newest - j) % o->nslots + o->nslots) % o->nslots; };
    lp.n = c.n; lp.k = k; lp.k_next = k < o->m ? k + 1 : o->m;
    lp.ring = (T *)o->S; lp.rowbytes = (uint32_t)o->rowbytes;
    for (int j = 0; j <= kLseMaxK; ++j) {
        const int sl = slot_at(j < k ? j : k);
        lp.soff[j] = (uint32_t)((uint64_t)(2 * sl) * (uint64_t)o->tile_stride);
        lp.slot[j] = (uint8_t)sl;
    }
    lp.c_off = (uint32_t)((uint64_t)(2 * o->nslots) * (uint64_t)o->tile_stride);
    lp.pscal = o->pscal; lp.lambda = o->ring_obj_lambda;
    lp.alpha = o->alpha; lp.coef = o->coef; lp.scale = o->scale;
    lp.d = (T *)o->d;
    lp.partials = c.problem->scratch;
    lp.changed = c.flag();
    lp.gram_partials = o->gram_partials;
}
template <typename T> static void (*lse_trial_kernel_for(dzo_lbfgs_s *o, bool first))(LseParams<T>) {
    (void)o;
    return first ? lse_trial_kernel<T, true> : lse_trial_kernel<T, false>;
}
template <typename T> static void (*lse_dots_kernel_for(dzo_lbfgs_s *o))(LseParams<T>) {
    if constexpr (sizeof(T) == 8) { if (o->m > 20) return lse_dots_kernel<T, 24>; }
    return o->m <= 8 ? lse_dots_kernel<T, 8> : o->m <= 12 ? lse_dots_kernel<T, 12> : lse_dots_kernel<T, 20>;
}
template <typename T> static int lse_grid(dzo_lbfgs_s *o, const void *kern) {
    const int64_t rows = o->ring_rows;
    int64_t blocks = (rows + kWaves - 1) / kWaves;
    const int64_t res = (int64_t)ctx().cus * resident_blocks(kern);
    if (blocks > res) blocks = res;
    if (blocks > (int64_t)o->gram_grid * kWaves) blocks = (int64_t)o->gram_grid * kWaves;
    if (blocks > kMaxPartialBlocks / 2) blocks = kMaxPartialBlocks / 2;
    return (int)(blocks < 1 ? 1 : blocks);
}

// step_direction of the last step on demand (as lbfgs_materialize_d_t): the trial pass over the view that step started from,
// writing d and nothing else
template <typename T> static int32_t lbfgs_materialize_d_lse(dzo_lbfgs_s *o) {
    OptCore &c = o->core;
    LseParams<T> lp;
    lse_fill_params<T>(o, lp, o->dview_k, o->dview_newest);
    lp.t = (T)1; lp.store_d = 1; lp.store_tile = 0;
    lp.new_off = (uint32_t)((uint64_t)(2 * o->spare()) * (uint64_t)o->tile_stride); lp.new_slot = (uint8_t)o->spare();
    auto kern = lse_trial_kernel_for<T>(o, false);
    const int grid = lse_grid<T>(o, (const void *)kern);
    {
        DZO_TIMED("lbfgs_direction_on_demand", c.stream);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), 0, c.stream, lp);
    }
    DZO_HIP(hipMemsetAsync(c.flag(), 0, sizeof(int32_t), c.stream));
    c.flag_armed = true;
    DZO_HIP(hipGetLastError());
    o->d_stale = false;
    return DZO_OK;
}

template <typename T> static int32_t lbfgs_step_points_lse(dzo_lbfgs_s *o) {
    OptCore &c = o->core;
    hipStream_t s = c.stream;
    const int k = o->k;
    if (k > 0) {
        DZO_TRY(gram_scalars<T>(o));                      // alpha / coef / scale of THIS step (left by the previous step's dots pass)
        o->scalars_ready = false;
    }
    LseParams<T> lp;
    lse_fill_params<T>(o, lp, k, o->newest);
    lp.new_off = (uint32_t)((uint64_t)(2 * o->spare()) * (uint64_t)o->tile_stride); lp.new_slot = (uint8_t)o->spare();
    lp.store_d = (k > 0 && !o->lazy_d) ? 1 : 0; lp.store_tile = 1;
    auto kern = lse_trial_kernel_for<T>(o, k == 0);
    const int grid = lse_grid<T>(o, (const void *)kern);
    o->d_stale = false;
    const int view_k = k, view_newest = o->newest;
    auto direction_pending = [&]() { if (k > 0 && o->lazy_d) { o->d_stale = true; o->dview_k = view_k; o->dview_newest = view_newest; } };
    auto stuck_fields = [&]() -> int32_t {                // delta_point = x_old (:118), delta_gradient still the previous step's
        if (o->k > 0) {
            DZO_TRY(lbfgs_ensure_g(o, o->slot_of(0))); DZO_TRY(lbfgs_ensure_g(o, o->slot_of(1)));
            ring_gather_diff<T>(o, o->y_slot_v(o->slot_of(0)), o->y_slot_v(o->slot_of(1)), o->dg_lin);
        }
        ring_gather<T>(o, o->s_slot_v(o->newest), o->dx_lin);
        DZO_HIP(hipGetLastError());
        o->lin_stale = false;
        return DZO_OK;
    };
    c.last_trials = 0;
    o->single_pass_steps += 1;
    int64_t halvings = 0;
    double t = 1.0;
    for (int attempt = 0;; ++attempt) {
        lp.t = (T)t;
        if (!c.flag_armed) DZO_HIP(hipMemsetAsync(c.flag(), 0, sizeof(int32_t), s));
        c.flag_armed = false;
        {
            DZO_TIMED(attempt == 0 ? "lbfgs_single_pass" : "lbfgs_single_pass_retry", s);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), 0, s, lp);
        }
        {
            DZO_TIMED("lbfgs_lse_decide", s);
            DecideArgs da = decide_args(c, nullptr, 0, 1.0);
            hipLaunchKernelGGL(lse_decide_kernel, dim3(1), dim3(kBlock), 0, s, (const double *)lp.partials, grid, o->pscal, (int)lp.slot[0], (int)lp.new_slot,
                               o->ring_obj_lambda, c.dtype == DZO_F32 ? 1 : 0, da);
            c.flag_armed = true;
        }
        DZO_HIP(hipGetLastError());
        DZO_TRY(core_wait_decision(c));
        const int32_t status = reinterpret_cast<int32_t *>(c.host + 3)[0];
        if (status == 2) {                                // :128-131
            c.is_stuck = true;
            direction_pending();
            return stuck_fields();
        }
        c.last_trials += 1;
        if (status == 1) {                                // :139-146; then the dots of the next two-loop over the k + 2 points
            const double f_new = round_to_dtype(c.dtype, c.host[0]);
            c.df = round_to_dtype(c.dtype, f_new - c.f);
            c.f = f_new;
            auto dk = lse_dots_kernel_for<T>(o);
            const int dgrid = lse_grid<T>(o, (const void *)dk);
            {
                DZO_TIMED("lbfgs_lse_dots", s);
                hipLaunchKernelGGL(dk, dim3(dgrid), dim3(kBlock), 0, s, lp);
            }
            const int sv_newest = o->newest, sv_k = o->k;
            o->newest = o->spare(); o->k = lp.k_next;     // as lbfgs_finish_push will leave them
            double *vals = o->gram_partials + (size_t)kGramValues * kMaxHistory * o->gram_grid * kWaves;
            {
                DZO_TIMED("lbfgs_gram_reduce", s);
                hipLaunchKernelGGL(gram_reduce_kernel, dim3(kGramValues * o->k), dim3(kBlock), 0, s, o->gram_partials, dgrid, vals, kGramValues * o->k,
                                   (const double *)nullptr, 0, (double *)nullptr, 0);
            }
            DZO_HIP(hipGetLastError());
            int32_t rc = gram_finish_launch(o, 0, true, vals, true, c.status());   // (status is 1: the gate only selects the *_sp scalar set)
            o->newest = sv_newest; o->k = sv_k;
            DZO_TRY(rc);
            o->g_valid &= ~(1u << o->spare());            // (the trial pass writes no gradient tiles: formed on demand)
            DZO_TRY(lbfgs_finish_push(o, 0, true, true, true));
            o->spec_scalars = true;
            o->gram_ready = false;
            o->gram_stale = 0;
            o->xg_lin_stale = true;
            direction_pending();
            lbfgs_mark_unsettled(o);
            return DZO_OK;
        }
        if (attempt == 0) o->single_pass_rejections += 1;
        t = round_to_dtype(c.dtype, t * 0.5);             // :151-152
        if (c.max_halvings > 0 && ++halvings >= c.max_halvings) {
            c.is_stuck = true;
            direction_pending();
            return stuck_fields();
        }
        o->single_pass_retries += 1;
    }
}

// Point ring, before a step: if the host changed current_point / current_gradient since they were last gathered
// (dzo_memcpy_*, its own kernels through the pointers of get_ptr, the arrays it passed to the constructor), the
// step must start from the caller's values (:393 aliasing).  The ring keeps its own point 0 for the pairs --
// exactly what the reference's stored delta_point_history is: unaffected by such a write -- and turns into the
// pair ring; the run continues on the pair kernels with x / g = the caller's arrays.
static int32_t lbfgs_adopt_host_writes(dzo_lbfgs_s *o) {
    if (!o->points || !o->xg_host_may_write) return DZO_OK;
    OptCore &c = o->core;
    o->xg_host_may_write = false;
    if (o->xg_lin_stale) return DZO_OK;                   // (the arrays do not hold a gathered copy at all)
    DZO_TRY(lbfgs_ensure_g(o, o->newest));
    o->host_write_checks += 1;
    DZO_HIP(hipMemsetAsync(o->xg_differs, 0, sizeof(int32_t), c.stream));
    DZO_DISPATCH(c.dtype, (ring_compare<T>(o, o->s_slot_v(o->newest), o->x_user), ring_compare<T>(o, o->y_slot_v(o->newest), o->g_user)));
    DZO_HIP(hipGetLastError());
    int32_t differs = 0;
    DZO_HIP(hipMemcpyAsync(&differs, o->xg_differs, sizeof(int32_t), hipMemcpyDeviceToHost, c.stream));
    DZO_HIP(hipStreamSynchronize(c.stream));
    if (!differs) return DZO_OK;
    DZO_TRY(lbfgs_leave_points(o));                       // (gathers nothing: the arrays are "fresh")
    // the scalars computed behind the last decision belong to the old gradient
    o->spec_scalars = false; o->gram_ready = false; o->scalars_ready = false; o->gram_rebuild = true;
    return DZO_OK;
}

static int32_t lbfgs_step(dzo_lbfgs_s *o) {
    OptCore &c = o->core;
    if (c.is_stuck) return DZO_OK;                        // :456-458
    std::lock_guard<std::recursive_mutex> step_lock(o->mu);
    DZO_REQUIRE(c.has_objective() && c.has_gradient(), DZO_ERR_STATE,
                "step! needs objective and gradient (callbacks or a built-in problem)");
    if (c.problem && c.problem->parent) {                 // decorators may have been changed on the user's handle
        problem_view_sync(c.problem);
        c.box_on = c.problem->cons_on; c.box_lo = c.problem->cons_lo; c.box_hi = c.problem->cons_hi;
    }
    bool quasi = false;
    o->last_step_kind = 0;
    DZO_TRY(lbfgs_adopt_host_writes(o));
    if (o->points) {
        if (points_ok(o)) {
            if (o->ring_obj == 2) { DZO_DISPATCH(c.dtype, return lbfgs_step_points_lse<T>(o)); }
            DZO_DISPATCH(c.dtype, return lbfgs_step_points<T>(o));
        }
        DZO_TRY(lbfgs_leave_points(o));                   // an option the passes do not serve: continue on the pair ring
    }
    if (single_pass_ok(o)) { DZO_DISPATCH(c.dtype, return lbfgs_step_single_pass<T>(o)); }
    if (c.iteration_count > 0) {
        DZO_TRY(lbfgs_direction(o));                      // :463-471
        quasi = o->k > 0;
        if (o->descent_check) {                           // legacy :682-692
            double gd = 0;
            DZO_TRY(lbfgs_aux_dot(o, o->d, c.g, &gd));
            if (!std::isfinite(gd)) { c.is_stuck = true; return DZO_OK; }
            if (gd >= 0.0) {
                DZO_TRY(lbfgs_steepest(o));
                o->descent_resets += 1;
                o->last_step_kind = 1;
                quasi = false;
            }
        }
    }
    DZO_TRY(lbfgs_search_and_post(o));
    if (c.is_stuck && o->sd_fallback && quasi) {          // legacy :588-610
        c.is_stuck = false;
        DZO_TRY(lbfgs_steepest(o));
        o->last_step_kind = 2;
        o->reset_on_push = true;
        DZO_TRY(lbfgs_search_and_post(o));
        o->reset_on_push = false;                         // (stays set only if the retry failed too)
    }
    return DZO_OK;
}

}  // namespace dzo

using namespace dzo;

static thread_local bool tl_want_blocked = false;       // dzo_lbfgs_create_problem -> dzo_lbfgs_create
static thread_local RingDecor tl_ring_dec;              // ... and the decorators the start point's gradient was formed under
static thread_local int tl_ring_obj = 0;                // ... and the chained objective (ChainObj) with its parameter
static thread_local double tl_ring_obj_lambda = 0;

// ============================================================================ C ABI
extern "C" {

int32_t dzo_lbfgs_create(int64_t n, int32_t history_length, int32_t dtype, void *x_dev, void *g_dev,
                         double initial_objective_value, double initial_step_length, dzo_lbfgs_t *out) {
    DZO_TRY(require_init());
    DZO_REQUIRE(out, DZO_ERR_INVALID, "null out");
    DZO_REQUIRE(dtype == DZO_F32 || dtype == DZO_F64, DZO_ERR_INVALID, "bad dtype %d", dtype);
    DZO_REQUIRE(n >= 1 && x_dev && g_dev, DZO_ERR_INVALID, "n must be >= 1 and x/g non-null");
    DZO_REQUIRE(history_length >= 1 && history_length <= kMaxHistory, DZO_ERR_INVALID,
                "history_length must be in 1..%d", kMaxHistory);
    DZO_REQUIRE(initial_step_length > 0, DZO_ERR_ASSERT, "@assert initial_step_length > 0 (src/DZOptimization.jl:380)");
    // :363-364 (and :366-378: everything `similar` allocates below lands on the same device by construction)
    DZO_TRY(require_same_backend("LBFGSOptimizer", "src/DZOptimization.jl:363-364", x_dev, "initial_point", g_dev, "initial_gradient"));
    dzo_lbfgs_s *o = new dzo_lbfgs_s();
    OptCore &c = o->core;
    c.n = n; c.dtype = dtype; c.x = x_dev; c.g = g_dev;
    o->x_user = x_dev; o->g_user = g_dev; o->device = ctx().device;
    o->ring_dec = tl_ring_dec; o->ring_obj = tl_ring_obj; o->ring_obj_lambda = tl_ring_obj_lambda;
    c.f = round_to_dtype(dtype, initial_objective_value);
    o->m = history_length;
    const size_t es = dtype_size(dtype);
    {
        // slot stride: n rounded up to 1 KiB, and an ODD number of KiB, so that the 2(m+1) streams
        // never sit a power-of-two distance apart (same HBM channel / bank for equal offsets)
        size_t sb = ((size_t)n * es + 1023) / 1024 * 1024;
        if ((sb / 1024) % 2 == 0 && tune("DZO_TUNE_STRIDE_SKEW", 1)) sb += 1024;
        o->stride = (int64_t)(sb / es);
    }
    o->last_step_length = round_to_dtype(dtype, initial_step_length);   // as legacy BFGS :779
    int32_t rc = core_alloc(c);
    if (rc != DZO_OK) { delete o; return rc; }
    {
        // blocked (tile-major) ring: when the constructor knows that the single-pass step applies
        const int vecn = 16 / (int)es;
        o->blocked = tl_want_blocked && tune("DZO_TUNE_BLOCKED", 1) != 0 && tune("DZO_TUNE_SINGLE_PASS", 1) != 0 &&
                     history_length <= point_max_k(dtype) && n >= 4 * vecn && (uint64_t)n * es < (1ull << 32) &&
                     (n % vecn == 0 || tune("DZO_TUNE_POINT_RING", 1) != 0);   // (a ragged n: phantom padding, load_vec_tail -- on the POINT ring only)
    }
    o->nslots = o->m + (o->blocked ? 2 : 1);
    const int m1 = o->nslots;
    const size_t slab = (size_t)m1 * (size_t)o->stride * es;
    hipError_t e;
#define ALLOC(ptr, bytes)                                                                          \
    e = hipMalloc((void **)&(ptr), (bytes));                                                       \
    if (e != hipSuccess) { dzo_lbfgs_destroy(o); if (e == hipErrorOutOfMemory) { set_error("out of device memory allocating the L-BFGS state (%zu bytes)", (size_t)(bytes)); (void)hipGetLastError(); return DZO_ERR_NOMEM; } return hip_fail(e, "hipMalloc", __FILE__, __LINE__); }
    o->interleaved = tune("DZO_TUNE_INTERLEAVE", 1) != 0;
    if (o->blocked) {
        const int64_t nvec = (n + 16 / (int64_t)es - 1) / (16 / (int64_t)es);
        o->ring_rows = (nvec + kRowOwn - 1) / kRowOwn;
        {
            const int64_t stream_bytes = ((o->ring_rows * kTileBytes + 1023) / 1024 | 1) * 1024;   // an odd number of KiB (HBM channel skew)
            const int ms = m1 + (o->ring_obj == 2 ? 1 : 0);      // (log-sum-exp: slot m1 holds the tiles of the centre vector c)
            const uint64_t total = (uint64_t)2 * ms * (uint64_t)stream_bytes;
            // (few streams: the whole wave-row of a tile-major ring sits in a handful of DRAM pages and its reads win --
            // n = 1e7: m = 5 pass 234 us tile-major / 253 us stream-major, m = 10 398 / 383, m = 20 684 / 660)
            if (tune("DZO_TUNE_STREAM_MAJOR", o->m >= 9 ? 1 : 0) != 0 && total + (1u << 20) < (1ull << 32)) {      // 32-bit byte offsets in the passes
                o->tile_stride = stream_bytes; o->rowbytes = kTileBytes; o->ring_bytes = (size_t)total;
            } else {
                o->tile_stride = kTileBytes; o->rowbytes = (int64_t)2 * ms * kTileBytes;
                o->ring_bytes = (size_t)o->ring_rows * (size_t)o->rowbytes;
            }
        }
        ALLOC(o->S, o->ring_bytes);
        o->Y = nullptr;
        o->pair_stride = 0;
        ALLOC(o->dx_lin, (size_t)o->stride * es);
        ALLOC(o->dg_lin, (size_t)o->stride * es);
    } else if (o->interleaved) {
        ALLOC(o->S, 2 * slab);
        o->Y = (char *)o->S + (size_t)o->stride * es;
        o->pair_stride = 2 * o->stride;
    } else {
        ALLOC(o->S, slab);
        ALLOC(o->Y, slab);
        o->pair_stride = o->stride;
    }
    ALLOC(o->d_alloc, (size_t)o->stride * es + 4096);
    o->d = (char *)o->d_alloc + 3 * 1024;             // off the allocator's alignment grid (see xbak / gbak)
    o->gram_u = tune("DZO_TUNE_GRAM_U", 4);
    o->fused_finish = tune("DZO_TUNE_FUSED_FINISH", 0) != 0;
    o->fused_finish_max = tune("DZO_TUNE_FUSED_FINISH_MAX", 65536);
    o->speculate = tune("DZO_TUNE_SPECULATE", 1) != 0;
    o->gram_variant = tune("DZO_TUNE_GRAM_VARIANT", 1);
    o->gram_peel = tune("DZO_TUNE_GRAM_PEEL", 1);
    o->gram_fresh_plain = tune("DZO_TUNE_GRAM_FRESH_PLAIN", 1);
    o->gram_skip0 = tune("DZO_TUNE_GRAM_SKIP0", 1);
    o->combine_fresh_plain = tune("DZO_TUNE_COMBINE_FRESH_PLAIN", 1);
    o->single_pass = tune("DZO_TUNE_SINGLE_PASS", 1) != 0;
    o->fused_post = tune("DZO_TUNE_FUSED_POST", 1) != 0;
    o->combine_nts = tune("DZO_TUNE_COMBINE_NTS", 1) != 0;
    o->combine_u = tune("DZO_TUNE_COMBINE_U", 4);
    o->combine_blocks_per_cu = tune("DZO_TUNE_COMBINE_BPC", 0);   // 0: resident blocks
    o->gram_bpc = tune("DZO_TUNE_GRAM_BPC", 0);                 // 0: as many blocks as are resident at once
    o->gram_grid = ctx().cus * (o->gram_bpc > 0 ? o->gram_bpc : 8);   // upper bound (sizes the partials)
    if (o->gram_grid > kMaxPartialBlocks) o->gram_grid = kMaxPartialBlocks;
    {
        const int64_t tile_v = 64 * (int64_t)o->gram_u;
        const int64_t tiles = (n / (16 / (int64_t)es) + tile_v - 1) / tile_v;
        if (tiles < o->gram_grid) o->gram_grid = (int)(tiles > 0 ? tiles : 1);
    }
    const size_t nscal = 2 + 2 + 2 * ((size_t)m1 + 2) + (size_t)m1 + 3 * kMaxHistory + 8 + 2 * (size_t)m1 * m1 + 2 * kMaxHistory + (2 * kMaxHistory + 8) +
                         (size_t)kGramValues * kMaxHistory * (o->gram_grid * kWaves + 1) + 4 * (size_t)kMaxPartialBlocks;
    double *base = nullptr;
    ALLOC(base, nscal * sizeof(double));
#undef ALLOC
    (void)hipMemset(base, 0, nscal * sizeof(double));
    (void)hipDeviceSynchronize();
    o->gram_ticket = reinterpret_cast<unsigned int *>(base); base += 2;   // (zeroed with the rest; re-armed by the kernel)
    o->xg_differs = reinterpret_cast<int32_t *>(base); base += 2;
    o->pscal = base; base += 2 * ((size_t)m1 + 2);
    o->rho = base; base += m1;
    o->alpha = base; base += kMaxHistory;
    o->coef = base; base += kMaxHistory;
    o->scale = base; base += 8;
    o->alpha_sp = base; base += kMaxHistory;
    o->coef_sp = base; base += kMaxHistory;
    o->scale_sp = base; base += 8;
    o->Gyy = base; base += (size_t)m1 * m1;
    o->Gsy = base; base += (size_t)m1 * m1;
    o->sg = base; base += kMaxHistory;
    o->yg = base; base += kMaxHistory;
    o->gram_partials = base; base += (size_t)kGramValues * kMaxHistory * (o->gram_grid * kWaves + 1);
    o->link_partials = base;
    // (from here on a failure must give the state back: the ring alone is gigabytes)
    auto finish = [&]() -> int32_t {
        // :366-374 zero-filled deltas: the whole ring starts zeroed
        if (o->blocked) {
            DZO_HIP(hipMemsetAsync(o->S, 0, o->ring_bytes, c.stream));
            DZO_HIP(hipMemsetAsync(o->dx_lin, 0, (size_t)o->stride * es, c.stream));
            DZO_HIP(hipMemsetAsync(o->dg_lin, 0, (size_t)o->stride * es, c.stream));
        } else if (o->interleaved) {
            DZO_HIP(hipMemsetAsync(o->S, 0, 2 * slab, c.stream));
        } else {
            DZO_HIP(hipMemsetAsync(o->S, 0, slab, c.stream));
            DZO_HIP(hipMemsetAsync(o->Y, 0, slab, c.stream));
        }
        o->k = 0; o->newest = o->nslots - 1;   // spare() == 0
        o->refresh_delta_ptrs();
        DZO_HIP(hipMemsetAsync(o->d_alloc, 0, (size_t)o->stride * es + 4096, c.stream));   // (the padding behind element n: +0, the first pass reads whole vectors)
        if (o->blocked && tune("DZO_TUNE_POINT_RING", 1) != 0) {
            // point ring: the start point and its gradient are point 0
            o->points = true;
            o->lazy_d = tune("DZO_TUNE_LAZY_D", 1) != 0;
            o->point_sets = tune("DZO_TUNE_POINT_SETS", 1) == 1 ? 1 : 2;
            o->xg_host_may_write = true;                      // (the caller owns x0 / g0 and may change them before the first step)
            DZO_DISPATCH(dtype, (ring_scatter<T>(o, x_dev, o->s_slot_v(o->newest)), ring_scatter<T>(o, g_dev, o->y_slot_v(o->newest))));
            o->g_valid = 1u << o->newest;
            DZO_HIP(hipGetLastError());
        }
        // :381-388
        double gnorm = 0;
        DZO_TRY(dot_blocking(c.stream, n, dtype, g_dev, g_dev, c.partials(), c.host, &gnorm));
        gnorm = dtype == DZO_F32 ? (double)sqrtf((float)gnorm) : sqrt(gnorm);
        c.is_stuck = (gnorm == 0.0);                          // :382 iszero
        if (c.is_stuck) {
            DZO_HIP(hipMemsetAsync(o->d, 0, (size_t)o->stride * es, c.stream));   // :384
        } else {
            const double sc = round_to_dtype(dtype, -initial_step_length / gnorm);
            DZO_DISPATCH(dtype, launch_scal_oop<T>(c.stream, n, (T *)o->d, (T)sc, (const T *)g_dev));  // :386-387
        }
        DZO_HIP(hipStreamSynchronize(c.stream));
        return DZO_OK;
    };
    rc = finish();
    if (rc != DZO_OK) { o->x_user = nullptr; dzo_lbfgs_destroy(o); return rc; }   // (x_user cleared: nothing to settle)
    *out = o;
    return DZO_OK;
}

int32_t dzo_lbfgs_destroy(dzo_lbfgs_t o) {
    if (!o) return DZO_OK;
    DeviceScope scope(o->device);
    if (o->core.stream && o->x_user) (void)lbfgs_settle(o);   // the caller's arrays end up holding the final point / gradient
    unsettled_retire(o);                                  // (a dzo_synchronize on another thread may be settling this handle right now)
    if (o->core.stream) (void)hipStreamSynchronize(o->core.stream);
    if (o->S) (void)hipFree(o->S);
    if (o->Y && !o->interleaved && !o->blocked) (void)hipFree(o->Y);
    if (o->d_alloc) (void)hipFree(o->d_alloc);
    if (o->twin_slab) (void)hipFree(o->twin_slab);
    if (o->dx_lin) (void)hipFree(o->dx_lin);
    if (o->dg_lin) (void)hipFree(o->dg_lin);
    if (o->export_slab) (void)hipFree(o->export_slab);
    if (o->xt) (void)hipFree(o->xt);
    if (o->gt) (void)hipFree(o->gt);
    if (o->gram_ticket) (void)hipFree(o->gram_ticket);
    core_free(o->core);
    delete o;
    return DZO_OK;
}

int32_t dzo_lbfgs_create_callbacks(dzo_constraint_fn constraint, dzo_objective_fn objective, dzo_gradient_fn gradient,
                                   void *cb_ctx, int64_t n, int32_t history_length, int32_t dtype, void *x_dev,
                                   double initial_step_length, dzo_lbfgs_t *out) {
    DZO_TRY(require_init());
    DZO_REQUIRE(objective && gradient && x_dev && out, DZO_ERR_INVALID, "null argument");
    DZO_REQUIRE(dtype == DZO_F32 || dtype == DZO_F64, DZO_ERR_INVALID, "bad dtype %d", dtype);
    DZO_TRY(require_same_backend("LBFGSOptimizer", "src/DZOptimization.jl:410,420", x_dev, "initial_point", nullptr, ""));   // before any callback sees it
    if (constraint) DZO_REQUIRE(constraint(cb_ctx, x_dev) != 0, DZO_ERR_ASSERT,
                                "@assert constraint_function!(initial_point) (src/DZOptimization.jl:412-414)");
    const double f0 = objective(cb_ctx, x_dev);           // :416
    void *g = nullptr;
    DZO_HIP(hipMalloc(&g, (size_t)((n + 63) / 64 * 64) * dtype_size(dtype)));   // :418 similar(x0)
    gradient(cb_ctx, g, x_dev);                           // :421
    int32_t rc = dzo_lbfgs_create(n, history_length, dtype, x_dev, g, f0, initial_step_length, out);
    if (rc != DZO_OK) { (void)hipFree(g); return rc; }
    (*out)->core.owns_g = true;
    return dzo_lbfgs_set_callbacks(*out, constraint, objective, gradient, cb_ctx);
}

int32_t dzo_lbfgs_create_problem(dzo_problem_t problem, int32_t history_length, void *x_dev,
                                 double initial_step_length, dzo_lbfgs_t *out) {
    DZO_TRY(require_init());
    DZO_REQUIRE(problem && x_dev && out, DZO_ERR_INVALID, "null argument");
    DZO_TRY(require_same_backend("LBFGSOptimizer", "src/DZOptimization.jl:410,420", x_dev, "initial_point", nullptr, ""));
    if (problem->cons_on)                                 // :412-414 @assert constraint_function!(initial_point)
        DZO_TRY(dzo_box_clamp(problem->n, problem->dtype, x_dev, problem->cons_lo, problem->cons_hi));
    double f0 = 0;
    DZO_TRY(dzo_problem_eval(problem, x_dev, &f0));       // :416
    void *g = nullptr;
    DZO_HIP(hipMalloc(&g, (size_t)((problem->n + 63) / 64 * 64) * dtype_size(problem->dtype)));
    int32_t rc = dzo_problem_grad(problem, g, x_dev);     // :421
    // the single-pass step will apply (built-in chained Rosenbrock, no decorators): tile-major history ring
    const bool lse_points = problem->kind == DZO_PROBLEM_LSE && !ring_decor_of(problem).any() && ((uintptr_t)problem->c & 15u) == 0 &&
                            tune("DZO_TUNE_LSE_POINTS", 1) != 0;
    tl_want_blocked = (problem->kind == DZO_PROBLEM_ROSENBROCK_CHAIN ||
                       (problem->kind == DZO_PROBLEM_QUADRATIC_CHAIN && !ring_decor_of(problem).any()) || lse_points) && (((uintptr_t)x_dev | (uintptr_t)g) & 15u) == 0;
    tl_ring_dec = ring_decor_of(problem);
    tl_ring_obj = ring_obj_of(problem) >= 1 ? ring_obj_of(problem) : 0; tl_ring_obj_lambda = problem->lambda;
    if (rc == DZO_OK) rc = dzo_lbfgs_create(problem->n, history_length, problem->dtype, x_dev, g, f0, initial_step_length, out);
    tl_want_blocked = false;
    tl_ring_dec = RingDecor(); tl_ring_obj = 0; tl_ring_obj_lambda = 0;
    if (rc != DZO_OK) { (void)hipFree(g); return rc; }
    (*out)->core.owns_g = true;
    if ((*out)->ring_obj == 2 && (*out)->points) {
        // log-sum-exp on the point ring: the centre vector's tiles, and the start point's two scalars as dzo_problem_grad
        // just left them in the handle's workspace (lse_finish_max_kernel / lse_finish_kernel: [max], [f, sum exp])
        dzo_lbfgs_s *o = *out;
        o->lse_c = problem->c;
        DZO_DISPATCH(o->core.dtype, ring_scatter<T>(o, problem->c, o->s_slot_v(o->nslots)));
        const double *ws = problem->scratch + 2 * kMaxPartialBlocks;
        DZO_HIP(hipMemcpyAsync(o->pscal + 2 * o->newest, ws, sizeof(double), hipMemcpyDeviceToDevice, o->core.stream));
        DZO_HIP(hipMemcpyAsync(o->pscal + 2 * o->newest + 1, ws + 3, sizeof(double), hipMemcpyDeviceToDevice, o->core.stream));
        DZO_HIP(hipStreamSynchronize(o->core.stream));
    }
    rc = problem_view_create(problem, &(*out)->core.problem);    // private partial-sum workspace per optimizer
    if (rc != DZO_OK) { dzo_lbfgs_destroy(*out); *out = nullptr; return rc; }
    (*out)->core.box_on = problem->cons_on; (*out)->core.box_lo = problem->cons_lo; (*out)->core.box_hi = problem->cons_hi;
    return DZO_OK;
}

int32_t dzo_lbfgs_set_callbacks(dzo_lbfgs_t o, dzo_constraint_fn constraint, dzo_objective_fn objective,
                                dzo_gradient_fn gradient, void *cb_ctx) {
    DZO_REQUIRE(o, DZO_ERR_INVALID, "null optimizer");
    DeviceScope scope(o->device);
    o->core.constraint = constraint; o->core.objective = objective; o->core.gradient = gradient;
    o->core.cb_ctx = cb_ctx;
    return DZO_OK;
}

int32_t dzo_lbfgs_set_problem(dzo_lbfgs_t o, dzo_problem_t problem) {
    DZO_REQUIRE(o, DZO_ERR_INVALID, "null optimizer");
    DZO_REQUIRE(!problem || (problem->n == o->core.n && problem->dtype == o->core.dtype), DZO_ERR_INVALID,
                "problem size/dtype does not match the optimizer");
    DeviceScope scope(o->device);
    (void)hipStreamSynchronize(o->core.stream);
    problem_view_destroy(o->core.problem);
    o->core.problem = nullptr;
    if (problem) DZO_TRY(problem_view_create(problem, &o->core.problem));
    o->core.box_on = problem && problem->cons_on;
    if (problem) { o->core.box_lo = problem->cons_lo; o->core.box_hi = problem->cons_hi; }
    return DZO_OK;
}

int32_t dzo_lbfgs_set_two_loop_mode(dzo_lbfgs_t o, int32_t mode) {
    DZO_REQUIRE(o, DZO_ERR_INVALID, "null optimizer");
    DZO_REQUIRE(mode == DZO_TWOLOOP_CHAIN || mode == DZO_TWOLOOP_GRAM, DZO_ERR_INVALID, "bad two-loop mode %d", mode);
    DeviceScope scope(o->device);
    DZO_TRY(lbfgs_flush_rho(o));
    if (mode == DZO_TWOLOOP_CHAIN) DZO_TRY(lbfgs_unblock(o));   // the chain kernels walk the pairs as plain vectors
    if (mode == DZO_TWOLOOP_GRAM && o->mode != DZO_TWOLOOP_GRAM) o->gram_rebuild = true;
    o->mode = mode;
    return DZO_OK;
}

int32_t dzo_lbfgs_set_max_halvings(dzo_lbfgs_t o, int64_t max_halvings) {
    DZO_REQUIRE(o, DZO_ERR_INVALID, "null optimizer");
    DeviceScope scope(o->device);
    o->core.max_halvings = max_halvings;
    return DZO_OK;
}

int32_t dzo_lbfgs_step(dzo_lbfgs_t o) {
    DZO_REQUIRE(o, DZO_ERR_INVALID, "null optimizer");
    DeviceScope scope(o->device);
    return lbfgs_step(o);
}

int32_t dzo_lbfgs_direction(dzo_lbfgs_t o) {
    DZO_REQUIRE(o, DZO_ERR_INVALID, "null optimizer");
    DeviceScope scope(o->device);
    std::lock_guard<std::recursive_mutex> lk(o->mu);
    DZO_TRY(lbfgs_leave_points(o));                       // the standalone two-loop works on the pair ring
    // returns after the enqueue (include/dzo.h, asynchrony): step_direction is complete once a getter
    // (dzo_lbfgs_get_ptr ...) or dzo_synchronize has returned, or for work enqueued on the handle's stream
    return lbfgs_direction(o);
}

int32_t dzo_lbfgs_begin_search(dzo_lbfgs_t o) {
    DZO_REQUIRE(o, DZO_ERR_INVALID, "null optimizer");
    DeviceScope scope(o->device);
    DZO_TRY(lbfgs_leave_points(o));                       // the host-driven step works on the contiguous arrays and the pair ring
    DZO_TRY(lbfgs_flush_rho(o));                          // a host-driven step follows: settle what the single pass deferred
    o->gram_ready = false; o->scalars_ready = false; o->spec_scalars = false;
    o->refresh_delta_ptrs();
    return core_begin_search(o->core);
}

int32_t dzo_lbfgs_trial(dzo_lbfgs_t o, double step_size, int32_t *changed) {
    DZO_REQUIRE(o && changed, DZO_ERR_INVALID, "null argument");
    DeviceScope scope(o->device);
    return core_trial(o->core, step_size, o->d, false, changed, nullptr, nullptr);
}

int32_t dzo_lbfgs_accept(dzo_lbfgs_t o, double next_objective_value) {
    DZO_REQUIRE(o, DZO_ERR_INVALID, "null optimizer");
    DeviceScope scope(o->device);
    DZO_TRY(core_accept(o->core, round_to_dtype(o->core.dtype, next_objective_value)));
    DZO_HIP(hipStreamSynchronize(o->core.stream));
    return DZO_OK;
}

int32_t dzo_lbfgs_reject(dzo_lbfgs_t o) {
    DZO_REQUIRE(o, DZO_ERR_INVALID, "null optimizer");
    DeviceScope scope(o->device);
    DZO_TRY(core_reject(o->core));
    DZO_HIP(hipStreamSynchronize(o->core.stream));
    return DZO_OK;
}

int32_t dzo_lbfgs_pre_gradient(dzo_lbfgs_t o) {
    DZO_REQUIRE(o, DZO_ERR_INVALID, "null optimizer");
    DeviceScope scope(o->device);
    OptCore &c = o->core;
    DZO_HIP(hipMemcpyAsync(c.dg, c.g, (size_t)c.n * dtype_size(c.dtype), hipMemcpyDeviceToDevice, c.stream));
    DZO_HIP(hipStreamSynchronize(c.stream));
    return DZO_OK;
}

int32_t dzo_lbfgs_post_gradient(dzo_lbfgs_t o) {
    DZO_REQUIRE(o, DZO_ERR_INVALID, "null optimizer");
    DeviceScope scope(o->device);
    DZO_TRY(lbfgs_post_gradient(o));
    DZO_HIP(hipStreamSynchronize(o->core.stream));
    return DZO_OK;
}

int32_t dzo_lbfgs_get_i(dzo_lbfgs_t o, int32_t what, int64_t *value) {
    DZO_REQUIRE(o && value, DZO_ERR_INVALID, "null argument");
    switch (what) {
    case 0: *value = o->core.is_stuck ? 1 : 0; break;
    case 1: *value = o->core.iteration_count; break;
    case 2: *value = o->core.n; break;
    case 3: *value = o->m; break;
    case 4: *value = o->k; break;
    case 5: *value = o->core.last_trials; break;
    case 6: *value = o->mode; break;
    case 7: *value = o->core.dtype; break;
    case 8: *value = o->history_resets; break;
    case 9: *value = o->descent_resets; break;
    case 10: *value = o->last_step_kind; break;
    case 11: *value = o->single_pass_steps; break;
    case 12: *value = o->single_pass_rejections; break;
    case 13: *value = o->single_pass_retries; break;
    case 14: *value = o->points ? 2 : (o->blocked ? 1 : 0); break;   // history layout: 0 slabs, 1 tiles of pairs, 2 tiles of points
    case 15: *value = o->blocked ? (o->tile_stride == kTileBytes ? 1 : 2) : 0; break;   // arrangement of the tiles: 1 tile-major, 2 stream-major
    case 16: *value = (o->points && DZO_PP_REGRAD != 0) ? 1 : 0; break;   // point pass: 1 = the points' gradients are recomputed from the point tiles, not streamed
    case 18: *value = o->host_write_checks; break;        // steps that first compared the aliased arrays with the point ring (after a pointer hand-out)
    case 17: { int sets = 0; if (o->points) { DZO_DISPATCH(o->core.dtype, sets = point_one_set<T>(o) ? 1 : 2); } *value = sets; break; }   // register sets per wave of the point pass
    default: set_error("dzo_lbfgs_get_i: unknown field %d", what); return DZO_ERR_INVALID;
    }
    return DZO_OK;
}

int32_t dzo_lbfgs_get_s(dzo_lbfgs_t o, int32_t what, double *value) {
    DZO_REQUIRE(o && value, DZO_ERR_INVALID, "null argument");
    DZO_REQUIRE(what >= 0 && what <= 2, DZO_ERR_INVALID, "unknown field %d", what);
    *value = what == 0 ? o->core.f : what == 1 ? o->core.df : o->last_step_length;
    return DZO_OK;
}

int32_t dzo_lbfgs_set_safeguards(dzo_lbfgs_t o, int32_t descent_check, int32_t steepest_descent_fallback) {
    DZO_REQUIRE(o, DZO_ERR_INVALID, "null optimizer");
    DeviceScope scope(o->device);
    o->descent_check = descent_check != 0;
    o->sd_fallback = steepest_descent_fallback != 0;
    return DZO_OK;
}

int32_t dzo_lbfgs_set_line_search(dzo_lbfgs_t o, int32_t kind, double c1, double c2, int32_t max_evals) {
    DZO_REQUIRE(o, DZO_ERR_INVALID, "null optimizer");
    DZO_REQUIRE(kind == DZO_LINE_SEARCH_BACKTRACKING || kind == DZO_LINE_SEARCH_WOLFE, DZO_ERR_INVALID,
                "unknown line search %d", kind);
    DZO_REQUIRE(!(c1 > 0 && c2 > 0) || c1 < c2, DZO_ERR_INVALID, "Wolfe constants need 0 < c1 < c2 < 1");
    DZO_REQUIRE(c2 < 1.0, DZO_ERR_INVALID, "Wolfe constants need 0 < c1 < c2 < 1");
    DeviceScope scope(o->device);
    o->line_search = kind;
    if (c1 > 0) o->wolfe_c1 = round_to_dtype(o->core.dtype, c1);
    if (c2 > 0) o->wolfe_c2 = round_to_dtype(o->core.dtype, c2);
    if (max_evals > 0) o->wolfe_max_evals = max_evals;
    return DZO_OK;
}

int32_t dzo_lbfgs_set_s(dzo_lbfgs_t o, int32_t what, double value) {
    DZO_REQUIRE(o, DZO_ERR_INVALID, "null optimizer");
    DZO_REQUIRE(what >= 0 && what <= 2, DZO_ERR_INVALID, "unknown field %d", what);
    (what == 0 ? o->core.f : what == 1 ? o->core.df : o->last_step_length) = round_to_dtype(o->core.dtype, value);
    return DZO_OK;
}

int32_t dzo_lbfgs_set_stuck(dzo_lbfgs_t o, int32_t is_stuck) {
    DZO_REQUIRE(o, DZO_ERR_INVALID, "null optimizer");
    o->core.is_stuck = is_stuck != 0;
    return DZO_OK;
}

// where field `what` can be read (after this returns, with the stream drained).  hand_out: the caller receives the pointer
// and may write through it (dzo_lbfgs_get_ptr); otherwise the library copies from it itself (dzo_lbfgs_read).
static int32_t lbfgs_field_ptr(dzo_lbfgs_t o, int32_t what, int32_t idx, bool hand_out, void **ptr_dev) {
    DZO_TRY(lbfgs_settle(o, hand_out));                   // current_point / current_gradient ARE the caller's arrays again
    if (what == 1 || what == 3) DZO_TRY(lbfgs_refresh_lin(o));   // blocked ring: delta_point / delta_gradient gathered on demand
    if (what == 4) DZO_TRY(lbfgs_materialize_d(o));              // point ring: step_direction formed on demand
    if ((what == 5 || what == 6) && o->blocked) {
        // S[i] / Y[i] of a blocked ring: a contiguous COPY of the pair's stream (read-only snapshot; use
        // dzo_lbfgs_set_history to install pairs)
        DZO_REQUIRE(idx >= 0 && idx < o->k, DZO_ERR_INVALID, "history index %d out of range [0,%d)", idx, o->k);
        const size_t vb = (size_t)o->stride * dtype_size(o->core.dtype);
        if (!o->export_slab) {
            hipError_t e = hipMalloc(&o->export_slab, 2 * (size_t)o->m * vb);
            if (e != hipSuccess) { set_error("out of device memory for the contiguous copies of the history (%zu bytes)", 2 * (size_t)o->m * vb); (void)hipGetLastError(); return DZO_ERR_NOMEM; }
        }
        void *dst = (char *)o->export_slab + ((what == 5 ? 0 : (size_t)o->m) + (size_t)idx) * vb;
        const void *src = what == 5 ? o->s_slot_v(o->slot_of(idx)) : o->y_slot_v(o->slot_of(idx));
        if (o->points) {                                  // pair idx = point idx - point idx+1
            if (what == 6) { DZO_TRY(lbfgs_ensure_g(o, o->slot_of(idx))); DZO_TRY(lbfgs_ensure_g(o, o->slot_of(idx + 1))); }
            const void *older = what == 5 ? o->s_slot_v(o->slot_of(idx + 1)) : o->y_slot_v(o->slot_of(idx + 1));
            DZO_DISPATCH(o->core.dtype, ring_gather_diff<T>(o, src, older, dst));
        } else {
            DZO_DISPATCH(o->core.dtype, ring_gather<T>(o, src, dst));
        }
        DZO_HIP(hipGetLastError());
        DZO_HIP(hipStreamSynchronize(o->core.stream));
        *ptr_dev = dst;
        return DZO_OK;
    }
    DZO_HIP(hipStreamSynchronize(o->core.stream));
    switch (what) {
    case 0: *ptr_dev = o->core.x; break;
    case 1: *ptr_dev = o->core.dx; break;
    case 2: *ptr_dev = o->core.g; break;
    case 3: *ptr_dev = o->core.dg; break;
    case 4: *ptr_dev = o->d; break;
    case 5:
    case 6:
        DZO_REQUIRE(idx >= 0 && idx < o->k, DZO_ERR_INVALID, "history index %d out of range [0,%d)", idx, o->k);
        *ptr_dev = what == 5 ? o->s_slot_v(o->slot_of(idx)) : o->y_slot_v(o->slot_of(idx));
        break;
    default: set_error("dzo_lbfgs_get_ptr: unknown field %d", what); return DZO_ERR_INVALID;
    }
    return DZO_OK;
}

int32_t dzo_lbfgs_get_ptr(dzo_lbfgs_t o, int32_t what, int32_t idx, void **ptr_dev) {
    DZO_REQUIRE(o && ptr_dev, DZO_ERR_INVALID, "null argument");
    DeviceScope scope(o->device);
    return lbfgs_field_ptr(o, what, idx, true, ptr_dev);
}

int32_t dzo_lbfgs_read(dzo_lbfgs_t o, int32_t what, int32_t idx, void *host_dst) {
    DZO_REQUIRE(o && host_dst, DZO_ERR_INVALID, "null argument");
    DeviceScope scope(o->device);
    std::lock_guard<std::recursive_mutex> lk(o->mu);
    void *src = nullptr;
    DZO_TRY(lbfgs_field_ptr(o, what, idx, false, &src));
    DZO_HIP(hipMemcpy(host_dst, src, (size_t)o->core.n * dtype_size(o->core.dtype), hipMemcpyDeviceToHost));
    return DZO_OK;
}

int32_t dzo_lbfgs_get_rho(dzo_lbfgs_t o, double *out, int32_t capacity, int32_t *count) {
    DZO_REQUIRE(o && count, DZO_ERR_INVALID, "null argument");
    DeviceScope scope(o->device);
    *count = o->k;
    if (!out) return DZO_OK;
    DZO_TRY(lbfgs_flush_rho(o));
    DZO_HIP(hipStreamSynchronize(o->core.stream));
    double tmp[kMaxHistory + 2];
    DZO_HIP(hipMemcpy(tmp, o->rho, sizeof(double) * o->nslots, hipMemcpyDeviceToHost));
    for (int i = 0; i < o->k && i < capacity; ++i) out[i] = tmp[o->slot_of(i)];
    return DZO_OK;
}

int32_t dzo_lbfgs_get_alpha(dzo_lbfgs_t o, double *out, int32_t capacity, int32_t *count) {
    DZO_REQUIRE(o && count, DZO_ERR_INVALID, "null argument");
    DeviceScope scope(o->device);
    *count = o->n_alpha;
    if (!out) return DZO_OK;
    DZO_HIP(hipStreamSynchronize(o->core.stream));
    double tmp[kMaxHistory];
    DZO_HIP(hipMemcpy(tmp, o->alpha, sizeof(double) * kMaxHistory, hipMemcpyDeviceToHost));
    for (int i = 0; i < o->n_alpha && i < capacity; ++i) out[i] = tmp[i];
    return DZO_OK;
}

int32_t dzo_lbfgs_set_history(dzo_lbfgs_t o, int32_t k, const void *S_dev, const void *Y_dev, const double *rho_or_null,
                              int64_t iteration_count) {
    DZO_REQUIRE(o, DZO_ERR_INVALID, "null optimizer");
    DZO_REQUIRE(k >= 0 && k <= o->m, DZO_ERR_INVALID, "k = %d exceeds history_length %d", k, o->m);
    DZO_REQUIRE(k == 0 || (S_dev && Y_dev), DZO_ERR_INVALID, "null history");
    DeviceScope scope(o->device);
    DZO_TRY(lbfgs_flush_post(o));                        // (a deferred tail belongs to the history that is being replaced: formed, then overwritten)
    if (o->points) {                                     // installed PAIRS: the ring is a pair ring from here on
        DZO_TRY(lbfgs_materialize_d(o));
        DZO_TRY(lbfgs_points_settle(o));
        o->points = false;
        lbfgs_mark_unsettled(o);
        if (ring_ragged(o)) { o->k = 0; DZO_TRY(lbfgs_unblock(o)); }   // (see lbfgs_leave_points; nothing of the old ring is kept)
    }
    o->rho_pending = false;                              // the whole history (and its rho) is replaced
    o->gram_ready = false; o->scalars_ready = false; o->spec_scalars = false;
    OptCore &c = o->core;
    hipStream_t s = c.stream;
    const size_t es = dtype_size(c.dtype);
    // pair i -> slot (k-1-i): newest = k-1 (or m when empty)
    o->k = k; o->n_alpha = k;
    o->newest = k > 0 ? k - 1 : o->nslots - 1;
    double rho_host[kMaxHistory + 2] = {0};
    for (int i = 0; i < k; ++i) {
        const int slot = o->slot_of(i);
        const void *si = (const char *)S_dev + (size_t)i * c.n * es, *yi = (const char *)Y_dev + (size_t)i * c.n * es;
        if (o->blocked) {
            DZO_REQUIRE((((uintptr_t)si | (uintptr_t)yi) & 15u) == 0, DZO_ERR_INVALID, "history rows must be 16-byte aligned");
            DZO_DISPATCH(c.dtype, (ring_scatter<T>(o, si, o->s_slot_v(slot)), ring_scatter<T>(o, yi, o->y_slot_v(slot))));
            DZO_HIP(hipGetLastError());
        } else {
            DZO_HIP(hipMemcpyAsync(o->s_slot_v(slot), si, (size_t)c.n * es, hipMemcpyDeviceToDevice, s));
            DZO_HIP(hipMemcpyAsync(o->y_slot_v(slot), yi, (size_t)c.n * es, hipMemcpyDeviceToDevice, s));
        }
        if (rho_or_null) {
            rho_host[slot] = rho_or_null[i];
        } else {
            double r = 0;
            DZO_TRY(dot_blocking(s, c.n, c.dtype, si, yi, c.partials(), c.host, &r));
            rho_host[slot] = round_to_dtype(c.dtype, r);
        }
    }
    DZO_HIP(hipStreamSynchronize(s));
    DZO_HIP(hipMemcpy(o->rho, rho_host, sizeof(double) * o->nslots, hipMemcpyHostToDevice));
    // the spare slots hold delta_point / delta_gradient: zero them like a fresh optimizer (:366-374)
    o->refresh_delta_ptrs();
    o->lin_stale = false;
    DZO_HIP(hipMemset(c.dx, 0, (size_t)o->stride * es));
    DZO_HIP(hipMemset(c.dg, 0, (size_t)o->stride * es));
    DZO_HIP(hipDeviceSynchronize());   // null-stream memset/D2D copies are asynchronous to the host and to our non-blocking streams
    c.iteration_count = iteration_count;
    o->gram_rebuild = true;
    o->gram_stale = 0;
    return DZO_OK;
}

int32_t dzo_lbfgs_stream(dzo_lbfgs_t o, void **hip_stream) {
    DZO_REQUIRE(o && hip_stream, DZO_ERR_INVALID, "null argument");
    *hip_stream = (void *)o->core.stream;
    return DZO_OK;
}

}  // extern "C"

// dev instrumentation (tools/wave_times.py; not part of include/dzo.h): the clocks the point pass left behind when run with
// DZO_TUNE_SP_DEBUG=1024 -- 100-MHz ticks, [2 w] = start and [2 w + 1] = end of wave w = 4 block + wave-in-block
extern "C" int32_t dzo_debug_wave_times(unsigned long long *out_host, int32_t count) {
    DZO_HIP(hipDeviceSynchronize());
    DZO_REQUIRE(out_host && count > 0 && count <= 1024 * 4 * 2, DZO_ERR_INVALID, "count out of range");
    DZO_HIP(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(dzo::g_dev_wave_times), sizeof(unsigned long long) * (size_t)count));
    return DZO_OK;
}

// ---------------------------------------------------------------------------- self-test of fd_div (see there)
namespace dzo {
__device__ __forceinline__ uint64_t fd_mix(uint64_t z) {       // splitmix64
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ double fd_make(uint64_t bits, int exp_span) {   // a double with the given significand bits and a mid-range exponent
    const uint64_t mant = bits & 0xfffffffffffffull;
    const int e = 1023 - exp_span + (int)((bits >> 52) % (uint64_t)(2 * exp_span + 1));
    const uint64_t sign = (bits >> 63) << 63;
    return __longlong_as_double((long long)(sign | ((uint64_t)e << 52) | mant));
}
// mode 0: random a, b.  1: b with a special significand (all ones, all zeros, one bit), random a.  2: a = RN(q b) for a
// random q (quotients that are exact or one rounding away from exact).  3: a = RN((q + half an ulp) b): quotients next to
// a rounding boundary.  4: small integers.  Counts the pairs whose fd_div differs from a / b in any bit.
__global__ __launch_bounds__(kBlock) void fd_selftest_kernel(uint64_t seed, int64_t per_thread, int mode, unsigned long long *mismatches,
                                                             unsigned long long *checked, double *first) {
    const uint64_t tid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    unsigned long long bad = 0, seen = 0;
    for (int64_t it = 0; it < per_thread; ++it) {
        const uint64_t r0 = fd_mix(seed + tid * 0x100000001b3ull + (uint64_t)it * 0x9e3779b97f4a7c15ull);
        const uint64_t r1 = fd_mix(r0), r2 = fd_mix(r1);
        double a = fd_make(r0, 400), b = fd_make(r1, 400);
        if (mode == 1) {
            const int pick = (int)(r2 % 6);
            uint64_t mant = pick == 0 ? 0xfffffffffffffull : pick == 1 ? 0ull : pick == 2 ? 1ull : pick == 3 ? 0xffffffffffffeull
                          : pick == 4 ? (1ull << (r2 >> 8) % 52) : 0x8000000000000ull;
            b = fd_make((r1 & ~0xfffffffffffffull) | mant, 400);
        } else if (mode == 2 || mode == 3) {
            const double q = fd_make(r2, 50);
            b = fd_make(r1, 100);
            double qq = q;
            if (mode == 3) qq = __longlong_as_double(__double_as_longlong(q)) ;   // (the half ulp comes from the product's own rounding below)
            a = qq * b;                                                           // RN(q b): a / b is q or next to a boundary around q
            if (mode == 3) a = __builtin_fma(qq, b, 0.5 * (__longlong_as_double(__double_as_longlong(qq) + 1) - qq) * b);
        } else if (mode == 4) {
            a = (double)(int64_t)(r0 % 100000) - 50000.0;
            b = (double)(int64_t)(r1 % 4096 + 1);
        }
        if (b == 0.0) continue;
        const double y = 1.0 / b;
        if (!(fd_mid(a) && fd_mid(b) && fd_mid(y))) continue;
        const double want = a / b, got = fd_div(a, b, y);
        seen += 1;
        if (__double_as_longlong(want) != __double_as_longlong(got)) {
            if (bad == 0 && atomicAdd(mismatches, 0ull) == 0) { first[0] = a; first[1] = b; first[2] = want; first[3] = got; }
            bad += 1;
        }
    }
    if (bad) atomicAdd(mismatches, bad);
    atomicAdd(checked, seen);
}
}  // namespace dzo

extern "C" int32_t dzo_selftest_fast_div(uint64_t seed, int64_t pairs, int32_t mode, int64_t *checked, int64_t *mismatches, double *first4) {
    DZO_TRY(require_init());
    DZO_REQUIRE(checked && mismatches && pairs >= 1 && mode >= 0 && mode <= 4, DZO_ERR_INVALID, "bad argument");
    unsigned long long *dev = nullptr;
    double *first = nullptr;
    DZO_HIP(hipMalloc((void **)&dev, 2 * sizeof(unsigned long long)));
    DZO_HIP(hipMalloc((void **)&first, 4 * sizeof(double)));
    DZO_HIP(hipMemset(dev, 0, 2 * sizeof(unsigned long long)));
    DZO_HIP(hipMemset(first, 0, 4 * sizeof(double)));
    const int grid = ctx().cus * 8;
    const int64_t per_thread = (pairs + (int64_t)grid * kBlock - 1) / ((int64_t)grid * kBlock);
    hipLaunchKernelGGL(fd_selftest_kernel, dim3(grid), dim3(kBlock), 0, ctx().stream, seed, per_thread, mode, dev, dev + 1, first);
    DZO_HIP(hipGetLastError());
    DZO_HIP(hipStreamSynchronize(ctx().stream));
    unsigned long long host[2] = {0, 0};
    DZO_HIP(hipMemcpy(host, dev, sizeof(host), hipMemcpyDeviceToHost));
    if (first4) DZO_HIP(hipMemcpy(first4, first, 4 * sizeof(double), hipMemcpyDeviceToHost));
    (void)hipFree(dev); (void)hipFree(first);
    *mismatches = (int64_t)host[0]; *checked = (int64_t)host[1];
    return DZO_OK;
}
