// dzo_problems.hip -- device-side synthetic objectives (K12).
//
// These are the *user's* callbacks in the reference (objective_function / gradient_function!,
// src/DZOptimization.jl:324-325); the reference ships only the 2-D Rosenbrock pair
// (legacy/ExampleFunctions.jl:10-24, README.md:25-31).  Elementwise expressions are the same
// explicit-fma expressions as oracle/dzo_oracle_impl.h so values agree bit-for-bit per term;
// only the (deterministic, two-stage) reduction order differs.
#include <limits>

#include "dzo_problems.h"
#include "dzo_rosen.h"

namespace dzo {

// ------------------------------------------------------------------ 2-D Rosenbrock (config 1)
template <typename T>
__global__ void rosen2d_eval_kernel(const T *__restrict__ x, double *__restrict__ result) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        T t1 = (T)1 - x[0];
        T t2 = x[1] - x[0] * x[0];
        result[0] = (double)(t1 * t1 + (T)100 * (t2 * t2));
    }
}
template <typename T>
__global__ void rosen2d_grad_kernel(T *__restrict__ g, const T *__restrict__ x) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        T t1 = (T)1 - x[0];
        T t2 = x[1] - x[0] * x[0];
        g[0] = (T)-2 * t1 - (T)400 * x[0] * t2;
        g[1] = (T)200 * t2;
    }
}

// ------------------------------------------------------------------ chained Rosenbrock (config 3)
template <typename T>
__global__ __launch_bounds__(kBlock) void rosen_chain_eval_kernel(int64_t n, const T *__restrict__ x,
                                                                  double *__restrict__ partials) {
    constexpr int N = Vec16<T>::N;
    __shared__ double lds[kWaves];
    double acc = 0;
    const int64_t nvec = n / N;
    const int64_t nthreads = (int64_t)gridDim.x * kBlock;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += nthreads) {
        const int64_t i = v * N;
        T xv[N];
        load16(x + i, xv);
        const T xnext = (i + N < n) ? x[i + N] : (T)0;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const T xn = (j + 1 < N) ? xv[(j + 1) % N] : xnext;
            if (i + j + 1 < n) acc += rosen_term<T>(xv[j], xn);
        }
    }
    const int64_t t = nvec * N + (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t + 1 < n) acc += rosen_term<T>(x[t], x[t + 1]);
    double r = block_sum(acc, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = r;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void rosen_chain_grad_kernel(int64_t n, T *__restrict__ g,
                                                                  const T *__restrict__ x) {
    constexpr int N = Vec16<T>::N;
    const int64_t nvec = n / N;
    const int64_t nthreads = (int64_t)gridDim.x * kBlock;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += nthreads) {
        const int64_t i = v * N;
        T xv[N], gv[N];
        load16(x + i, xv);
        const T xprev = (i > 0) ? x[i - 1] : (T)0;
        const T xnext = (i + N < n) ? x[i + N] : (T)0;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const T xp = (j > 0) ? xv[(j + N - 1) % N] : xprev;
            const T xn = (j + 1 < N) ? xv[(j + 1) % N] : xnext;
            gv[j] = rosen_grad_elem<T>(i + j, n, xp, xv[j], xn);
        }
        store16(g + i, gv);
    }
    const int64_t t = nvec * N + (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t < n) g[t] = rosen_grad_elem<T>(t, n, t > 0 ? x[t - 1] : (T)0, x[t], t + 1 < n ? x[t + 1] : (T)0);
}

// ------------------------------------------------------------------ chained quadratic (dzo_rosen.h)
// (plain elementwise kernels: this objective's fast path is the L-BFGS point pass, which recomputes these in registers)
template <typename T>
__global__ __launch_bounds__(kBlock) void qchain_eval_kernel(int64_t n, const T *__restrict__ x, T lambda, double *__restrict__ partials) {
    __shared__ double lds[kWaves];
    double acc = 0;
    const int64_t nthreads = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += nthreads)
        acc += qchain_term<T>(qchain_coef<T>(i, n, lambda), x[i], i + 1 < n ? x[i + 1] : (T)0);
    const double r = block_sum(acc, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = r;
}
template <typename T>
__global__ __launch_bounds__(kBlock) void qchain_grad_kernel(int64_t n, T *__restrict__ g, const T *__restrict__ x, T lambda) {
    const int64_t nthreads = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += nthreads)
        g[i] = qchain_grad_coef<T>(qchain_coef<T>(i, n, lambda), i > 0 ? x[i - 1] : (T)0, x[i], i + 1 < n ? x[i + 1] : (T)0);
}

// Fused tail of an accepted L-BFGS step for the chained Rosenbrock objective (K3 + K5 + K12):
//   delta_point    = x - x_old                       (src/DZOptimization.jl:145)
//   g_new          = grad f(x)                       (:479)
//   delta_gradient = g_new - g_old                   (:478,:480)
//   partials of rho = delta_point . delta_gradient   (:505)
// One pass: reads x, x_old (in dx), g_old; writes dx, g (in place), dg.  6n elements instead of
// the 11n of the four separate passes (axpby 3n, copy 2n, gradient 2n, delta+rho 4n).  Neighbour
// values for the gradient stencil come from wave shuffles; only lanes 0 / 63 touch memory again.
template <typename T>
__global__ __launch_bounds__(kBlock) void rosen_accept_grad_delta_kernel(int64_t n, const T *__restrict__ x,
                                                                         T *dx, T *g,
                                                                         T *__restrict__ dg,
                                                                         double *__restrict__ partials,
                                                                         const int32_t *__restrict__ gate,
                                                                         const T *xold, const T *gold) {
    // xold / gold: where x_old and g_old are read from -- dx and g themselves on the ordinary path
    // (in place), the single pass's backups after its first trial was rejected (no restore copies)
    constexpr int N = Vec16<T>::N;
    constexpr int U = 2;
    __shared__ double lds[kWaves];
    // speculative launch: the host enqueues this kernel before it knows whether the trial was
    // accepted; the device-side decision (decide_kernel) gates it
    if (gate && *gate != 1) return;
    const int lane = threadIdx.x & 63;
    double acc = 0;
    const int64_t nvec = n / N;
    const int64_t nthreads = (int64_t)gridDim.x * kBlock;
    for (int64_t base = (int64_t)blockIdx.x * kBlock * U; base < nvec; base += nthreads * U) {
        T xv[U][N], xo[U][N], go[U][N];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t v = base + (int64_t)u * kBlock + threadIdx.x;
            ok[u] = v < nvec;
            if (ok[u]) { load16(x + v * N, xv[u]); load16(xold + v * N, xo[u]); load16(gold + v * N, go[u]); }
            else {
#pragma unroll
                for (int j = 0; j < N; ++j) { xv[u][j] = 0; xo[u][j] = 0; go[u][j] = 0; }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t v = base + (int64_t)u * kBlock + threadIdx.x;
            const int64_t i = v * N;
            // neighbours: previous lane's last element, next lane's first element
            T xprev = lane_prev<T>(xv[u][N - 1]);
            T xnext = lane_next<T>(xv[u][0]);
            if (ok[u]) {
                if (lane == 0) xprev = i > 0 ? x[i - 1] : (T)0;
                if (lane == 63 || v + 1 >= nvec) xnext = (i + N < n) ? x[i + N] : (T)0;
                T gn[N], dxn[N], dgn[N];
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const T xp = j > 0 ? xv[u][(j + N - 1) % N] : xprev;
                    const T xn = j + 1 < N ? xv[u][(j + 1) % N] : xnext;
                    gn[j] = rosen_grad_elem<T>(i + j, n, xp, xv[u][j], xn);
                    dxn[j] = xv[u][j] - xo[u][j];
                    dgn[j] = gn[j] - go[u][j];
                    acc = __builtin_fma((double)dxn[j], (double)dgn[j], acc);
                }
                store16(dx + i, dxn);
                store16(g + i, gn);
                store16(dg + i, dgn);
            }
        }
    }
    // scalar tail
    const int64_t t = nvec * N + (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t < n) {
        const T gn = rosen_grad_elem<T>(t, n, t > 0 ? x[t - 1] : (T)0, x[t], t + 1 < n ? x[t + 1] : (T)0);
        const T dxn = x[t] - xold[t];
        const T dgn = gn - gold[t];
        acc = __builtin_fma((double)dxn, (double)dgn, acc);
        dx[t] = dxn; g[t] = gn; dg[t] = dgn;
    }
    const double r = block_sum(acc, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = r;
}

// Trial point + objective in one pass (K2 + K12): x = fma(t, d, x_old) (src/DZOptimization.jl:124,
// bit-identical to trial_kernel), the "any element changed" flag (:128), the backup of x_old on
// the first trial (:118) and the per-block partial sums of every objective term f_i(x_i, x_{i+1})
// whose two elements are held by one wave-row (64 lanes x 16 B): the right-hand neighbour of a
// lane's last element comes from the next lane by shuffle.  The terms that straddle two rows
// (one per 64 vectors) and the scalar tail are summed by rosen_edge_terms_kernel afterwards --
// they cannot be formed here because on the first trial x is overwritten in place by other
// waves.  Saves the objective kernel's own pass over x (n T per trial).
template <typename T, bool FIRST>
__global__ __launch_bounds__(kBlock) void rosen_trial_eval_kernel(int64_t n, T *x, T *__restrict__ backup,
                                                                  const T *__restrict__ d, T t,
                                                                  int32_t *__restrict__ changed,
                                                                  double *__restrict__ partials) {
    constexpr int N = Vec16<T>::N;
    constexpr int U = 4;
    __shared__ double lds[kWaves];
    __shared__ int lds_flag;
    const int lane = threadIdx.x & 63;
    const int64_t nvec = n / N;
    const int64_t nthreads = (int64_t)gridDim.x * kBlock;
    const T *src = FIRST ? x : backup;
    bool diff = false;
    double acc = 0;
    for (int64_t base = (int64_t)blockIdx.x * kBlock * U; base < nvec; base += nthreads * U) {
        T xo[U][N], dv[U][N];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t v = base + (int64_t)u * kBlock + threadIdx.x;
            ok[u] = v < nvec;
            if (ok[u]) {
                load16(src + v * N, xo[u]);
                load16(d + v * N, dv[u]);
            } else {
#pragma unroll
                for (int j = 0; j < N; ++j) { xo[u][j] = (T)0; dv[u][j] = (T)0; }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t v = base + (int64_t)u * kBlock + threadIdx.x;
            T xn[N];
#pragma unroll
            for (int j = 0; j < N; ++j) {
                xn[j] = dfma(t, dv[u][j], xo[u][j]);                 // :124
                diff |= ok[u] && !is_equal(xn[j], xo[u][j]);         // :128
            }
            const T xnext = lane_next<T>(xn[0]);
            // the row's last vector (lane 63, or the last vector of x) leaves its final term to the edge kernel
            const bool has_next = lane != 63 && v + 1 < nvec;
            if (ok[u]) {
#pragma unroll
                for (int j = 0; j + 1 < N; ++j) acc += rosen_term<T>(xn[j], xn[j + 1]);
                if (has_next) acc += rosen_term<T>(xn[N - 1], xnext);
                store16(x + v * N, xn);
                if (FIRST) store16(backup + v * N, xo[u]);           // :118
            }
        }
    }
    // scalar tail (n not a multiple of the vector width): the point only; its terms are edge terms
    const int64_t e = nvec * N + (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e < n) {
        const T xo = src[e];
        const T xn = dfma(t, d[e], xo);
        diff |= !is_equal(xn, xo);
        if (FIRST) backup[e] = xo;
        x[e] = xn;
    }
    block_raise_flag(diff, changed, &lds_flag);
    const double r = block_sum(acc, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = r;
}

// Objective terms that straddle two wave-rows of rosen_trial_eval_kernel, one thread per row, plus
// the terms of the scalar tail.  Reads the finished x (2 elements per term).
template <typename T>
__global__ __launch_bounds__(kBlock) void rosen_edge_terms_kernel(int64_t n, const T *__restrict__ x,
                                                                  double *__restrict__ partials) {
    constexpr int N = Vec16<T>::N;
    __shared__ double lds[kWaves];
    const int64_t nvec = n / N;
    const int64_t rows = (nvec + 63) / 64;
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    double acc = 0;
    if (r < rows) {
        const int64_t vend = (r + 1) * 64 < nvec ? (r + 1) * 64 : nvec;   // one past the row's last vector
        const int64_t i = vend * N - 1;
        if (i + 1 < n) acc += rosen_term<T>(x[i], x[i + 1]);
    }
    if (r == 0) {
        for (int64_t e = nvec * N; e + 1 < n; ++e) acc += rosen_term<T>(x[e], x[e + 1]);
        if (nvec == 0 && n >= 2) { /* all terms are tail terms, handled by the loop above */ }
    }
    const double s = block_sum(acc, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// ------------------------------------------------------------------ dense quadratic (config 2)
// One block per column j of the column-major symmetric A: c_j = A[:,j].x (coalesced along
// the column), g_j = c_j, and the objective partial is x_j*c_j.  No cross-block reduction.
template <typename T, bool WRITE_G>
__global__ __launch_bounds__(kBlock) void quadratic_kernel(int64_t n, const T *__restrict__ A,
                                                           const T *__restrict__ x, T *__restrict__ g,
                                                           double *__restrict__ partials) {
    constexpr int N = Vec16<T>::N;
    __shared__ double lds[kWaves];
    for (int64_t j = blockIdx.x; j < n; j += gridDim.x) {
        const T *col = A + j * n;
        double acc = 0;
        const bool vec = ((n % N) == 0);  // column starts stay 16-B aligned
        if (vec) {
            for (int64_t i = (int64_t)threadIdx.x * N; i < n; i += (int64_t)kBlock * N) {
                T av[N], xv[N];
                load16(col + i, av);
                load16(x + i, xv);
#pragma unroll
                for (int q = 0; q < N; ++q) acc = __builtin_fma((double)av[q], (double)xv[q], acc);
            }
        } else {
            for (int64_t i = threadIdx.x; i < n; i += kBlock) acc = __builtin_fma((double)col[i], (double)x[i], acc);
        }
        double c = block_sum(acc, lds);
        if (threadIdx.x == 0) {
            if (WRITE_G) g[j] = (T)c;
            else partials[j] = c * (double)x[j];
        }
    }
}

__global__ __launch_bounds__(kBlock) void finish_scaled_sum_kernel(const double *__restrict__ partials, int64_t count,
                                                                   double scale, double *__restrict__ result) {
    __shared__ double lds[kWaves];
    double v = 0;
    int64_t i = threadIdx.x;
    for (; i + 7 * kBlock < count; i += 8 * kBlock) {           // (loads first, adds in the plain loop's order)
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = partials[i + (int64_t)u * kBlock];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += t[u];
    }
    for (; i < count; i += kBlock) v += partials[i];
    double r = block_sum(v, lds);
    if (threadIdx.x == 0) result[0] = scale * r;
}

// phi(t) = f(x + ts*dir) for the dense quadratic (legacy/DZOptimization.jl:25-46) without a separate
// trial-point launch: the point is formed on the fly (same fma as phi_point_kernel) and block j
// also stores element j of it and raises the bracket's two flags; finish_scaled_sum_kernel then
// sums the n partials, so the value is bit-identical to the three-launch sequence it replaces.
// (Folding the final sum in as well, "last block done", needs a device-scope fence per block --
// an L2 write-back on this multi-die part: 2.1 ms per step instead of 0.58.)
template <typename T>
__global__ __launch_bounds__(kBlock) void quadratic_phi_kernel(int64_t n, const T *__restrict__ A, const T *__restrict__ x,
                                                               const T *__restrict__ dir, T ts, T *__restrict__ point_out,
                                                               double *__restrict__ partials, int32_t *__restrict__ flags,
                                                               const T *__restrict__ ref) {
    constexpr int N = Vec16<T>::N;
    __shared__ double lds[kWaves];
    for (int64_t j = blockIdx.x; j < n; j += gridDim.x) {
        const T *col = A + j * n;
        double acc = 0;
        const bool vec = ((n % N) == 0);
        if (vec) {
            for (int64_t i = (int64_t)threadIdx.x * N; i < n; i += (int64_t)kBlock * N) {
                T av[N], xv[N], dv[N];
                load16(col + i, av);
                load16(x + i, xv);
                load16(dir + i, dv);
#pragma unroll
                for (int q = 0; q < N; ++q) acc = __builtin_fma((double)av[q], (double)dfma(ts, dv[q], xv[q]), acc);
            }
        } else {
            for (int64_t i = threadIdx.x; i < n; i += kBlock) acc = __builtin_fma((double)col[i], (double)dfma(ts, dir[i], x[i]), acc);
        }
        const double c = block_sum(acc, lds);
        if (threadIdx.x == 0) {
            const T xo = x[j], dj = dir[j];
            const T xt = dfma(ts, dj, xo);
            partials[j] = c * (double)xt;
            point_out[j] = xt;
            if (flags) {                                   // the bracket's flags (:71-80) and its :150 stagnation test
                if (xo != xt) flags[0] = 1;
                if (dj != (T)0) flags[1] = 1;
                if (ref && !is_equal(xt, ref[j])) flags[2] = 1;
            }
        }
    }
}

// Two line-search evaluations in one pass over A (the two searches of the dense BFGS step run
// side by side, legacy/DZOptimization.jl:922-932): request r evaluates f(x + ts[r]*dir[r]); each A
// column is loaded once and used for both.  Per request the arithmetic is that of
// quadratic_phi_kernel, so every value is bit-identical to a separate evaluation.
// One search direction with up to three step sizes evaluated in the same pass (the request the search
// needs now plus the one or two it will most likely need next).
template <typename T> struct PhiDir {
    const T *dir;
    T ts[3];
    T *point_out[3];
    const T *ref[3];       // stagnation test against a stored point (may be null) ...
    int ref_req[3];        // ... or against request ref_req[q] of the same direction in this launch (-1: none)
    T *grad_out[3];        // may be null: A*x_t, the gradient at the trial point, is a by-product (c_j = A[:,j].x_t)
    int active[3];
};

// device-driven searches: the requests as the previous round's finish kernel posted them (see quadratic_phi6_kernel)
template <typename T> __device__ __forceinline__ int phi6_take_requests(PhiDir<T> &a, PhiDir<T> &b, const PhiReqDev *__restrict__ dreq) {
    int any = 0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        a.ts[r] = (T)dreq->ts[0][r]; a.active[r] = dreq->active[0][r]; a.ref_req[r] = dreq->ref_req[0][r];
        b.ts[r] = (T)dreq->ts[1][r]; b.active[r] = dreq->active[1][r]; b.ref_req[r] = dreq->ref_req[1][r];
        any |= a.active[r] | b.active[r];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) { a.ref[r] = nullptr; b.ref[r] = nullptr; }
    const int ra = dreq->use_ref[0][0], rb = dreq->use_ref[1][0];
    if (ra) a.ref[0] = ra == 1 ? a.point_out[0] : (ra == 2 ? a.point_out[1] : a.point_out[2]);
    if (rb) b.ref[0] = rb == 1 ? b.point_out[0] : (rb == 2 ? b.point_out[1] : b.point_out[2]);
    return any;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void quadratic_phi6_kernel(int64_t n, const T *__restrict__ A, const T *__restrict__ x,
                                                                PhiDir<T> a, PhiDir<T> b,
                                                                double *__restrict__ partials, int32_t *__restrict__ flags,
                                                                const PhiReqDev *__restrict__ dreq) {
    constexpr int N = Vec16<T>::N;
    __shared__ double lds6[6 * kWaves];
    // device-driven search (dzo_bfgs.hip): which requests run, their step sizes and their :150 references were posted by the
    // previous round's finish kernel; a round behind a finished search does nothing
    if (dreq) { if (!phi6_take_requests<T>(a, b, dreq)) return; }
    for (int64_t j = blockIdx.x; j < n; j += gridDim.x) {
        const T *col = A + j * n;
        double acc[6] = {0, 0, 0, 0, 0, 0};
        const bool vec = ((n % N) == 0);
        if (vec) {
            for (int64_t i = (int64_t)threadIdx.x * N; i < n; i += (int64_t)kBlock * N) {
                T av[N], xv[N], da[N], db[N];
                load16(col + i, av);
                load16(x + i, xv);
                load16(a.dir + i, da);
                load16(b.dir + i, db);
#pragma unroll
                for (int q = 0; q < N; ++q) {
#pragma unroll
                    for (int r = 0; r < 3; ++r) {      // (wave-uniform skips: later rounds carry one or two requests per direction)
                        if (a.active[r]) acc[r] = __builtin_fma((double)av[q], (double)dfma(a.ts[r], da[q], xv[q]), acc[r]);
                        if (b.active[r]) acc[3 + r] = __builtin_fma((double)av[q], (double)dfma(b.ts[r], db[q], xv[q]), acc[3 + r]);
                    }
                }
            }
        } else {
            for (int64_t i = threadIdx.x; i < n; i += kBlock) {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    acc[r] = __builtin_fma((double)col[i], (double)dfma(a.ts[r], a.dir[i], x[i]), acc[r]);
                    acc[3 + r] = __builtin_fma((double)col[i], (double)dfma(b.ts[r], b.dir[i], x[i]), acc[3 + r]);
                }
            }
        }
        double c[6];
        block_sum_multi<6>(acc, lds6, c);                  // (inactive requests sum zeros; each value = block_sum of it)
        if (threadIdx.x == 0) {
            const T xo = x[j];
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const PhiDir<T> &d = side ? b : a;
                const T dj = d.dir[j];
                T xt[3], refv[3];
#pragma unroll
                for (int r = 0; r < 3; ++r) xt[r] = dfma(d.ts[r], dj, xo);
#pragma unroll
                for (int r = 0; r < 3; ++r) refv[r] = (d.active[r] && d.ref[r]) ? d.ref[r][j] : (T)0;   // first: a reference may be one of this launch's own output buffers
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    if (!d.active[r]) continue;
                    const int slot = side * 3 + r;
                    partials[(int64_t)slot * n + j] = c[slot] * (double)xt[r];
                    d.point_out[r][j] = xt[r];
                    if (d.grad_out[r]) d.grad_out[r][j] = (T)c[slot];      // exactly what quadratic_kernel<WRITE_G> stores
                    if (xo != xt[r]) flags[slot * 3 + 0] = 1;
                    if (dj != (T)0) flags[slot * 3 + 1] = 1;
                    if (d.ref[r]) { if (!is_equal(xt[r], refv[r])) flags[slot * 3 + 2] = 1; }
                    else if (d.ref_req[r] >= 0) { if (!is_equal(xt[r], xt[d.ref_req[r]])) flags[slot * 3 + 2] = 1; }
                }
            }
        }
    }
}

// The same launch with a block walking CG columns at once and three register sets of requests in flight (round 4).  The
// one-column kernel waits for each iteration's four loads before it asks for the next ones (24 KB per CU in flight).  Here
// the loads of iterations it + 1 and it + 2 are on their way while iteration it is computed -- and they are UNCONDITIONAL
// (addresses clamped to the last vector, the sums predicated instead): a prefetch under `if (i < n)` makes the compiler
// wait for vmcnt(0) before the older set may be used, because requests complete in order and a younger request that may
// not have been issued cannot be counted on (that is why the first pipelined forms of this kernel measured nothing,
// profiles/r04_phi6_experiments.md).  x, dir_a, dir_b are loaded and the six trial values formed once per CG columns.
// Element i still belongs to thread (i / N) % 256 in iteration i / (256 N), the per-thread sums run over q, then over the
// iterations, and the block sum is wave_sum + the waves in order: every value is bit for bit what quadratic_phi6_kernel
// (and quadratic_phi_kernel, quadratic_kernel) give.  n % N == 0 only.
template <typename T, int CG, int DEPTH = 2, int BPC = 1>
__global__ __launch_bounds__(kBlock, BPC) void quadratic_phi6_cols_kernel(int64_t n, const T *__restrict__ A, const T *__restrict__ x,
                                                                     PhiDir<T> a, PhiDir<T> b,
                                                                     double *__restrict__ partials, int32_t *__restrict__ flags,
                                                                     const PhiReqDev *__restrict__ dreq) {
    constexpr int N = Vec16<T>::N;
    __shared__ double lds[CG * 6 * kWaves];
    if (dreq) { if (!phi6_take_requests<T>(a, b, dreq)) return; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t groups = (n + CG - 1) / CG;
    const int64_t st = (int64_t)kBlock * N;
    const int64_t i0 = (int64_t)threadIdx.x * N;
    const int iters = (int)((n + st - 1) / st);                             // (uniform; a thread's own count may be one less)
    const int64_t ilast = n - N;
    struct Tile { T av[CG][N], xv[N], da[N], db[N]; };
    for (int64_t grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const int64_t j0 = grp * CG;
        double acc[CG][6];
#pragma unroll
        for (int c = 0; c < CG; ++c)
#pragma unroll
            for (int r = 0; r < 6; ++r) acc[c][r] = 0;
        auto fetch = [&](Tile &t, int it) {                                  // always issued; past the end: the last vector again
            int64_t i = i0 + (int64_t)it * st;
            i = i < n ? i : ilast;
#pragma unroll
            for (int c = 0; c < CG; ++c) {
                const int64_t jc = j0 + c < n ? j0 + c : n - 1;              // (a column past the end: loaded, never used)
                load16(A + jc * n + i, t.av[c]);
            }
            load16(x + i, t.xv);
            load16(a.dir + i, t.da);
            load16(b.dir + i, t.db);
        };
        auto consume = [&](const Tile &t, int it) {
            if (i0 + (int64_t)it * st >= n) return;
            double xt[6][N];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int q = 0; q < N; ++q) {
                    xt[r][q] = a.active[r] ? (double)dfma(a.ts[r], t.da[q], t.xv[q]) : 0.0;
                    xt[3 + r][q] = b.active[r] ? (double)dfma(b.ts[r], t.db[q], t.xv[q]) : 0.0;
                }
            }
#pragma unroll
            for (int c = 0; c < CG; ++c) {
#pragma unroll
                for (int q = 0; q < N; ++q) {
#pragma unroll
                    for (int r = 0; r < 3; ++r) {                              // (wave-uniform skips, as in the one-column kernel)
                        if (a.active[r]) acc[c][r] = __builtin_fma((double)t.av[c][q], xt[r][q], acc[c][r]);
                        if (b.active[r]) acc[c][3 + r] = __builtin_fma((double)t.av[c][q], xt[3 + r][q], acc[c][3 + r]);
                    }
                }
            }
        };
        if constexpr (DEPTH == 2) {
            Tile t0, t1, t2;
            fetch(t0, 0);
            fetch(t1, 1);
            int it = 0;
            for (; it + 3 <= iters; it += 3) {
                fetch(t2, it + 2); consume(t0, it);
                fetch(t0, it + 3); consume(t1, it + 1);
                fetch(t1, it + 4); consume(t2, it + 2);
            }
            if (it < iters) consume(t0, it);
            if (it + 1 < iters) consume(t1, it + 1);
        } else {
            Tile t0, t1;
            fetch(t0, 0);
            int it = 0;
            for (; it + 2 <= iters; it += 2) {
                fetch(t1, it + 1); consume(t0, it);
                fetch(t0, it + 2); consume(t1, it + 1);
            }
            if (it < iters) consume(t0, it);
        }
#pragma unroll
        for (int c = 0; c < CG; ++c) {
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const int on = r < 3 ? a.active[r] : b.active[r - 3];
                if (!on) continue;                                           // (uniform; an inactive request's sum is never read)
                const double w = wave_sum(acc[c][r]);
                if (lane == 0) lds[(c * 6 + r) * kWaves + wave] = w;
            }
        }
        __syncthreads();
        if (threadIdx.x < CG && j0 + threadIdx.x < n) {
            const int c = threadIdx.x;
            const int64_t j = j0 + c;
            const T xo = x[j];
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const PhiDir<T> &d = side ? b : a;
                const T dj = d.dir[j];
                T xtj[3], refv[3];
#pragma unroll
                for (int r = 0; r < 3; ++r) xtj[r] = dfma(d.ts[r], dj, xo);
#pragma unroll
                for (int r = 0; r < 3; ++r) refv[r] = (d.active[r] && d.ref[r]) ? d.ref[r][j] : (T)0;   // first: a reference may be one of this launch's own output buffers
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    if (!d.active[r]) continue;
                    const int slot = side * 3 + r;
                    double cs = 0;
#pragma unroll
                    for (int w = 0; w < kWaves; ++w) cs += lds[(c * 6 + slot) * kWaves + w];
                    partials[(int64_t)slot * n + j] = cs * (double)xtj[r];
                    d.point_out[r][j] = xtj[r];
                    if (d.grad_out[r]) d.grad_out[r][j] = (T)cs;             // exactly what quadratic_kernel<WRITE_G> stores
                    if (xo != xtj[r]) flags[slot * 3 + 0] = 1;
                    if (dj != (T)0) flags[slot * 3 + 1] = 1;
                    if (d.ref[r]) { if (!is_equal(xtj[r], refv[r])) flags[slot * 3 + 2] = 1; }
                    else if (d.ref_req[r] >= 0) { if (!is_equal(xtj[r], xtj[d.ref_req[r]])) flags[slot * 3 + 2] = 1; }
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// The same evaluations on the LOWER TRIANGLE of A (round 4).  A is symmetric by definition (f = 1/2 x'Ax, "dense
// symmetric A", include/dzo.h), so c = A x_t needs every element of the triangle once:
//     row part   c_i += A[i,j] x_t[j]   (j <= i)   thread-local
//     col part   c_j += A[i,j] x_t[i]   (i >  j)   summed across the wave
// -- the split of tri_pass_kernel (dzo_bfgs.hip), here for up to six trial points x_t = fma(ts, dir, x) at once (the two
// line searches' requests of a round, legacy/DZOptimization.jl:922-932) and with a block walking kQtWPB windows, so that
// the row part leaves one value per 128 columns instead of one per 32.  At config 2 a round reads 67 MB of A instead of
// 135 MB (both from the Infinity Cache); a second, small kernel adds the parts in a fixed order and does what thread 0
// of quadratic_phi6_kernel does per element (objective partial, trial point, its gradient, the bracket's flags).
// EVERY evaluation of the dense quadratic goes this way when the triangle form is on (objective, gradient, phi, phi6):
// the stored gradient of a point and a recomputed one are then the same bits (run_and_test!, legacy :1025-1032).
constexpr int kQtPH = 256;           // rows per panel (128 row pairs)
constexpr int kQtCW = 32;            // columns per window (16 per parity)
constexpr int kQtUJ = 8;             // columns of one parity per chunk
#ifndef DZO_QT_WPB
#define DZO_QT_WPB 4
#endif
constexpr int kQtWPB = DZO_QT_WPB;   // windows per block

template <typename T> __device__ __forceinline__ void qt_load2(const T *p, T (&hv)[2]) {
    if constexpr (sizeof(T) == 8) { const double2 q = *reinterpret_cast<const double2 *>(p); hv[0] = q.x; hv[1] = q.y; }
    else { const float2 q = *reinterpret_cast<const float2 *>(p); hv[0] = q.x; hv[1] = q.y; }
}

// rowpart: [6][ranges][n], colpart: [6][panels][n] (ranges = ceil(n / (kQtCW kQtWPB)), panels = ceil(n / kQtPH))
template <typename T>
__global__ __launch_bounds__(kBlock, 2) void quadratic_tri6_kernel(int64_t n, const T *__restrict__ A, const T *__restrict__ x,
                                                                   PhiDir<T> a, PhiDir<T> b, double *__restrict__ rowpart,
                                                                   double *__restrict__ colpart, const PhiReqDev *__restrict__ dreq) {
    const int P = blockIdx.y, R = blockIdx.x;
    const int64_t r_first = (int64_t)R * (kQtCW * kQtWPB);
    if (r_first >= ((int64_t)P + 1) * kQtPH || r_first >= n) return;         // the range lies right of the panel's diagonal
    if (dreq) { if (!phi6_take_requests<T>(a, b, dreq)) return; }
    __shared__ double wp[6][kWaves][kQtCW];
    __shared__ double rp[6][kQtPH];
    // (wave-uniform, and told so: the column index of a load then is a scalar, and x[j] / dir[j] arrive through the scalar cache)
    const int half = __builtin_amdgcn_readfirstlane((int)threadIdx.x / 128);
    const int lane_h = threadIdx.x % 128;
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    const int64_t row = (int64_t)P * kQtPH + 2 * lane_h;
    const int64_t ranges = (n + kQtCW * kQtWPB - 1) / (kQtCW * kQtWPB), panels = (n + kQtPH - 1) / kQtPH;
    int act[6]; T ts[6]; const T *dir[6];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        act[q] = a.active[q]; ts[q] = a.ts[q]; dir[q] = a.dir;
        act[3 + q] = b.active[q]; ts[3 + q] = b.ts[q]; dir[3 + q] = b.dir;
    }
    // the trial points at this thread's two rows
    double v0[6], v1[6];
    {
        const bool in = row < n;
        const T x0 = in ? x[row] : (T)0, x1 = in ? x[row + 1] : (T)0;
        const T a0 = in ? a.dir[row] : (T)0, a1 = in ? a.dir[row + 1] : (T)0;
        const T b0 = in ? b.dir[row] : (T)0, b1 = in ? b.dir[row + 1] : (T)0;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            v0[q] = in ? (double)dfma(ts[q], q < 3 ? a0 : b0, x0) : 0.0;
            v1[q] = in ? (double)dfma(ts[q], q < 3 ? a1 : b1, x1) : 0.0;
        }
    }
    double acc[6][2];
#pragma unroll
    for (int q = 0; q < 6; ++q) { acc[q][0] = 0; acc[q][1] = 0; }
    // loads: unconditional (masked lanes read the first bytes of A), a chunk of eight columns per request, one chunk ahead
    auto issue = [&](int64_t c_first, int c0, T (&hv)[kQtUJ][2]) {
#pragma unroll
        for (int u = 0; u < kQtUJ; ++u) {
            const int64_t j = c_first + half + 2 * (c0 + u);
            const bool on = c0 + u < kQtCW / 2 && j < n && row < n && row + 1 >= j;
            qt_load2<T>(on ? A + j * n + row : A, hv[u]);
        }
    };
    auto chunk = [&](int64_t c_first, int c0, const T (&hv)[kQtUJ][2]) {
        double h0[kQtUJ], h1[kQtUJ], m0[kQtUJ], m1[kQtUJ];
        T xj[kQtUJ], aj[kQtUJ], bj[kQtUJ];
#pragma unroll
        for (int u = 0; u < kQtUJ; ++u) {
            const int64_t j = c_first + half + 2 * (c0 + u);
            const bool inw = c0 + u < kQtCW / 2 && j < n;
            const bool on = inw && row < n && row + 1 >= j;
            const bool lo0 = on && row >= j;                                 // (row == j - 1: an upper element of the pair, left alone)
            const int64_t jc = inw ? j : 0;
            xj[u] = x[jc]; aj[u] = a.dir[jc]; bj[u] = b.dir[jc];
            h0[u] = lo0 ? (double)hv[u][0] : 0.0; h1[u] = on ? (double)hv[u][1] : 0.0;
            m0[u] = row > j ? h0[u] : 0.0; m1[u] = row + 1 > j ? h1[u] : 0.0;   // strictly lower: the mirror part
            if (!inw) { xj[u] = (T)0; aj[u] = (T)0; bj[u] = (T)0; }
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            if (!act[q]) continue;                                          // (wave-uniform: later rounds carry one or two requests per direction)
            double col[8];
#pragma unroll
            for (int u = 0; u < kQtUJ; ++u) {
                const double vj = (double)dfma(ts[q], q < 3 ? aj[u] : bj[u], xj[u]);
                acc[q][0] = __builtin_fma(h0[u], vj, acc[q][0]);
                acc[q][1] = __builtin_fma(h1[u], vj, acc[q][1]);
                col[u] = __builtin_fma(m1[u], v1[q], m0[u] * v0[q]);
            }
            const double tot = wave_sum8(col, ln);
            const int cw = 2 * (c0 + wave_sum8_owner(ln)) + half;           // column within the window
            if (ln < 8 && cw < kQtCW) wp[q][wv][cw] = tot;
        }
    };
    // chunks of a block in order: window w = c / 2, columns (c % 2) kQtUJ ... of each parity; two buffers, one chunk ahead
    T hvA[kQtUJ][2], hvB[kQtUJ][2];
    auto window_on = [&](int w) { const int64_t c = r_first + (int64_t)w * kQtCW; return w < kQtWPB && c < n && c < ((int64_t)P + 1) * kQtPH; };
    auto flush_window = [&](int w) {                                        // the window's column parts: the two waves of a parity
        __syncthreads();
        if (threadIdx.x < kQtCW) {
            const int cw = threadIdx.x, hw = cw & 1;                        // parity hw lives in waves 2 hw, 2 hw + 1
            const int64_t j = r_first + (int64_t)w * kQtCW + cw;
            if (j < n) {
#pragma unroll
                for (int q = 0; q < 6; ++q) if (act[q]) colpart[((int64_t)q * panels + P) * n + j] = wp[q][2 * hw][cw] + wp[q][2 * hw + 1][cw];
            }
        }
        __syncthreads();
    };
    int nwin = 0;
    while (window_on(nwin)) ++nwin;                                         // (uniform)
    const int nchunks = 2 * nwin;
    issue(r_first, 0, hvA);
    for (int c = 0; c < nchunks; c += 2) {
        const int w = c / 2;
        const int64_t c_first = r_first + (int64_t)w * kQtCW;
        issue(c_first, kQtUJ, hvB);                                         // chunk c + 1: the second half of window w
        chunk(c_first, 0, hvA);
        if (c + 2 < nchunks) issue(c_first + kQtCW, 0, hvA);               // chunk c + 2: the first half of window w + 1
        chunk(c_first, kQtUJ, hvB);
        flush_window(w);
    }
    // row part: the two column parities of a row pair
    if (half == 1) {
#pragma unroll
        for (int q = 0; q < 6; ++q) { rp[q][2 * lane_h] = acc[q][0]; rp[q][2 * lane_h + 1] = acc[q][1]; }
    }
    __syncthreads();
    if (half == 0 && row < n) {
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            if (!act[q]) continue;
            double *dst = rowpart + ((int64_t)q * ranges + R) * n + row;
            dst[0] = acc[q][0] + rp[q][2 * lane_h];
            dst[1] = acc[q][1] + rp[q][2 * lane_h + 1];
        }
    }
}

// c_i of every active request = its row parts of the ranges up to the diagonal + its column parts of the panels from
// the diagonal on, fixed order; then, per element, what thread 0 of quadratic_phi6_kernel does for its column.
// partials: [6][n] (the objective's x_t[i] c_i); flags may be null (plain objective / gradient evaluations).
constexpr int kQtRI = 32, kQtRG = kBlock / kQtRI;
template <typename T>
__global__ __launch_bounds__(kBlock) void quadratic_tri6_reduce_kernel(int64_t n, const T *__restrict__ x, PhiDir<T> a, PhiDir<T> b,
                                                                       const double *__restrict__ rowpart, const double *__restrict__ colpart,
                                                                       double *__restrict__ partials, int32_t *__restrict__ flags,
                                                                       const PhiReqDev *__restrict__ dreq) {
    if (dreq) { if (!phi6_take_requests<T>(a, b, dreq)) return; }
    __shared__ double grp[6][kQtRG][kQtRI];
    const int li = threadIdx.x % kQtRI, gq = threadIdx.x / kQtRI;
    const int64_t i = (int64_t)blockIdx.x * kQtRI + li;
    const int64_t ranges = (n + kQtCW * kQtWPB - 1) / (kQtCW * kQtWPB), panels = (n + kQtPH - 1) / kQtPH;
    int act[6];
#pragma unroll
    for (int q = 0; q < 3; ++q) { act[q] = a.active[q]; act[3 + q] = b.active[q]; }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        double s = 0;
        if (act[q] && i < n) {
            for (int64_t r = gq; r <= i / (kQtCW * kQtWPB); r += kQtRG) s += rowpart[((int64_t)q * ranges + r) * n + i];
            for (int64_t pp = i / kQtPH + gq; pp < panels; pp += kQtRG) s += colpart[((int64_t)q * panels + pp) * n + i];
        }
        grp[q][gq][li] = s;
    }
    __syncthreads();
    if (gq != 0 || i >= n) return;
    const T xo = x[i];
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const PhiDir<T> &d = side ? b : a;
        const T dj = d.dir[i];
        T xt[3], refv[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) xt[r] = dfma(d.ts[r], dj, xo);
#pragma unroll
        for (int r = 0; r < 3; ++r) refv[r] = (d.active[r] && d.ref[r]) ? d.ref[r][i] : (T)0;   // first: a reference may be one of this launch's own output buffers
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            if (!d.active[r]) continue;
            const int slot = side * 3 + r;
            double c = 0;
#pragma unroll
            for (int g8 = 0; g8 < kQtRG; ++g8) c += grp[slot][g8][li];
            partials[(int64_t)slot * n + i] = c * (double)xt[r];
            if (d.point_out[r]) d.point_out[r][i] = xt[r];
            if (d.grad_out[r]) d.grad_out[r][i] = (T)c;
            if (flags) {
                if (xo != xt[r]) flags[slot * 3 + 0] = 1;
                if (dj != (T)0) flags[slot * 3 + 1] = 1;
                if (d.ref[r]) { if (!is_equal(xt[r], refv[r])) flags[slot * 3 + 2] = 1; }
                else if (d.ref_req[r] >= 0) { if (!is_equal(xt[r], xt[d.ref_req[r]])) flags[slot * 3 + 2] = 1; }
            }
        }
    }
}

// values to out[0..5], flags {changed, nonzero, differs from ref} of request q to the int32 view of
// out[8..] at 3q; flags re-armed
__global__ __launch_bounds__(kBlock) void finish_phi6_kernel(const double *__restrict__ partials, int64_t count, double scale,
                                                             double *__restrict__ out, int32_t *__restrict__ flags, double ticket) {
    __shared__ double lds6[6 * kWaves];
    double v[6] = {0, 0, 0, 0, 0, 0};
    // (one block pulling 6 x count values through one CU is bound by load latency: eight strided steps of loads are
    // in flight before the first add; the adds keep the order of the plain loop, so every value is what
    // finish_phi_kernel gives for the same partials -- 9.2 -> about 3 us at n = 4096)
    int64_t i = threadIdx.x;
    for (; i + 7 * kBlock < count; i += 8 * kBlock) {
        double t[8][6];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int r = 0; r < 6; ++r) t[u][r] = partials[(int64_t)r * count + i + (int64_t)u * kBlock];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int r = 0; r < 6; ++r) v[r] += t[u][r];
    }
    for (; i < count; i += kBlock) {
#pragma unroll
        for (int r = 0; r < 6; ++r) v[r] += partials[(int64_t)r * count + i];
    }
    double sres[6];
    block_sum_multi<6>(v, lds6, sres);
    if (threadIdx.x == 0) {
        unsigned long long seal = seal_bits(ticket);             // (wait_sealed: out[0..16], seal in out[17])
#pragma unroll
        for (int r = 0; r < 6; ++r) { const double v = scale * sres[r]; out[r] = v; seal ^= seal_bits(v); }
        out[6] = 0; out[7] = 0;
        int32_t *ho = reinterpret_cast<int32_t *>(out + 8);
        for (int q = 0; q < 18; q += 2) {
            const int32_t f0 = flags[q], f1 = flags[q + 1];
            flags[q] = 0; flags[q + 1] = 0;
            ho[q] = f0; ho[q + 1] = f1;
            seal ^= (unsigned long long)(uint32_t)f0 | ((unsigned long long)(uint32_t)f1 << 32);
        }
        store_seal(out + 17, seal);
        __threadfence_system();
        out[20] = ticket;                                        // the host spins on this word (wait_ticket)
        __threadfence_system();
    }
}

// finish of quadratic_phi_kernel: value and the three flags straight into the pinned host buffer
// (out[0] = f, int32 view of out[4..5] = flags), flags re-armed for the next evaluation
__global__ __launch_bounds__(kBlock) void finish_phi_kernel(const double *__restrict__ partials, int64_t count, double scale,
                                                            double *__restrict__ out, int32_t *__restrict__ flags) {
    __shared__ double lds[kWaves];
    double v = 0;
    int64_t i = threadIdx.x;
    for (; i + 7 * kBlock < count; i += 8 * kBlock) {           // (loads first, adds in the plain loop's order: see finish_phi6_kernel)
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = partials[i + (int64_t)u * kBlock];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += t[u];
    }
    for (; i < count; i += kBlock) v += partials[i];
    const double r = block_sum(v, lds);
    if (threadIdx.x == 0) {
        out[0] = scale * r;
        if (flags) {
            int32_t *ho = reinterpret_cast<int32_t *>(out + 4);
            ho[0] = flags[0]; ho[1] = flags[1]; ho[2] = flags[2];
            flags[0] = 0; flags[1] = 0; flags[2] = 0;
        }
        __threadfence_system();
    }
}

// ------------------------------------------------------------------ log-sum-exp (config 4)
template <typename T>
__global__ __launch_bounds__(kBlock) void lse_max_kernel(int64_t n, const T *__restrict__ x,
                                                         double *__restrict__ partials) {
    __shared__ double lds[kWaves];
    double mx = -1.7976931348623157e308;
    const int64_t nthreads = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += nthreads) mx = fmax(mx, (double)x[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = lds[0];
        for (int w = 1; w < kWaves; ++w) r = fmax(r, lds[w]);
        partials[blockIdx.x] = r;
    }
}
__global__ __launch_bounds__(kBlock) void lse_finish_max_kernel(const double *__restrict__ partials, int count,
                                                                double *__restrict__ result) {
    __shared__ double lds[kWaves];
    double mx = -1.7976931348623157e308;
    for (int i = threadIdx.x; i < count; i += kBlock) mx = fmax(mx, partials[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = lds[0];
        for (int w = 1; w < kWaves; ++w) r = fmax(r, lds[w]);
        result[0] = r;
    }
}
// partials[b] = sum exp(x - mx), partials[grid + b] = sum (x - c)^2
template <typename T>
__global__ __launch_bounds__(kBlock) void lse_sums_kernel(int64_t n, const T *__restrict__ x, const T *__restrict__ c,
                                                          const double *__restrict__ mx_dev,
                                                          double *__restrict__ partials) {
    __shared__ double lds[kWaves];
    const T mx = (T)mx_dev[0];
    double se = 0, sq = 0;
    const int64_t nthreads = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += nthreads) {
        se += exp((double)(x[i] - mx));
        const T dlt = x[i] - c[i];
        sq = __builtin_fma((double)dlt, (double)dlt, sq);
    }
    double r1 = block_sum(se, lds);
    double r2 = block_sum(sq, lds);
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = r1;
        partials[gridDim.x + blockIdx.x] = r2;
    }
}
// result[0] = mx + log(se) + lambda/2 * sq ; result[1] = se
__global__ __launch_bounds__(kBlock) void lse_finish_kernel(const double *__restrict__ partials, int count,
                                                            const double *__restrict__ mx_dev, double lambda,
                                                            double *__restrict__ result) {
    __shared__ double lds[kWaves];
    double se = reduce_partials_all(partials, count, lds);
    double sq = reduce_partials_all(partials + count, count, lds);
    if (threadIdx.x == 0) {
        result[1] = se;
        result[0] = mx_dev[0] + log(se) + 0.5 * lambda * sq;
    }
}
template <typename T>
__global__ __launch_bounds__(kBlock) void lse_grad_kernel(int64_t n, T *__restrict__ g, const T *__restrict__ x,
                                                          const T *__restrict__ c, const double *__restrict__ mx_dev,
                                                          const double *__restrict__ se_dev, double lambda) {
    const T mx = (T)mx_dev[0];
    const double se = se_dev[0];
    const int64_t nthreads = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += nthreads) {
        const double sm = exp((double)(x[i] - mx)) / se;
        g[i] = (T)(sm + lambda * (double)(x[i] - c[i]));
    }
}

// ------------------------------------------------------------------ decorators (legacy :219-296)
template <typename T>
__global__ __launch_bounds__(kBlock) void box_clamp_kernel(int64_t n, T *__restrict__ x, T lo, T hi) {
    const int64_t nthreads = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += nthreads) {
        const T v = x[i];
        x[i] = v < lo ? lo : (v > hi ? hi : v);              // clamp(x[i], lo, hi) :269
    }
}
// L2GradientWrapper (:247) then UniformBoxGradientWrapper (:289-294), one pass over g
template <typename T>
__global__ __launch_bounds__(kBlock) void grad_decorate_kernel(int64_t n, T *__restrict__ g, const T *__restrict__ x,
                                                               T two_lambda, int box, T lo, T hi) {
    const int64_t nthreads = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += nthreads) {
        T gi = g[i];
        const T xi = x[i];
        if (two_lambda != (T)0) gi = dfma(two_lambda, xi, gi);
        if (box && ((xi <= lo && gi >= (T)0) || (xi >= hi && gi <= (T)0))) gi = (T)0;
        g[i] = gi;
    }
}
template <typename T>
__global__ __launch_bounds__(kBlock) void sumsq_kernel(int64_t n, const T *__restrict__ x, double *__restrict__ partials) {
    __shared__ double lds[kWaves];
    double acc = 0;
    const int64_t nthreads = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += nthreads) acc = __builtin_fma((double)x[i], (double)x[i], acc);
    const double r = block_sum(acc, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = r;
}
// result[0] = f + lambda * norm2(x)  (:232), in T arithmetic
__global__ __launch_bounds__(kBlock) void l2_finish_kernel(const double *__restrict__ partials, int count, double lambda,
                                                           int to_f32, double *__restrict__ result) {
    __shared__ double lds[kWaves];
    const double ss = reduce_partials_all(partials, count, lds);
    if (threadIdx.x == 0) {
        if (to_f32) result[0] = (double)((float)result[0] + (float)lambda * (float)ss);
        else result[0] = result[0] + lambda * ss;
    }
}

int32_t box_clamp_async(hipStream_t s, int64_t n, int32_t dtype, void *x, double lo, double hi) {
    DZO_TIMED("box_clamp", s);
    const int grid = stream_grid(n, 4);
    DZO_DISPATCH(dtype, hipLaunchKernelGGL(box_clamp_kernel<T>, dim3(grid), dim3(kBlock), 0, s, n, (T *)x, (T)lo, (T)hi));
    DZO_HIP(hipGetLastError());
    return DZO_OK;
}

// ------------------------------------------------------------------ host side
// ---- the triangle form of the dense quadratic's evaluations (quadratic_tri6_kernel)
// MEASURED (round 4, config 2, rocprofv3) and OFF by default (DZO_TUNE_QUAD_TRI=1 selects it, read when a
// problem handle is created): the round's six evaluations take 26.7 us in quadratic_tri6_kernel + 21.1 us in its reduce
// kernel (one window per block: 25 MB of row parts) or 50.6 us in all with four windows per block, against 32.8 us for
// quadratic_phi6_kernel + its share of the finish reading the FULL matrix -- 5250-5330 against 6430-6460 step!()/s.  The
// full matrix streams out of the Infinity Cache at 4.4 TB/s from 4096 independent column blocks; the triangle halves the
// bytes but pays for six transposed wave reductions per chunk and a second kernel, and has 272 (or 1100) tiles to hide
// its load latency with.  Bit-compatible with every search variant (the whole BFGS suite passes with it on).
static bool quad_tri_knob() { return getenv("DZO_TUNE_QUAD_TRI") ? atoi(getenv("DZO_TUNE_QUAD_TRI")) != 0 : false; }
static bool quad_tri_on(const dzo_problem_s *p) {
    static const int min_n = getenv("DZO_TUNE_QUAD_TRI_MIN_N") ? atoi(getenv("DZO_TUNE_QUAD_TRI_MIN_N")) : 1024;
    return p->tri_part != nullptr && p->n % 2 == 0 && p->n >= min_n && (reinterpret_cast<uintptr_t>(p->A) & 15u) == 0;
}
static inline int64_t quad_tri_doubles(int64_t n) {
    const int64_t ranges = (n + kQtCW * kQtWPB - 1) / (kQtCW * kQtWPB), panels = (n + kQtPH - 1) / kQtPH;
    return 6 * (ranges + panels) * n;
}
template <typename T>
static void quad_tri_launch(dzo_problem_s *p, hipStream_t s, const T *x, const PhiDir<T> &a, const PhiDir<T> &b, int32_t *flags, const PhiReqDev *dreq) {
    const int64_t n = p->n;
    const int64_t ranges = (n + kQtCW * kQtWPB - 1) / (kQtCW * kQtWPB), panels = (n + kQtPH - 1) / kQtPH;
    double *rowpart = p->tri_part, *colpart = p->tri_part + 6 * ranges * n;
    hipLaunchKernelGGL(quadratic_tri6_kernel<T>, dim3((unsigned)ranges, (unsigned)panels), dim3(kBlock), 0, s, n, (const T *)p->A, x, a, b, rowpart, colpart, dreq);
    hipLaunchKernelGGL(quadratic_tri6_reduce_kernel<T>, dim3((unsigned)((n + kQtRI - 1) / kQtRI)), dim3(kBlock), 0, s, n, x, a, b,
                       (const double *)rowpart, (const double *)colpart, p->scratch, flags, dreq);
}
// one plain evaluation at x itself (objective partials in scratch[0 .. n), gradient to g when given): request 0 of side a with dir = x, ts = 0
template <typename T> static void quad_tri_eval(dzo_problem_s *p, hipStream_t s, const T *x, T *g) {
    PhiDir<T> a, b;
    memset(&a, 0, sizeof(a)); memset(&b, 0, sizeof(b));
    a.dir = x; b.dir = x;
    for (int r = 0; r < 3; ++r) { a.ref_req[r] = -1; b.ref_req[r] = -1; }
    a.active[0] = 1; a.grad_out[0] = g;                     // x_t = fma(0, x, x) = x
    quad_tri_launch<T>(p, s, x, a, b, nullptr, nullptr);
}

template <typename T> static int32_t eval_async_t(dzo_problem_s *p, hipStream_t s, const T *x, double *result_dev) {
    const int64_t n = p->n;
    switch (p->kind) {
    case DZO_PROBLEM_ROSENBROCK2D: {
        DZO_TIMED("objective_rosenbrock2d", s);
        hipLaunchKernelGGL(rosen2d_eval_kernel<T>, dim3(1), dim3(64), 0, s, x, result_dev);
        break;
    }
    case DZO_PROBLEM_ROSENBROCK_CHAIN: {
        DZO_TIMED("objective_rosenbrock_chain", s);
        const int grid = stream_grid(n, Vec16<T>::N * 2);
        hipLaunchKernelGGL(rosen_chain_eval_kernel<T>, dim3(grid), dim3(kBlock), 0, s, n, x, p->scratch);
        hipLaunchKernelGGL(finish_scaled_sum_kernel, dim3(1), dim3(kBlock), 0, s, p->scratch, (int64_t)grid, 1.0,
                           result_dev);
        break;
    }
    case DZO_PROBLEM_QUADRATIC: {
        DZO_TIMED("objective_quadratic", s);
        const int grid = (int)(n < 65535 ? n : 65535);
        if (quad_tri_on(p)) quad_tri_eval<T>(p, s, x, (T *)nullptr);
        else hipLaunchKernelGGL((quadratic_kernel<T, false>), dim3(grid), dim3(kBlock), 0, s, n, (const T *)p->A, x,
                                (T *)nullptr, p->scratch);
        hipLaunchKernelGGL(finish_scaled_sum_kernel, dim3(1), dim3(kBlock), 0, s, p->scratch, n, 0.5, result_dev);
        break;
    }
    case DZO_PROBLEM_QUADRATIC_CHAIN: {
        DZO_TIMED("objective_quadratic_chain", s);
        const int grid = stream_grid(n, 4);
        hipLaunchKernelGGL(qchain_eval_kernel<T>, dim3(grid), dim3(kBlock), 0, s, n, x, (T)p->lambda, p->scratch);
        hipLaunchKernelGGL(finish_scaled_sum_kernel, dim3(1), dim3(kBlock), 0, s, p->scratch, (int64_t)grid, 1.0, result_dev);
        break;
    }
    case DZO_PROBLEM_LSE: {
        DZO_TIMED("objective_lse", s);
        const int grid = stream_grid(n, 4);
        double *mx = p->scratch + 2 * kMaxPartialBlocks;       // [mx]
        double *res = p->scratch + 2 * kMaxPartialBlocks + 2;   // [f, se]
        hipLaunchKernelGGL(lse_max_kernel<T>, dim3(grid), dim3(kBlock), 0, s, n, x, p->scratch);
        hipLaunchKernelGGL(lse_finish_max_kernel, dim3(1), dim3(kBlock), 0, s, p->scratch, grid, mx);
        hipLaunchKernelGGL(lse_sums_kernel<T>, dim3(grid), dim3(kBlock), 0, s, n, x, (const T *)p->c, mx, p->scratch);
        hipLaunchKernelGGL(lse_finish_kernel, dim3(1), dim3(kBlock), 0, s, p->scratch, grid, mx, p->lambda, res);
        DZO_HIP(hipMemcpyAsync(result_dev, res, sizeof(double), hipMemcpyDeviceToDevice, s));
        break;
    }
    default:
        set_error("unknown problem kind %d", p->kind);
        return DZO_ERR_INVALID;
    }
    DZO_HIP(hipGetLastError());
    return DZO_OK;
}

template <typename T> static int32_t grad_async_t(dzo_problem_s *p, hipStream_t s, T *g, const T *x) {
    const int64_t n = p->n;
    switch (p->kind) {
    case DZO_PROBLEM_ROSENBROCK2D: {
        DZO_TIMED("gradient_rosenbrock2d", s);
        hipLaunchKernelGGL(rosen2d_grad_kernel<T>, dim3(1), dim3(64), 0, s, g, x);
        break;
    }
    case DZO_PROBLEM_ROSENBROCK_CHAIN: {
        DZO_TIMED("gradient_rosenbrock_chain", s);
        hipLaunchKernelGGL(rosen_chain_grad_kernel<T>, dim3(stream_grid(n, Vec16<T>::N * 2)), dim3(kBlock), 0, s, n,
                           g, x);
        break;
    }
    case DZO_PROBLEM_QUADRATIC: {
        DZO_TIMED("gradient_quadratic", s);
        const int grid = (int)(n < 65535 ? n : 65535);
        if (quad_tri_on(p)) quad_tri_eval<T>(p, s, x, g);
        else hipLaunchKernelGGL((quadratic_kernel<T, true>), dim3(grid), dim3(kBlock), 0, s, n, (const T *)p->A, x, g,
                                (double *)nullptr);
        break;
    }
    case DZO_PROBLEM_QUADRATIC_CHAIN: {
        DZO_TIMED("gradient_quadratic_chain", s);
        hipLaunchKernelGGL(qchain_grad_kernel<T>, dim3(stream_grid(n, 4)), dim3(kBlock), 0, s, n, g, x, (T)p->lambda);
        break;
    }
    case DZO_PROBLEM_LSE: {
        DZO_TIMED("gradient_lse", s);
        const int grid = stream_grid(n, 4);
        double *mx = p->scratch + 2 * kMaxPartialBlocks;
        double *res = p->scratch + 2 * kMaxPartialBlocks + 2;
        hipLaunchKernelGGL(lse_max_kernel<T>, dim3(grid), dim3(kBlock), 0, s, n, x, p->scratch);
        hipLaunchKernelGGL(lse_finish_max_kernel, dim3(1), dim3(kBlock), 0, s, p->scratch, grid, mx);
        hipLaunchKernelGGL(lse_sums_kernel<T>, dim3(grid), dim3(kBlock), 0, s, n, x, (const T *)p->c, mx, p->scratch);
        hipLaunchKernelGGL(lse_finish_kernel, dim3(1), dim3(kBlock), 0, s, p->scratch, grid, mx, p->lambda, res);
        hipLaunchKernelGGL(lse_grad_kernel<T>, dim3(grid), dim3(kBlock), 0, s, n, g, x, (const T *)p->c, mx, res + 1,
                           p->lambda);
        break;
    }
    default:
        set_error("unknown problem kind %d", p->kind);
        return DZO_ERR_INVALID;
    }
    DZO_HIP(hipGetLastError());
    return DZO_OK;
}

bool problem_has_fused_post(const dzo_problem_s *p, const void *x, const void *dx, const void *g, const void *dg) {
    auto al = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    return p && p->kind == DZO_PROBLEM_ROSENBROCK_CHAIN && p->l2 == 0.0 && !p->bg_on && al(x) && al(dx) && al(g) && al(dg);
}

int32_t problem_fused_post_async(dzo_problem_s *p, hipStream_t s, const void *x, void *dx, void *g, void *dg,
                                 double *partials, int *grid_out, const int32_t *gate, const void *xold_src, const void *gold_src) {
    const int64_t n = p->n;
    DZO_TIMED("lbfgs_fused_accept_grad_delta", s);
    const int vecn = p->dtype == DZO_F64 ? 2 : 4;
    const int grid = stream_grid(n, vecn * 2);
    const void *xo = xold_src ? xold_src : dx, *go = gold_src ? gold_src : g;
    if (p->dtype == DZO_F64)
        hipLaunchKernelGGL(rosen_accept_grad_delta_kernel<double>, dim3(grid), dim3(kBlock), 0, s, n, (const double *)x,
                           (double *)dx, (double *)g, (double *)dg, partials, gate, (const double *)xo, (const double *)go);
    else
        hipLaunchKernelGGL(rosen_accept_grad_delta_kernel<float>, dim3(grid), dim3(kBlock), 0, s, n, (const float *)x,
                           (float *)dx, (float *)g, (float *)dg, partials, gate, (const float *)xo, (const float *)go);
    *grid_out = grid;
    DZO_HIP(hipGetLastError());
    return DZO_OK;
}

// Fused trial point + objective partials (chained Rosenbrock, 16-B aligned operands, no decorators).
bool problem_trial_eval_async(dzo_problem_s *p, hipStream_t s, void *x, void *backup, const void *d, double t,
                              bool first, int32_t *changed, const double **partials, int64_t *count, double *scale) {
    if (!p || p->kind != DZO_PROBLEM_ROSENBROCK_CHAIN || p->l2 != 0.0 || p->cons_on) return false;
    if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(backup) | reinterpret_cast<uintptr_t>(d)) & 15u) != 0) return false;
    const int64_t n = p->n;
    const int vecn = p->dtype == DZO_F64 ? 2 : 4;
    const int grid = stream_grid(n, vecn * 4);
    const int64_t rows = (n / vecn + 63) / 64;
    int egrid = (int)((rows + kBlock - 1) / kBlock);
    if (egrid < 1) egrid = 1;
    if ((int64_t)grid + egrid > 2 * kMaxPartialBlocks) return false;    // partials live in p->scratch
    {
        DZO_TIMED("lbfgs_trial_objective", s);
#define L(TT, F) hipLaunchKernelGGL((rosen_trial_eval_kernel<TT, F>), dim3(grid), dim3(kBlock), 0, s, n, (TT *)x, (TT *)backup, (const TT *)d, (TT)t, changed, p->scratch)
        if (p->dtype == DZO_F64) { if (first) L(double, true); else L(double, false); }
        else { if (first) L(float, true); else L(float, false); }
#undef L
    }
    {
        DZO_TIMED("objective_edge_terms", s);
        if (p->dtype == DZO_F64) hipLaunchKernelGGL(rosen_edge_terms_kernel<double>, dim3(egrid), dim3(kBlock), 0, s, n, (const double *)x, p->scratch + grid);
        else hipLaunchKernelGGL(rosen_edge_terms_kernel<float>, dim3(egrid), dim3(kBlock), 0, s, n, (const float *)x, p->scratch + grid);
    }
    *partials = p->scratch; *count = grid + egrid; *scale = 1.0;
    return true;
}

// f(x + ts*dir) of the dense quadratic in one launch; also materialises the trial point.  false when
// this objective / these options have no such kernel.
bool problem_phi_async(dzo_problem_s *p, hipStream_t s, const void *x, const void *dir, double ts, void *point_out,
                       int32_t *flags, double *result_dev, const void *ref) {
    if (!p || p->kind != DZO_PROBLEM_QUADRATIC || p->l2 != 0.0 || p->cons_on) return false;
    const int64_t n = p->n;
    DZO_TIMED("objective_quadratic_phi", s);
    const int grid = (int)(n < 65535 ? n : 65535);
    if (quad_tri_on(p)) {
        auto go = [&](auto tag) {
            using T = decltype(tag);
            PhiDir<T> a, b;
            memset(&a, 0, sizeof(a)); memset(&b, 0, sizeof(b));
            a.dir = (const T *)dir; b.dir = (const T *)dir;
            for (int r = 0; r < 3; ++r) { a.ref_req[r] = -1; b.ref_req[r] = -1; }
            a.active[0] = 1; a.ts[0] = (T)ts; a.point_out[0] = (T *)point_out; a.ref[0] = (const T *)ref;
            // (single-request flags live in flags[0..2]: the same words request 0 of side a uses)
            quad_tri_launch<T>(p, s, (const T *)x, a, b, flags, nullptr);
        };
        if (p->dtype == DZO_F64) go(double{}); else go(float{});
    } else if (p->dtype == DZO_F64)
        hipLaunchKernelGGL(quadratic_phi_kernel<double>, dim3(grid), dim3(kBlock), 0, s, n, (const double *)p->A, (const double *)x,
                           (const double *)dir, ts, (double *)point_out, p->scratch, flags, (const double *)ref);
    else
        hipLaunchKernelGGL(quadratic_phi_kernel<float>, dim3(grid), dim3(kBlock), 0, s, n, (const float *)p->A, (const float *)x,
                           (const float *)dir, (float)ts, (float *)point_out, p->scratch, flags, (const float *)ref);
    hipLaunchKernelGGL(finish_phi_kernel, dim3(1), dim3(kBlock), 0, s, (const double *)p->scratch, n, 0.5, result_dev, flags);
    return true;
}

// Up to three step sizes along each of two directions in one pass over A.  req[side] describes the
// requests (inactive ones are skipped); values and flags land in result_dev (>= 17 doubles, may be
// pinned host memory): values [0..5], int32 flags from [8].  `flags`: 18 zeroed int32 on the device.
bool problem_phi6_async(dzo_problem_s *p, hipStream_t s, const void *x, const PhiDirHost req[2], int32_t *flags, double *result_dev, double ticket,
                        const PhiReqDev *dreq) {
    if (!p || p->kind != DZO_PROBLEM_QUADRATIC || p->l2 != 0.0 || p->cons_on) return false;
    const int64_t n = p->n;
    if (p->scratch_doubles < 6 * n) return false;
    DZO_TIMED("objective_quadratic_phi", s);
    const int grid = (int)(n < 65535 ? n : 65535);
    auto launch = [&](auto tag) {
        using T = decltype(tag);
        PhiDir<T> d[2];
        for (int side = 0; side < 2; ++side) {
            d[side].dir = (const T *)req[side].dir;
            for (int r = 0; r < 3; ++r) {
                d[side].ts[r] = (T)req[side].ts[r];
                d[side].point_out[r] = (T *)req[side].point_out[r];
                d[side].ref[r] = (const T *)req[side].ref[r];
                d[side].ref_req[r] = req[side].ref_req[r];
                d[side].grad_out[r] = (T *)req[side].grad_out[r];
                d[side].active[r] = req[side].active[r] ? 1 : 0;
            }
        }
        if (quad_tri_on(p)) quad_tri_launch<T>(p, s, (const T *)x, d[0], d[1], flags, dreq);
        else {
            // DZO_TUNE_PHI6_COLS: columns of A a block walks at once.  1: the one-column kernel (32.4 us per round at config 2);
            // 2, the default: two columns, ONE set of requests ahead, 108 registers = four blocks per CU, so that the 2048 groups
            // are two full rounds of the 1024 resident blocks (27.3 us); 22: two sets ahead, 140 registers, three blocks per CU,
            // 2.67 rounds (28.8 us); 4: four columns, two sets ahead (30.2 us).  Forms pressed into more blocks per CU than their
            // registers allow spill and lose everything (35-55 us).
            const int cols_knob = getenv("DZO_TUNE_PHI6_COLS") ? atoi(getenv("DZO_TUNE_PHI6_COLS")) : 2;
            constexpr int N = Vec16<T>::N;
            const int cg = (n % N != 0 || n < 64) ? 1 : cols_knob;
            const int cgc = cg == 22 ? 2 : (cg > 1 ? cg : 1);
            const int64_t groups = (n + cgc - 1) / cgc;
            const int ggrid = (int)(groups < 65535 ? groups : 65535);
            if (cg == 4) hipLaunchKernelGGL((quadratic_phi6_cols_kernel<T, 4>), dim3(ggrid), dim3(kBlock), 0, s, n, (const T *)p->A, (const T *)x, d[0], d[1], p->scratch, flags, dreq);
            else if (cg == 2) hipLaunchKernelGGL((quadratic_phi6_cols_kernel<T, 2, 1, (sizeof(T) == 8 ? 4 : 2)>), dim3(ggrid), dim3(kBlock), 0, s, n, (const T *)p->A, (const T *)x, d[0], d[1], p->scratch, flags, dreq);
            else if (cg == 22) hipLaunchKernelGGL((quadratic_phi6_cols_kernel<T, 2, 2, 1>), dim3(ggrid), dim3(kBlock), 0, s, n, (const T *)p->A, (const T *)x, d[0], d[1], p->scratch, flags, dreq);
            else hipLaunchKernelGGL(quadratic_phi6_kernel<T>, dim3(grid), dim3(kBlock), 0, s, n, (const T *)p->A, (const T *)x, d[0], d[1],
                                    p->scratch, flags, dreq);
        }
    };
    if (p->dtype == DZO_F64) launch(double{}); else launch(float{});
    if (dreq) return true;                                       // (the caller's finish kernel sums, advances the searches and posts the next requests)
    hipLaunchKernelGGL(finish_phi6_kernel, dim3(1), dim3(kBlock), 0, s, (const double *)p->scratch, n, 0.5, result_dev, flags, ticket);
    return true;
}

// Objective WITHOUT its final one-block sum: leaves per-block partials so that the caller's
// decision kernel can do the sum itself (one launch fewer per trial).  Returns false for the
// objectives whose finish is not a plain scaled sum.
bool problem_eval_partials_async(dzo_problem_s *p, hipStream_t s, const void *x, const double **partials,
                                 int64_t *count, double *scale) {
    if (p->l2 != 0.0) return false;
    const int64_t n = p->n;
    if (p->kind == DZO_PROBLEM_ROSENBROCK_CHAIN) {
        DZO_TIMED("objective_rosenbrock_chain", s);
        const int vecn = p->dtype == DZO_F64 ? 2 : 4;
        const int grid = stream_grid(n, vecn * 2);
        if (p->dtype == DZO_F64) hipLaunchKernelGGL(rosen_chain_eval_kernel<double>, dim3(grid), dim3(kBlock), 0, s, n, (const double *)x, p->scratch);
        else hipLaunchKernelGGL(rosen_chain_eval_kernel<float>, dim3(grid), dim3(kBlock), 0, s, n, (const float *)x, p->scratch);
        *partials = p->scratch; *count = grid; *scale = 1.0;
        return true;
    }
    if (p->kind == DZO_PROBLEM_QUADRATIC_CHAIN) {
        DZO_TIMED("objective_quadratic_chain", s);
        const int grid = stream_grid(n, 4);
        if (p->dtype == DZO_F64) hipLaunchKernelGGL(qchain_eval_kernel<double>, dim3(grid), dim3(kBlock), 0, s, n, (const double *)x, (double)p->lambda, p->scratch);
        else hipLaunchKernelGGL(qchain_eval_kernel<float>, dim3(grid), dim3(kBlock), 0, s, n, (const float *)x, (float)p->lambda, p->scratch);
        *partials = p->scratch; *count = grid; *scale = 1.0;
        return true;
    }
    if (p->kind == DZO_PROBLEM_QUADRATIC) {
        DZO_TIMED("objective_quadratic", s);
        const int grid = (int)(n < 65535 ? n : 65535);
        if (quad_tri_on(p)) { if (p->dtype == DZO_F64) quad_tri_eval<double>(p, s, (const double *)x, (double *)nullptr); else quad_tri_eval<float>(p, s, (const float *)x, (float *)nullptr); }
        else if (p->dtype == DZO_F64) hipLaunchKernelGGL((quadratic_kernel<double, false>), dim3(grid), dim3(kBlock), 0, s, n, (const double *)p->A, (const double *)x, (double *)nullptr, p->scratch);
        else hipLaunchKernelGGL((quadratic_kernel<float, false>), dim3(grid), dim3(kBlock), 0, s, n, (const float *)p->A, (const float *)x, (float *)nullptr, p->scratch);
        *partials = p->scratch; *count = n; *scale = 0.5;
        return true;
    }
    return false;
}

int32_t problem_eval_async(dzo_problem_s *p, hipStream_t s, const void *x, double *result_dev) {
    DZO_DISPATCH(p->dtype, DZO_TRY(eval_async_t<T>(p, s, (const T *)x, result_dev)));
    if (p->l2 != 0.0) {                                      // L2RegularizationWrapper (:231-232)
        DZO_TIMED("objective_l2_term", s);
        const int grid = stream_grid(p->n, 4);
        double *part = p->scratch;                           // the base evaluation's partials are consumed
        DZO_DISPATCH(p->dtype, hipLaunchKernelGGL(sumsq_kernel<T>, dim3(grid), dim3(kBlock), 0, s, p->n, (const T *)x, part));
        hipLaunchKernelGGL(l2_finish_kernel, dim3(1), dim3(kBlock), 0, s, (const double *)part, grid, p->l2,
                           p->dtype == DZO_F32 ? 1 : 0, result_dev);
        DZO_HIP(hipGetLastError());
    }
    return DZO_OK;
}
int32_t problem_grad_async(dzo_problem_s *p, hipStream_t s, void *g, const void *x) {
    DZO_DISPATCH(p->dtype, DZO_TRY(grad_async_t<T>(p, s, (T *)g, (const T *)x)));
    if (p->l2 != 0.0 || p->bg_on) {                          // L2GradientWrapper, UniformBoxGradientWrapper
        DZO_TIMED("gradient_decorators", s);
        const int grid = stream_grid(p->n, 4);
        const double l2 = p->dtype == DZO_F32 ? (double)((float)p->l2 + (float)p->l2) : p->l2 + p->l2;   // :247 lambda + lambda
        DZO_DISPATCH(p->dtype, hipLaunchKernelGGL(grad_decorate_kernel<T>, dim3(grid), dim3(kBlock), 0, s, p->n, (T *)g,
                                                  (const T *)x, (T)l2, p->bg_on ? 1 : 0, (T)p->bg_lo, (T)p->bg_hi));
        DZO_HIP(hipGetLastError());
    }
    return DZO_OK;
}

static int32_t problem_alloc_workspace(dzo_problem_s *p) {
    const int64_t n = p->n;
    const int64_t scratch = (p->kind == DZO_PROBLEM_QUADRATIC ? (6 * n > 2 * kMaxPartialBlocks ? 6 * n : 2 * kMaxPartialBlocks)
                                                              : 2 * kMaxPartialBlocks) + 16;
    p->scratch_doubles = scratch - 16;
    hipError_t e = hipMalloc((void **)&p->scratch, sizeof(double) * (size_t)scratch);
    if (e != hipSuccess) { p->scratch = nullptr; return hip_fail(e, "hipMalloc(problem scratch)", __FILE__, __LINE__); }
    e = hipHostMalloc((void **)&p->host, sizeof(double) * 4, hipHostMallocDefault);
    if (e != hipSuccess) { (void)hipFree(p->scratch); p->scratch = nullptr; p->host = nullptr; return hip_fail(e, "hipHostMalloc", __FILE__, __LINE__); }
    p->result = p->scratch + scratch - 8;
    p->tri_part = nullptr;
    if (p->kind == DZO_PROBLEM_QUADRATIC && n % 2 == 0 && n >= 512 && quad_tri_knob()) {
        // row / column parts of the triangle form (2.3 % of A's own size); without them the full-matrix kernels serve
        if (hipMalloc((void **)&p->tri_part, sizeof(double) * (size_t)quad_tri_doubles(n)) != hipSuccess) { (void)hipGetLastError(); p->tri_part = nullptr; }
    }
    return DZO_OK;
}

int32_t problem_view_create(dzo_problem_s *parent, dzo_problem_s **out) {
    dzo_problem_s *root = parent->parent ? parent->parent : parent;
    dzo_problem_s *v = new dzo_problem_s(*root);
    v->parent = root; v->scratch = nullptr; v->result = nullptr; v->host = nullptr; v->tri_part = nullptr;
    const int32_t rc = problem_alloc_workspace(v);
    if (rc != DZO_OK) { delete v; return rc; }
    *out = v;
    return DZO_OK;
}

void problem_view_sync(dzo_problem_s *v) {
    if (!v || !v->parent) return;
    const dzo_problem_s *r = v->parent;
    v->kind = r->kind; v->n = r->n; v->dtype = r->dtype; v->A = r->A; v->c = r->c; v->lambda = r->lambda;
    v->l2 = r->l2; v->bg_on = r->bg_on; v->bg_lo = r->bg_lo; v->bg_hi = r->bg_hi;
    v->cons_on = r->cons_on; v->cons_lo = r->cons_lo; v->cons_hi = r->cons_hi;
}

void problem_view_destroy(dzo_problem_s *v) {
    if (!v || !v->parent) return;                            // only views are owned by optimizers
    if (v->scratch) (void)hipFree(v->scratch);
    if (v->tri_part) (void)hipFree(v->tri_part);
    if (v->host) (void)hipHostFree(v->host);
    delete v;
}

}  // namespace dzo

using namespace dzo;

extern "C" {

int32_t dzo_problem_create(int32_t kind, int64_t n, int32_t dtype, const void *A_dev, const void *c_dev,
                           double lambda, dzo_problem_t *out) {
    DZO_TRY(require_init());
    DZO_REQUIRE(out, DZO_ERR_INVALID, "null out");
    DZO_REQUIRE(dtype == DZO_F32 || dtype == DZO_F64, DZO_ERR_INVALID, "bad dtype %d", dtype);
    DZO_REQUIRE(kind >= 0 && kind <= DZO_PROBLEM_QUADRATIC_CHAIN, DZO_ERR_INVALID, "unknown problem kind %d", kind);
    DZO_REQUIRE(kind != DZO_PROBLEM_QUADRATIC_CHAIN || lambda > 0, DZO_ERR_INVALID, "the chained quadratic needs lambda > 0 (it is what makes it strictly convex)");
    DZO_REQUIRE(n >= 1, DZO_ERR_INVALID, "n must be >= 1");
    DZO_REQUIRE(kind != DZO_PROBLEM_ROSENBROCK2D || n == 2, DZO_ERR_INVALID, "2-D Rosenbrock needs n == 2");
    DZO_REQUIRE(kind != DZO_PROBLEM_QUADRATIC || A_dev, DZO_ERR_INVALID, "quadratic problem needs A");
    DZO_REQUIRE(kind != DZO_PROBLEM_LSE || c_dev, DZO_ERR_INVALID, "LSE problem needs c");
    dzo_problem_s *p = new dzo_problem_s();
    p->kind = kind; p->n = n; p->dtype = dtype; p->A = A_dev; p->c = c_dev; p->lambda = lambda;
    const int32_t rc = problem_alloc_workspace(p);
    if (rc != DZO_OK) { delete p; return rc; }
    *out = p;
    return DZO_OK;
}

int32_t dzo_problem_set_l2(dzo_problem_t p, double lambda) {
    DZO_REQUIRE(p, DZO_ERR_INVALID, "null problem");
    p->l2 = lambda;
    return DZO_OK;
}

int32_t dzo_problem_set_box_gradient(dzo_problem_t p, int32_t enable, double lower_bound, double upper_bound) {
    DZO_REQUIRE(p, DZO_ERR_INVALID, "null problem");
    DZO_REQUIRE(!enable || lower_bound <= upper_bound, DZO_ERR_INVALID, "lower_bound > upper_bound");
    p->bg_on = enable != 0; p->bg_lo = lower_bound; p->bg_hi = upper_bound;
    return DZO_OK;
}

int32_t dzo_problem_set_box_constraint(dzo_problem_t p, int32_t enable, double lower_bound, double upper_bound) {
    DZO_REQUIRE(p, DZO_ERR_INVALID, "null problem");
    DZO_REQUIRE(!enable || lower_bound <= upper_bound, DZO_ERR_INVALID, "lower_bound > upper_bound");
    p->cons_on = enable != 0; p->cons_lo = lower_bound; p->cons_hi = upper_bound;
    return DZO_OK;
}

int32_t dzo_box_clamp(int64_t n, int32_t dtype, void *x_dev, double lower_bound, double upper_bound) {
    DZO_TRY(require_init());
    DZO_REQUIRE(n >= 0 && (x_dev || n == 0), DZO_ERR_INVALID, "bad argument");
    DZO_REQUIRE(dtype == DZO_F32 || dtype == DZO_F64, DZO_ERR_INVALID, "bad dtype %d", dtype);
    if (n > 0) DZO_TRY(box_clamp_async(ctx().stream, n, dtype, x_dev, lower_bound, upper_bound));
    DZO_HIP(hipStreamSynchronize(ctx().stream));
    return DZO_OK;
}

int32_t dzo_problem_destroy(dzo_problem_t p) {
    if (!p) return DZO_OK;
    (void)hipFree(p->scratch);
    if (p->tri_part) (void)hipFree(p->tri_part);
    (void)hipHostFree(p->host);
    delete p;
    return DZO_OK;
}

int32_t dzo_problem_eval(dzo_problem_t p, const void *x_dev, double *f) {
    DZO_TRY(require_init());
    DZO_REQUIRE(p && x_dev && f, DZO_ERR_INVALID, "null argument");
    hipStream_t s = ctx().stream;
    DZO_TRY(problem_eval_async(p, s, x_dev, p->result));
    DZO_HIP(hipMemcpyAsync(p->host, p->result, sizeof(double), hipMemcpyDeviceToHost, s));
    DZO_HIP(hipStreamSynchronize(s));
    *f = p->dtype == DZO_F32 ? (double)(float)p->host[0] : p->host[0];
    return DZO_OK;
}

int32_t dzo_problem_grad(dzo_problem_t p, void *g_dev, const void *x_dev) {
    DZO_TRY(require_init());
    DZO_REQUIRE(p && x_dev && g_dev, DZO_ERR_INVALID, "null argument");
    DZO_TRY(problem_grad_async(p, ctx().stream, g_dev, x_dev));
    DZO_HIP(hipStreamSynchronize(ctx().stream));
    return DZO_OK;
}

// the built-in objectives in the shape of the reference's callbacks (src/DZOptimization.jl:323-325); ctx = dzo_problem_t
double dzo_problem_objective_cb(void *problem, const void *x_dev) {
    double f = 0;
    if (dzo_problem_eval(static_cast<dzo_problem_t>(problem), x_dev, &f) != DZO_OK) return std::numeric_limits<double>::infinity();
    return f;
}
void dzo_problem_gradient_cb(void *problem, void *g_dev, const void *x_dev) {
    dzo_problem_t p = static_cast<dzo_problem_t>(problem);
    if (dzo_problem_grad(p, g_dev, x_dev) != DZO_OK && p && g_dev)
        (void)hipMemset(g_dev, 0xff, (size_t)p->n * dtype_size(p->dtype));    // NaN in both element types
}
int32_t dzo_problem_constraint_cb(void *problem, void *x_dev) {
    dzo_problem_t p = static_cast<dzo_problem_t>(problem);
    if (!p || !x_dev) return 0;
    if (!p->cons_on) return 1;
    return dzo_box_clamp(p->n, p->dtype, x_dev, p->cons_lo, p->cons_hi) == DZO_OK ? 1 : 0;
}

}  // extern "C"
