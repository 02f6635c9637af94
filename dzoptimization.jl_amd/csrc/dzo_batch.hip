// dzo_batch.hip -- batched dense BFGS (config 5, K11): B independent BFGSOptimizer instances,
// one 256-thread workgroup per instance, the WHOLE step! (legacy/DZOptimization.jl:891-994) on
// the device: both quadratic line searches, accept / reset / terminate, the inverse-Hessian
// update and the next direction.  No host round trip per trial ("run multiple optimizers in
// parallel", README.md:12).
//
// Per instance: vectors (x, g, d, ...) live in LDS for the duration of the launch; H (n x n,
// column-major, 512 KiB at n = 256 fp64) streams from HBM.  H is symmetric bit for bit (the update
// expression :882-884 commutes in i and j), so the step kernel reads and writes its LOWER triangle only:
// a thread owns a PAIR of rows (16-B accesses, coalesced along each column), the two 128-thread halves of
// the workgroup take even / odd columns, and of column j a thread touches the rows >= j.  A symmetric
// product u = H v is then a row part (thread-local) plus a column part (summed across the wave by a
// transposed DPP butterfly, eight or four columns at a time):
//     t        = H * dg                                      (:875)
//     H[i,j]  += delta*(d'_i d'_j) - (t_i d'_j + d'_i t_j)   (:882-884, reference order; i >= j)
//     dnext    = Hnew * g                                    (fused into the update pass, :958-960)
// Traffic per BFGS instance-step: 1.5 n^2 elements (the reference's pass structure moves 4 n^2, SURVEY.md
// 8(d) counts 3 n^2 for a full-storage implementation).  The upper triangle in memory is stale between
// host accesses; dzo_bfgs_batch_get_ptr mirrors it before handing out H.  The two line searches of a step
// run on one wave (see Inst below).
//
// The objective is the chained Rosenbrock function (n = 2 is exactly the README's 2-D
// Rosenbrock); the search logic mirrors oracle/dzo_oracle_impl.h line for line.
#include <cmath>
#include <cstdlib>

#include "dzo_common.h"
#include "dzo_problems.h"

namespace dzo {

constexpr int kHalf = kBlock / 2;

// the objective of every instance: chained Rosenbrock (n = 2: the README's 2-D Rosenbrock) or the dense
// quadratic 1/2 x'Ax with one symmetric A shared by all instances (A_stride = 0) or one A per instance
// (A_stride = elements between consecutive instances' matrices), plus the decorators of
// legacy/DZOptimization.jl:219-296 and QuadraticLineSearch.max_increases (:181-188)
struct BatchObjective {
    int kind = DZO_PROBLEM_ROSENBROCK_CHAIN;
    const void *A = nullptr;            // QUADRATIC: n x n column-major, device
    int64_t A_stride = 0;               // 0: shared by all instances; else instance b's matrix starts at A + b * A_stride
    double l2 = 0;                      // L2RegularizationWrapper / L2GradientWrapper lambda (:225-249); 0 = off
    int bg_on = 0; double bg_lo = 0, bg_hi = 0;       // UniformBoxGradientWrapper (:275-296)
    int cons_on = 0; double cons_lo = 0, cons_hi = 0; // UniformBoxConstraint (:258-272) as constraint_function!
    int max_increases = 0;              // 0 = no cap (:138-151)
};

struct BatchState {
    BatchObjective ob;
    int64_t batch, n;
    void *x, *g, *dx, *dg, *d, *H;
    double *f, *last_step_length;
    int32_t *last_step_type, *has_terminated;
    int64_t *iteration_count;
    unsigned long long *stats;       // [0..7] dev: cycle sums per phase (DZO_TUNE_BATCH_DEBUG bit 2); [8..15] target of masked stores
};

__device__ __forceinline__ double bsum(double v, double *red) {
    v = wave_sum_all(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const double r = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return r;
}

__device__ __forceinline__ bool bany(bool p, int *flag) {
    if (threadIdx.x == 0) *flag = 0;
    __syncthreads();
    if (__any(p) && (threadIdx.x & 63) == 0) *flag = 1;
    __syncthreads();
    const bool r = *flag != 0;
    __syncthreads();
    return r;
}

template <typename T> __device__ __forceinline__ T t_sqrt(T v);
template <> __device__ __forceinline__ double t_sqrt<double>(double v) { return sqrt(v); }
template <> __device__ __forceinline__ float t_sqrt<float>(float v) { return sqrtf(v); }
template <typename T> __device__ __forceinline__ bool t_finite(T v) { return isfinite(v); }
template <typename T> __device__ __forceinline__ T t_max();
template <> __device__ __forceinline__ double t_max<double>() { return 1.7976931348623157e308; }
template <> __device__ __forceinline__ float t_max<float>() { return 3.4028234663852886e38f; }

// same elementwise expressions as dzo_problems.hip / the oracle
template <typename T> __device__ __forceinline__ double b_rosen_term(T xi, T xn) {
    T t1 = (T)1 - xi;
    T t2 = dfma(-xi, xi, xn);
    return (double)dfma((T)100 * t2, t2, t1 * t1);
}
template <typename T> __device__ __forceinline__ T b_rosen_grad(int i, int n, T xp, T xi, T xn) {
    T gi = (T)0;
    if (i + 1 < n) {
        T t2 = dfma(-xi, xi, xn);
        T t1 = (T)1 - xi;
        gi = dfma((T)-400 * xi, t2, (T)-2 * t1);
    }
    if (i > 0) {
        T t2p = dfma(-xp, xp, xi);
        gi = dfma((T)200, t2p, gi);
    }
    return gi;
}

// The line searches of an instance run on ONE wave (lane-strided loops over the LDS vectors, sums and
// votes across the wave by DPP / ballot): at n = 256 a thread of the workgroup would own a single
// element, and every objective evaluation would be a handful of workgroup barriers and LDS round trips --
// measured, that was 99 of the 222 us of a step.  The other three waves wait at one barrier per step.
__device__ __forceinline__ void wave_lds_fence() {
    // LDS operations of one wave execute in order; only the compiler must not move them across
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <typename T> __device__ __forceinline__ T b_clamp(T v, T lo, T hi) { return v < lo ? lo : (v > hi ? hi : v); }

// decorators applied to one gradient component (legacy :241-249 then :289-294), same expressions as
// dzo_problems.hip / the oracle
template <typename T> __device__ __forceinline__ T b_decorate_grad(const BatchObjective &ob, T gi, T xi) {
    if (ob.l2 != 0.0) gi = dfma((T)ob.l2 + (T)ob.l2, xi, gi);                      // :247 g += (lambda + lambda) x
    if (ob.bg_on && ((xi <= (T)ob.bg_lo && gi >= (T)0) || (xi >= (T)ob.bg_hi && gi <= (T)0))) gi = (T)0;
    return gi;
}

// row i of A*p for the shared dense symmetric A (column-major: consecutive i are contiguous)
template <typename T> __device__ __forceinline__ double b_quad_row(const T *A, int n, int i, const T *p) {
    double col = 0;
    for (int j = 0; j < n; ++j) col = __builtin_fma((double)A[(int64_t)j * n + i], (double)p[j], col);
    return col;
}

template <typename T> struct Inst {
    BatchObjective ob;
    int n, lane;
    T *x, *g, *d, *dg, *dx, *y, *yref, *tv;     // LDS vectors
    double *red;                                // LDS [4]
    int *flag;                                  // LDS
    int evals;

    __device__ T objective(const T *p) {
        double acc = 0;
        T f;
        if (ob.kind == DZO_PROBLEM_QUADRATIC) {
            for (int i = lane; i < n; i += 64) acc = __builtin_fma(b_quad_row<T>((const T *)ob.A, n, i, p), (double)p[i], acc);
            f = (T)(0.5 * wave_sum_all_dpp(acc));
        } else {
            for (int i = lane; i + 1 < n; i += 64) acc += b_rosen_term<T>(p[i], p[i + 1]);
            f = (T)wave_sum_all_dpp(acc);
        }
        if (ob.l2 != 0.0) {                                      // L2RegularizationWrapper (:231-232)
            double ss = 0;
            for (int i = lane; i < n; i += 64) ss = __builtin_fma((double)p[i], (double)p[i], ss);
            f = f + (T)ob.l2 * (T)wave_sum_all_dpp(ss);
        }
        evals += 1;
        return f;
    }
    __device__ T norm(const T *p) {
        double acc = 0;
        for (int i = lane; i < n; i += 64) acc = __builtin_fma((double)p[i], (double)p[i], acc);
        return t_sqrt<T>((T)wave_sum_all_dpp(acc));
    }
    // y = x - t*dir with the bracket flags (legacy :71-80)
    __device__ void point(const T *dir, T t, bool *changed, bool *nonzero) {
        bool ch = false, nz = false;
        wave_lds_fence();
        for (int i = lane; i < n; i += 64) {
            const T nw = dfma(-t, dir[i], x[i]);
            ch |= (x[i] != nw);                                  // the flags look at the raw step (:71-80) ...
            nz |= (dir[i] != (T)0);
            y[i] = ob.cons_on ? b_clamp<T>(nw, (T)ob.cons_lo, (T)ob.cons_hi) : nw;   // ... the objective at P(x - t dir) (:36)
        }
        wave_lds_fence();
        if (changed) *changed = __any(ch) != 0;
        if (nonzero) *nonzero = __any(nz) != 0;
    }
    __device__ T phi(const T *dir, T t) {
        point(dir, t, nullptr, nullptr);
        return objective(y);
    }
    __device__ bool same(const T *a, const T *b) {
        bool df = false;
        for (int i = lane; i < n; i += 64) df |= !is_equal(a[i], b[i]);
        return __any(df) == 0;
    }
    __device__ void copy(T *dst, const T *src) {
        wave_lds_fence();
        for (int i = lane; i < n; i += 64) dst[i] = src[i];
        wave_lds_fence();
    }

    // find_three_point_bracket (legacy :49-172) started at t0; see oracle bfgs_bracket
    __device__ void bracket(const T *dir, T f0, T t0, T &x1, T &f1, T &x2, T &f2) {
        x1 = 0; f1 = f0; x2 = 0; f2 = f0;
        if (!t_finite(f0)) return;                               // :64-66
        if (!(t0 > (T)0) || !t_finite(t0)) return;
        bool changed, nonzero;
        point(dir, t0, &changed, &nonzero);                      // :71-80
        if (!nonzero) return;                                    // :83-85
        T step = t0;
        bool small = false;
        while (!changed) {                                       // :91-101
            step += step;
            small = true;
            if (!t_finite(step)) return;
            point(dir, step, &changed, nullptr);
        }
        T fa = objective(y);                                     // :126
        if (small && same(x, y)) return;                         // :119-121
        if (fa <= f0) {                                          // :130
            int increases = 0;
            copy(yref, y);                                       // :136
            for (;;) {                                           // :143-156
                const T dbl = step + step;
                const T fb = phi(dir, dbl);
                increases += 1;
                if ((ob.max_increases > 0 && increases >= ob.max_increases) || !t_finite(fb) || fb > fa || same(y, yref)) {   // :147-150
                    x1 = step; f1 = fa; x2 = dbl; f2 = fb;
                    return;
                }
                step = dbl;
                fa = fb;
                copy(yref, y);                                   // :155
            }
        } else {                                                 // :157-171
            for (;;) {
                const T hs = (T)0.5 * step;
                const T fb = phi(dir, hs);
                if (fb <= f0) { x1 = hs; f1 = fb; x2 = step; f2 = fa; return; }
                if (hs == (T)0) return;
                step = hs;
                fa = fb;
            }
        }
    }

    // QuadraticLineSearch (legacy :191-216)
    __device__ void search(const T *dir, T f0, T t0, T &tb, T &fbest) {
        T x1, f1, x2, f2;
        bracket(dir, f0, t0, x1, f1, x2, f2);                    // :195
        T xb = 0, fb = f0;                                       // :196
        if (f1 < fb) { xb = x1; fb = f1; }
        if (f2 < fb) { xb = x2; fb = f2; }
        const T d1 = f0 - f1, d2 = f2 - f1, sum = d1 + d2;       // :203-205
        if (d1 >= (T)0 && d2 >= (T)0 && sum > (T)0) {            // :206
            const T ratio = ((d1 + d1) + sum) / (sum + sum);     // :207-208
            const T xq = ratio * x1;                             // :209
            const T fq = phi(dir, xq);                           // :210
            if (fq < fb) { xb = xq; fb = fq; }
        }
        tb = xb; fbest = fb;
    }
};

// block-wide gradient of the move phase (all threads; one element each at n = 256), decorators included
template <typename T> __device__ __forceinline__ void block_gradient(const BatchObjective &ob, int n, T *out, const T *p) {
    for (int i = threadIdx.x; i < n; i += kBlock) {
        T gi;
        if (ob.kind == DZO_PROBLEM_QUADRATIC) gi = (T)b_quad_row<T>((const T *)ob.A, n, i, p);
        else gi = b_rosen_grad<T>(i, n, i > 0 ? p[i - 1] : (T)0, p[i], i + 1 < n ? p[i + 1] : (T)0);
        out[i] = b_decorate_grad<T>(ob, gi, p[i]);
    }
    __syncthreads();
}

// RP = row pairs per thread: n <= 256*RP, n even.
// RP = 2 (256 < n <= 512) comes in two forms.  Four columns of H in flight at two blocks per CU, as RP = 1 has them, do not fit
// 256 registers (round 3 shipped that: 48 spilled values, 196 bytes of scratch).  Measured on the 1024-instance shard (round 4,
// tools/run_batch_rp2_ab.sh, instance-steps/s):            n = 384      n = 512
//     4 columns, 2 blocks per CU, spilling                  2.89 M       1.57 M
//     2 columns, 2 blocks per CU (207 registers)            2.82 M       1.58 M      <- WIDE = 0, n <= 448
//     4 columns, 1 block per CU (308 registers)             2.55 M       1.67 M      <- WIDE = 1, beyond
// Neither spills; the narrow form wins while two instances' H still share a CU's bandwidth well, the wide one at the top.
template <typename T, int RP, int WIDE = 0>
__global__ __launch_bounds__(kBlock, RP == 1 ? 2 : (RP == 2 && WIDE == 0 ? 2 : 1)) void batch_step_kernel(BatchState st, int steps, int debug) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n = (int)st.n;
    const int64_t b = blockIdx.x;
    if (st.has_terminated[b]) return;

    T *lds = reinterpret_cast<T *>(smem);
    const int np = (n + 1) & ~1;
    Inst<T> in;
    in.ob = st.ob;
    if (in.ob.A) in.ob.A = (const T *)in.ob.A + b * in.ob.A_stride;     // this instance's matrix
    in.n = n; in.lane = threadIdx.x & 63;
    in.x = lds; in.g = lds + np; in.d = lds + 2 * np; in.dg = lds + 3 * np; in.dx = lds + 4 * np;
    in.y = lds + 5 * np; in.yref = lds + 6 * np; in.tv = lds + 7 * np;
    T *gold = lds + 8 * np;                       // previous gradient (also the GD direction)
    double *part = reinterpret_cast<double *>(lds + 9 * np);   // [n] half-combine scratch
    double *wpart = part + np;                                  // [4 waves][n] column parts of the symmetric products
    in.red = wpart + 4 * np;                                    // [16]: block sums use [0..3], the searches' results [8..13]
    in.flag = reinterpret_cast<int *>(in.red + 16);
    in.evals = 0;

    T *gx = (T *)st.x + b * n, *gg = (T *)st.g + b * n, *gd = (T *)st.d + b * n;
    T *gdx = (T *)st.dx + b * n, *gdg = (T *)st.dg + b * n;
    T *H = (T *)st.H + b * (int64_t)n * n;
    for (int i = threadIdx.x; i < n; i += kBlock) { in.x[i] = gx[i]; in.g[i] = gg[i]; in.d[i] = gd[i]; in.dx[i] = gdx[i]; in.dg[i] = gdg[i]; }
    __syncthreads();
    T f = (T)st.f[b];
    T last_len = (T)st.last_step_length[b];
    int last_type = st.last_step_type[b];
    int64_t iters = st.iteration_count[b];
    bool terminated = false;

    const int half = threadIdx.x / kHalf, lane_h = threadIdx.x % kHalf;

    double *res = in.red + 8;                                     // [6] results of wave 0's searches
    for (int s = 0; s < steps && !terminated; ++s) {
        __syncthreads();                                          // the LDS vectors of the previous step are complete
        long long tk0 = (debug & 2) ? clock64() : 0;
        if (threadIdx.x < 64) {
            const T gn = in.norm(in.g);                           // :921
            T tg, fg, tb, fb;
            in.search(in.g, f, last_len / gn, tg, fg);            // :922-925
            const T bn = in.norm(in.d);                           // :928
            in.search(in.d, f, last_len / bn, tb, fb);            // :929-932
            if (threadIdx.x == 0) { res[0] = (double)gn; res[1] = (double)tg; res[2] = (double)fg; res[3] = (double)bn; res[4] = (double)tb; res[5] = (double)fb; }
        }
        __syncthreads();
        const T grad_norm = (T)res[0], t_g = (T)res[1], f_g = (T)res[2], bfgs_norm = (T)res[3], t_b = (T)res[4], f_b = (T)res[5];
        if ((debug & 2) && threadIdx.x == 0) { const long long tk = clock64(); atomicAdd(st.stats + 0, (unsigned long long)(tk - tk0)); tk0 = tk; }
        const bool take_bfgs = f_b < f && !(f_b > f_g);           // :934
        const bool take_grad = !take_bfgs && f_g < f;             // :962
        if (!take_bfgs && !take_grad) { terminated = true; break; }   // :989
        const T tt = take_bfgs ? t_b : t_g;
        const T *dir = take_bfgs ? in.d : in.g;
        f = take_bfgs ? f_b : f_g;                                // :937 / :965
        last_len = take_bfgs ? t_b * bfgs_norm : t_g * grad_norm; // :938 / :966
        last_type = take_bfgs ? DZO_STEP_BFGS : DZO_STEP_GRADIENT_DESCENT;
        iters += 1;
        // move (:943-950 / :971-978): y = x_new, dx = x_new - x_old, then the new gradient
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += kBlock) {
            const T xo = in.x[i];
            T xn = dfma(-tt, dir[i], xo);
            if (st.ob.cons_on) xn = b_clamp<T>(xn, (T)st.ob.cons_lo, (T)st.ob.cons_hi);   // :946 constraint_function!(point)
            in.y[i] = xn;
            in.dx[i] = xn - xo;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += kBlock) { in.x[i] = in.y[i]; gold[i] = in.g[i]; }
        __syncthreads();
        block_gradient<T>(in.ob, n, in.g, in.x);
        for (int i = threadIdx.x; i < n; i += kBlock) in.dg[i] = in.g[i] - gold[i];
        __syncthreads();

        if ((debug & 2) && threadIdx.x == 0) { const long long tk = clock64(); atomicAdd(st.stats + 1, (unsigned long long)(tk - tk0)); tk0 = tk; }
        if (take_bfgs && !(debug & 1)) {
            // H is symmetric bit for bit (the update expression :882-884 commutes in i, j), so only its LOWER
            // triangle (row >= column) is read and written here: 1.5 n^2 T per step instead of 3 n^2 T.  A
            // thread owns row pairs and walks the columns of its half's parity; of column j it touches the
            // rows >= j only.  A symmetric product u = H v then has two parts:
            //   row part   u_i += H[i,j] v_j   (j <= i)   thread-local, as before
            //   col part   u_j += H[i,j] v_i   (i >  j)   a sum over the threads of the wave -> wpart[wave][j]
            // The upper triangle of the stored matrix is stale until somebody asks for H (dzo_bfgs_batch_get_ptr
            // mirrors it first).
            const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
            auto load2 = [&](const T *p, T (&hv)[2]) {
                if constexpr (sizeof(T) == 8) { const double2 q = *reinterpret_cast<const double2 *>(p); hv[0] = q.x; hv[1] = q.y; }
                else { const float2 q = *reinterpret_cast<const float2 *>(p); hv[0] = q.x; hv[1] = q.y; }
            };
            auto store2 = [&](T *p, const T (&hv)[2]) {
                if constexpr (sizeof(T) == 8) { double2 q; q.x = hv[0]; q.y = hv[1]; *reinterpret_cast<double2 *>(p) = q; }
                else { float2 q; q.x = hv[0]; q.y = hv[1]; *reinterpret_cast<float2 *>(p) = q; }
            };
            // combine the two halves' row parts and the waves' column parts: u_i, fixed order
            auto finish_symmetric = [&](double (&acc)[RP][2], T *out) {
                if (half == 1) {
#pragma unroll
                    for (int r = 0; r < RP; ++r) { const int row = 2 * (lane_h + kHalf * r); if (row < n) { part[row] = acc[r][0]; part[row + 1] = acc[r][1]; } }
                }
                __syncthreads();
                if (half == 0) {
#pragma unroll
                    for (int r = 0; r < RP; ++r) {
                        const int row = 2 * (lane_h + kHalf * r);
                        if (row < n) {
                            // column `row` is even -> half 0 = waves 0, 1; column row + 1 is odd -> waves 2, 3
                            const double c0 = wpart[0 * np + row] + wpart[1 * np + row];
                            const double c1 = wpart[2 * np + row + 1] + wpart[3 * np + row + 1];
                            out[row] = (T)((acc[r][0] + part[row]) + c0);
                            out[row + 1] = (T)((acc[r][1] + part[row + 1]) + c1);
                        }
                    }
                }
                __syncthreads();
            };
            // Columns are taken UJ at a time and software-pipelined two chunks deep: the loads of chunk c + 1
            // are in flight while chunk c is computed, and the column parts of a chunk share one transposed
            // butterfly (wave_sum8).  Every load is UNCONDITIONAL -- lanes outside the lower triangle read
            // the first 16 bytes of H instead (a broadcast) and their values are zeroed -- because a load
            // behind a branch makes the compiler wait for it at the join (vmcnt(0) per load: measured 6.6 us
            // per chunk of eight columns, i.e. eight serial round trips).
            constexpr int UJ = RP == 1 ? 4 : (RP == 2 ? (WIDE ? 4 : 2) : 2);
            const int my_cols = (n - half + 1) / 2;                // columns half, half + 2, ... < n
            auto issue = [&](int c0, T (&hv)[UJ][RP][2]) {
#pragma unroll
                for (int u = 0; u < UJ; ++u) {
                    const int j = half + 2 * (c0 + u);
#pragma unroll
                    for (int r = 0; r < RP; ++r) {
                        const int row = 2 * (lane_h + kHalf * r);
                        const bool act = j < n && row < n && row + 1 >= j;
                        load2(act ? H + (int64_t)j * n + row : H, hv[u][r]);
                    }
                }
            };
            // ---- t = H*dg (:875)
            double acc[RP][2];
#pragma unroll
            for (int r = 0; r < RP; ++r) { acc[r][0] = 0; acc[r][1] = 0; }
            auto symv_chunk = [&](int c0, const T (&hv)[UJ][RP][2]) {
                double col[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int u = 0; u < UJ; ++u) {
                    const int j = half + 2 * (c0 + u);
                    const double vj = j < n ? (double)in.dg[j] : 0.0;
#pragma unroll
                    for (int r = 0; r < RP; ++r) {
                        const int row = 2 * (lane_h + kHalf * r);
                        const bool act = j < n && row < n && row + 1 >= j;   // the pair reaches the lower triangle of column j
                        const double h0 = (act && row >= j) ? (double)hv[u][r][0] : 0.0;   // (row == j - 1: an upper element)
                        const double h1 = act ? (double)hv[u][r][1] : 0.0;
                        const int rc = row < n ? row : 0;
                        acc[r][0] = __builtin_fma(h0, vj, acc[r][0]);
                        acc[r][1] = __builtin_fma(h1, vj, acc[r][1]);
                        const double m0 = row > j ? h0 : 0.0, m1 = row + 1 > j ? h1 : 0.0;   // strictly lower: the mirror part
                        col[u] = __builtin_fma(m0, (double)in.dg[rc], col[u]);
                        col[u] = __builtin_fma(m1, (double)in.dg[rc + 1], col[u]);
                    }
                }
                const double tot = wave_sum8(col, ln);
                const int own = wave_sum8_owner(ln);
                const int ju = half + 2 * (c0 + own);
                if (ln < 8 && own < UJ && ju < n) wpart[wv * np + ju] = tot;
            };
            {
                T hvA[UJ][RP][2], hvB[UJ][RP][2];
                issue(0, hvA);
                for (int c0 = 0; c0 < my_cols; c0 += 2 * UJ) {
                    issue(c0 + UJ, hvB);
                    symv_chunk(c0, hvA);
                    issue(c0 + 2 * UJ, hvA);
                    symv_chunk(c0 + UJ, hvB);
                }
            }
            finish_symmetric(acc, in.tv);
            if ((debug & 2) && threadIdx.x == 0) { const long long tk = clock64(); atomicAdd(st.stats + 2, (unsigned long long)(tk - tk0)); tk0 = tk; }
            // ---- scalars (:873-876) with lambda = -t_b (:954)
            double a = 0, c = 0;
            for (int i = threadIdx.x; i < n; i += kBlock) {
                a = __builtin_fma((double)in.d[i], (double)in.dg[i], a);
                c = __builtin_fma((double)in.dg[i], (double)in.tv[i], c);
            }
            const T overlap = (T)bsum(a, in.red);                // :873
            const T dgt = (T)bsum(c, in.red);
            const T inv = (T)1 / overlap;
            const T delta = (-t_b) * overlap + dgt;              // :876
            for (int i = threadIdx.x; i < n; i += kBlock) in.d[i] = in.d[i] * inv;   // :874
            __syncthreads();
            // ---- rank-2 update (:878-886) of the lower triangle fused with dnext = Hnew*g (:958-960)
            T di[RP][2], ti[RP][2];
#pragma unroll
            for (int r = 0; r < RP; ++r) {
                const int row = 2 * (lane_h + kHalf * r);
                const bool ok = row < n;
                di[r][0] = ok ? in.d[row] : (T)0; di[r][1] = ok ? in.d[row + 1] : (T)0;
                ti[r][0] = ok ? in.tv[row] : (T)0; ti[r][1] = ok ? in.tv[row + 1] : (T)0;
                acc[r][0] = 0; acc[r][1] = 0;
            }
            T *dummy = reinterpret_cast<T *>(st.stats + 8) + 2 * (threadIdx.x & 1);   // 32 bytes nobody reads: target of the masked lanes' stores
            auto update_chunk = [&](int c0, const T (&hv)[UJ][RP][2]) {
                double col[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int u = 0; u < UJ; ++u) {
                    const int j = half + 2 * (c0 + u);
                    const int jc = j < n ? j : 0;
                    const T sj = in.d[jc], tj = in.tv[jc];         // :879-880
                    const double gj = j < n ? (double)in.g[jc] : 0.0;
#pragma unroll
                    for (int r = 0; r < RP; ++r) {
                        const int row = 2 * (lane_h + kHalf * r);
                        const bool act = j < n && row < n && row + 1 >= j;
                        const bool lo0 = act && row >= j;          // (row == j - 1: an upper element, written back as it was)
                        T nv[2];
                        nv[0] = hv[u][r][0] + (delta * (di[r][0] * sj) - (ti[r][0] * sj + di[r][0] * tj));   // :882-884
                        nv[1] = hv[u][r][1] + (delta * (di[r][1] * sj) - (ti[r][1] * sj + di[r][1] * tj));
                        if (!lo0) nv[0] = hv[u][r][0];
                        const double h0 = lo0 ? (double)nv[0] : 0.0, h1 = act ? (double)nv[1] : 0.0;
                        const int rc = row < n ? row : 0;
                        acc[r][0] = __builtin_fma(h0, gj, acc[r][0]);
                        acc[r][1] = __builtin_fma(h1, gj, acc[r][1]);
                        const double m0 = row > j ? h0 : 0.0, m1 = row + 1 > j ? h1 : 0.0;
                        col[u] = __builtin_fma(m0, (double)in.g[rc], col[u]);
                        col[u] = __builtin_fma(m1, (double)in.g[rc + 1], col[u]);
                        store2(act ? H + (int64_t)j * n + row : dummy, nv);
                    }
                }
                const double tot = wave_sum8(col, ln);
                const int own = wave_sum8_owner(ln);
                const int ju = half + 2 * (c0 + own);
                if (ln < 8 && own < UJ && ju < n) wpart[wv * np + ju] = tot;
            };
            {
                T hvA[UJ][RP][2], hvB[UJ][RP][2];
                issue(0, hvA);
                for (int c0 = 0; c0 < my_cols; c0 += 2 * UJ) {
                    issue(c0 + UJ, hvB);
                    update_chunk(c0, hvA);
                    issue(c0 + 2 * UJ, hvA);
                    update_chunk(c0 + UJ, hvB);
                }
            }
            __syncthreads();   // every read of the scaled d / t above is done before d is overwritten
            finish_symmetric(acc, in.d);
            if ((debug & 2) && threadIdx.x == 0) { const long long tk = clock64(); atomicAdd(st.stats + 3, (unsigned long long)(tk - tk0)); atomicAdd(st.stats + 4, 1ull); tk0 = tk; }
        } else {
            // :981-986  H = I, d = g.  Only the lower triangle is ever read again (see above): it alone is
            // reset, with the same row-pair / column-parity ownership and 16-byte stores as the update pass.
            for (int j = half; j < n; j += 2) {
#pragma unroll
                for (int r = 0; r < RP; ++r) {
                    const int row = 2 * (lane_h + kHalf * r);
                    if (row < n && row + 1 >= j) {
                        T *p = H + (int64_t)j * n + row;
                        if constexpr (sizeof(T) == 8) { double2 q; q.x = row == j ? 1.0 : 0.0; q.y = row + 1 == j ? 1.0 : 0.0; *reinterpret_cast<double2 *>(p) = q; }
                        else { float2 q; q.x = row == j ? 1.0f : 0.0f; q.y = row + 1 == j ? 1.0f : 0.0f; *reinterpret_cast<float2 *>(p) = q; }
                    }
                }
            }
            for (int i = threadIdx.x; i < n; i += kBlock) in.d[i] = in.g[i];
            __syncthreads();
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += kBlock) { gx[i] = in.x[i]; gg[i] = in.g[i]; gd[i] = in.d[i]; gdx[i] = in.dx[i]; gdg[i] = in.dg[i]; }
    if (threadIdx.x == 0) {
        st.f[b] = (double)f;
        st.last_step_length[b] = (double)last_len;
        st.last_step_type[b] = last_type;
        st.iteration_count[b] = iters;
        st.has_terminated[b] = terminated ? 1 : 0;
    }
}

// constructor per instance (:762-810): constraint on x0 (:770), f0, g0, H0 = I, d0 = g
template <typename T>
__global__ __launch_bounds__(kBlock) void batch_init_kernel(BatchState st, double initial_step_length) {
    extern __shared__ __attribute__((aligned(16))) char init_smem[];
    __shared__ double red[4];
    BatchObjective ob = st.ob;
    const int n = (int)st.n;
    const int64_t b = blockIdx.x;
    if (ob.A) ob.A = (const T *)ob.A + b * ob.A_stride;                 // this instance's matrix
    T *x = (T *)st.x + b * n;
    T *xs = reinterpret_cast<T *>(init_smem);                  // the (projected) starting point, shared
    T *g = (T *)st.g + b * n, *d = (T *)st.d + b * n, *dx = (T *)st.dx + b * n, *dg = (T *)st.dg + b * n;
    T *H = (T *)st.H + b * (int64_t)n * n;
    for (int i = threadIdx.x; i < n; i += kBlock) {
        T xi = x[i];
        if (ob.cons_on) { xi = b_clamp<T>(xi, (T)ob.cons_lo, (T)ob.cons_hi); x[i] = xi; }   // :770
        xs[i] = xi;
    }
    __syncthreads();
    double acc = 0, ss = 0;
    if (ob.kind == DZO_PROBLEM_QUADRATIC) {
        for (int i = threadIdx.x; i < n; i += kBlock) acc = __builtin_fma(b_quad_row<T>((const T *)ob.A, n, i, xs), (double)xs[i], acc);
    } else {
        for (int i = threadIdx.x; i + 1 < n; i += kBlock) acc += b_rosen_term<T>(xs[i], xs[i + 1]);
    }
    for (int i = threadIdx.x; i < n; i += kBlock) ss = __builtin_fma((double)xs[i], (double)xs[i], ss);
    T fT = ob.kind == DZO_PROBLEM_QUADRATIC ? (T)(0.5 * bsum(acc, red)) : (T)bsum(acc, red);
    const double ssum = bsum(ss, red);
    if (ob.l2 != 0.0) fT = fT + (T)ob.l2 * (T)ssum;
    const double f = (double)fT;
    for (int i = threadIdx.x; i < n; i += kBlock) {
        T gi;
        if (ob.kind == DZO_PROBLEM_QUADRATIC) gi = (T)b_quad_row<T>((const T *)ob.A, n, i, xs);
        else gi = b_rosen_grad<T>(i, n, i > 0 ? xs[i - 1] : (T)0, xs[i], i + 1 < n ? xs[i + 1] : (T)0);
        gi = b_decorate_grad<T>(ob, gi, xs[i]);
        g[i] = gi; d[i] = gi; dx[i] = (T)0; dg[i] = (T)0;
    }
    for (int64_t e = threadIdx.x; e < (int64_t)n * n; e += kBlock) H[e] = (e / n == e % n) ? (T)1 : (T)0;
    if (threadIdx.x == 0) {
        st.f[b] = f;
        st.last_step_length[b] = (double)(T)initial_step_length;
        st.last_step_type[b] = DZO_STEP_NULL;
        st.iteration_count[b] = 0;
        st.has_terminated[b] = (f != f) ? 1 : 0;
    }
}

// upper triangle <- lower triangle (the step kernel maintains the lower one only); run before the host looks at H
template <typename T>
__global__ __launch_bounds__(kBlock) void batch_mirror_kernel(int64_t n, T *__restrict__ Hall) {
    T *H = Hall + (int64_t)blockIdx.y * n * n;
    const int64_t total = n * n;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlock) {
        const int64_t col = e / n, row = e % n;
        if (row < col) H[e] = H[row * n + col];
    }
}

__global__ __launch_bounds__(kBlock) void batch_count_active_kernel(const int32_t *__restrict__ term, int64_t batch,
                                                                    unsigned long long *__restrict__ out) {
    unsigned long long c = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < batch; i += (int64_t)gridDim.x * kBlock) c += term[i] ? 0 : 1;
    if (c) atomicAdd(out, c);   // integer: order-free
}

}  // namespace dzo

static inline int tune_env(const char *name, int dflt) { const char *v = getenv(name); return v ? atoi(v) : dflt; }
typedef void (*batch_kernel_fn)(dzo::BatchState, int, int);
static batch_kernel_fn batch_kernel_of(int32_t dtype, int rp, int wide) {
    using namespace dzo;
    if (dtype == DZO_F64) return rp == 1 ? batch_step_kernel<double, 1> : rp == 2 ? (wide ? batch_step_kernel<double, 2, 1> : batch_step_kernel<double, 2, 0>) : batch_step_kernel<double, 4>;
    return rp == 1 ? batch_step_kernel<float, 1> : rp == 2 ? (wide ? batch_step_kernel<float, 2, 1> : batch_step_kernel<float, 2, 0>) : batch_step_kernel<float, 4>;
}

struct dzo_bfgs_batch_s {
    dzo::BatchState st;
    int device = 0;                     // the shard's GPU; every entry point enters it (DeviceScope)
    int32_t dtype = DZO_F64;
    hipStream_t stream = nullptr;
    unsigned long long *count_dev = nullptr;
    unsigned long long *count_host = nullptr;
    size_t lds_bytes = 0;
    int rp = 1;
    int wide = 0;                   // RP = 2 only: the four-column, one-block-per-CU form (see batch_step_kernel)
    bool upper_stale = false;           // steps ran since the upper triangles of H were last mirrored from the lower ones
};

using namespace dzo;

namespace dzo {
int batch_device(const dzo_bfgs_batch_s *b) { return b->device; }

// count of live instances, split so that several shards can count concurrently
int32_t batch_count_active_enqueue(dzo_bfgs_batch_s *b) {
    DeviceScope scope(b->device);
    DZO_HIP(hipMemsetAsync(b->count_dev, 0, sizeof(unsigned long long), b->stream));
    int grid = (int)((b->st.batch + kBlock - 1) / kBlock);
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(batch_count_active_kernel, dim3(grid), dim3(kBlock), 0, b->stream, b->st.has_terminated, b->st.batch, b->count_dev);
    DZO_HIP(hipGetLastError());
    DZO_HIP(hipMemcpyAsync(b->count_host, b->count_dev, sizeof(unsigned long long), hipMemcpyDeviceToHost, b->stream));
    return DZO_OK;
}
int32_t batch_count_active_finish(dzo_bfgs_batch_s *b, int64_t *active) {
    DeviceScope scope(b->device);
    DZO_HIP(hipStreamSynchronize(b->stream));
    *active = (int64_t)*b->count_host;
    return DZO_OK;
}
}  // namespace dzo

extern "C" {

int32_t dzo_bfgs_batch_destroy(dzo_bfgs_batch_t b) {
    if (!b) return DZO_OK;
    DeviceScope scope(b->device);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    void *ptrs[] = {b->st.x, b->st.g, b->st.dx, b->st.dg, b->st.d, b->st.H, b->st.f, b->st.last_step_length,
                    b->st.last_step_type, b->st.has_terminated, b->st.iteration_count, b->count_dev, b->st.stats};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (b->count_host) (void)hipHostFree(b->count_host);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
    return DZO_OK;
}

static int32_t batch_create_impl(const BatchObjective &ob, int64_t batch, int64_t n, int32_t dtype, const void *x0_dev,
                                 double initial_step_length, dzo_bfgs_batch_t *out) {
    DZO_TRY(require_init());
    DZO_REQUIRE(out && x0_dev, DZO_ERR_INVALID, "null argument");
    DZO_REQUIRE(dtype == DZO_F32 || dtype == DZO_F64, DZO_ERR_INVALID, "bad dtype %d", dtype);
    const int32_t problem_kind = ob.kind;
    DZO_REQUIRE(problem_kind == DZO_PROBLEM_ROSENBROCK_CHAIN || (problem_kind == DZO_PROBLEM_ROSENBROCK2D && n == 2) ||
                    (problem_kind == DZO_PROBLEM_QUADRATIC && ob.A),
                DZO_ERR_UNSUPPORTED, "batched mode implements the (chained) Rosenbrock objective and the dense quadratic (one shared A or one A per instance)");
    DZO_REQUIRE(batch >= 1, DZO_ERR_INVALID, "batch must be >= 1");
    DZO_TRY(require_same_backend("batched BFGSOptimizer", "src/DZOptimization.jl:363-364", x0_dev, "initial_points",
                                 ob.A_stride ? ob.A : nullptr, "matrices"));     // (the caller's per-instance matrices; a shared A is the problem handle's own copy)
    DZO_REQUIRE(n >= 2 && n % 2 == 0 && n <= 1024, DZO_ERR_UNSUPPORTED,
                "batched mode needs an even n in 2..1024 (got %lld)", (long long)n);
    dzo_bfgs_batch_s *b = new dzo_bfgs_batch_s();
    b->dtype = dtype;
    b->device = ctx().device;
    b->st.ob = ob;
    if (b->st.ob.kind == DZO_PROBLEM_ROSENBROCK2D) b->st.ob.kind = DZO_PROBLEM_ROSENBROCK_CHAIN;   // n = 2: the same function
    b->st.batch = batch; b->st.n = n;
    b->rp = n <= 256 ? 1 : (n <= 512 ? 2 : 4);
    b->wide = (b->rp == 2 && n > tune_env("DZO_TUNE_BATCH_RP2_WIDE_ABOVE", 448)) ? 1 : 0;
    const size_t es = dtype_size(dtype);
    const size_t np = (size_t)((n + 1) & ~(int64_t)1);
    b->lds_bytes = 9 * np * es + (5 * np + 16) * sizeof(double) + 16;
    const size_t vb = (size_t)batch * (size_t)n * es;
    hipError_t e = hipSuccess;
    auto alloc = [&](void **p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes); };
    alloc(&b->st.x, vb); alloc(&b->st.g, vb); alloc(&b->st.dx, vb); alloc(&b->st.dg, vb); alloc(&b->st.d, vb);
    alloc(&b->st.H, vb * (size_t)n);
    alloc((void **)&b->st.f, batch * sizeof(double)); alloc((void **)&b->st.last_step_length, batch * sizeof(double));
    alloc((void **)&b->st.last_step_type, batch * sizeof(int32_t)); alloc((void **)&b->st.has_terminated, batch * sizeof(int32_t));
    alloc((void **)&b->st.iteration_count, batch * sizeof(int64_t)); alloc((void **)&b->count_dev, sizeof(unsigned long long));
    alloc((void **)&b->st.stats, 16 * sizeof(unsigned long long));   // [0..7] dev cycle counters, [8..15] dummy store target
    if (e != hipSuccess) {
        dzo_bfgs_batch_destroy(b);
        set_error("out of device memory for %lld instances of %lld x %lld", (long long)batch, (long long)n, (long long)n);
        (void)hipGetLastError(); return DZO_ERR_NOMEM;
    }
    DZO_HIP(hipHostMalloc((void **)&b->count_host, sizeof(unsigned long long), hipHostMallocDefault));
    DZO_HIP(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    DZO_HIP(hipMemset(b->st.stats, 0, 16 * sizeof(unsigned long long)));
    DZO_HIP(hipMemcpy(b->st.x, x0_dev, vb, hipMemcpyDeviceToDevice));        // :769 copy
    DZO_HIP(hipDeviceSynchronize());   // null-stream memset/D2D copies are asynchronous to the host and to our non-blocking streams
    if (b->lds_bytes > 48 * 1024) {
        // gfx950 has 160 KiB of LDS per CU; dynamic requests above the default need the attribute
        DZO_HIP(hipFuncSetAttribute((const void *)batch_kernel_of(dtype, b->rp, b->wide), hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_bytes));
    }
    {
        DZO_TIMED("bfgs_batch_init", b->stream);
        DZO_DISPATCH(dtype, hipLaunchKernelGGL(batch_init_kernel<T>, dim3((unsigned)batch), dim3(kBlock), np * es, b->stream, b->st, initial_step_length));
    }
    DZO_HIP(hipGetLastError());
    DZO_HIP(hipStreamSynchronize(b->stream));
    *out = b;
    return DZO_OK;
}

int32_t dzo_bfgs_batch_create(int32_t problem_kind, int64_t batch, int64_t n, int32_t dtype, const void *x0_dev,
                              double initial_step_length, dzo_bfgs_batch_t *out) {
    DZO_REQUIRE(problem_kind != DZO_PROBLEM_QUADRATIC, DZO_ERR_INVALID,
                "the quadratic objective needs its matrix: use dzo_bfgs_batch_create_problem");
    BatchObjective ob;
    ob.kind = problem_kind;
    return batch_create_impl(ob, batch, n, dtype, x0_dev, initial_step_length, out);
}

// device < 0: the device the calling thread selected; otherwise the shard is created on that device
static int32_t batch_create_on_device(const BatchObjective &ob, int64_t batch, int64_t n, int32_t dtype, const void *x0_dev,
                                           double initial_step_length, int32_t device, dzo_bfgs_batch_t *out) {
    if (device < 0) return batch_create_impl(ob, batch, n, dtype, x0_dev, initial_step_length, out);
    int count = 0;
    DZO_HIP(hipGetDeviceCount(&count));
    DZO_REQUIRE(device < count && device < kMaxDevices, DZO_ERR_INVALID, "device %d out of range [0,%d)", device, count);
    const int prev = ctx().ready ? ctx().device : -1;
    DZO_TRY(dzo_init(device));
    int32_t rc;
    {
        DeviceScope scope(device);
        rc = batch_create_impl(ob, batch, n, dtype, x0_dev, initial_step_length, out);
    }
    if (prev >= 0 && prev != device) (void)dzo_init(prev);
    return rc;
}

// The batched constructor from a problem handle: kind, n, dtype, the shared matrix A of the quadratic and
// the decorators (dzo_problem_set_l2 / set_box_gradient / set_box_constraint) are taken from it at
// creation time.  device < 0: the device the calling thread selected.
int32_t dzo_bfgs_batch_create_problem(dzo_problem_t problem, int64_t batch, const void *x0_dev, double initial_step_length,
                                      int32_t device, dzo_bfgs_batch_t *out) {
    DZO_REQUIRE(problem && out, DZO_ERR_INVALID, "null argument");
    DZO_REQUIRE(problem->kind != DZO_PROBLEM_LSE && problem->kind != DZO_PROBLEM_QUADRATIC_CHAIN, DZO_ERR_UNSUPPORTED,
                "batched mode implements the chained Rosenbrock and the dense quadratic objectives only");
    BatchObjective ob;
    ob.kind = problem->kind; ob.A = problem->A;
    ob.l2 = problem->l2;
    ob.bg_on = problem->bg_on ? 1 : 0; ob.bg_lo = problem->bg_lo; ob.bg_hi = problem->bg_hi;
    ob.cons_on = problem->cons_on ? 1 : 0; ob.cons_lo = problem->cons_lo; ob.cons_hi = problem->cons_hi;
    return batch_create_on_device(ob, batch, problem->n, problem->dtype, x0_dev, initial_step_length, device, out);
}

// "Run multiple optimizers in parallel" (README.md:12) with a DIFFERENT quadratic per instance: the problem handle
// gives kind (QUADRATIC), n, dtype and the decorators; instance b minimises 1/2 x'A_b x with A_b = the n x n
// column-major symmetric matrix at matrices_dev + b * matrix_stride elements (matrix_stride >= n*n; the caller owns
// the array and keeps it alive and unchanged while the batch exists).
int32_t dzo_bfgs_batch_create_problem_matrices(dzo_problem_t problem, int64_t batch, const void *matrices_dev, int64_t matrix_stride,
                                               const void *x0_dev, double initial_step_length, int32_t device, dzo_bfgs_batch_t *out) {
    DZO_REQUIRE(problem && out && matrices_dev, DZO_ERR_INVALID, "null argument");
    DZO_REQUIRE(problem->kind == DZO_PROBLEM_QUADRATIC, DZO_ERR_INVALID, "per-instance matrices belong to the quadratic objective");
    DZO_REQUIRE(matrix_stride >= problem->n * problem->n, DZO_ERR_INVALID, "matrix_stride %lld < n*n = %lld",
                (long long)matrix_stride, (long long)(problem->n * problem->n));
    BatchObjective ob;
    ob.kind = problem->kind; ob.A = matrices_dev; ob.A_stride = matrix_stride;   // (backend assert: batch_create_impl, on the shard's device)
    ob.l2 = problem->l2;
    ob.bg_on = problem->bg_on ? 1 : 0; ob.bg_lo = problem->bg_lo; ob.bg_hi = problem->bg_hi;
    ob.cons_on = problem->cons_on ? 1 : 0; ob.cons_lo = problem->cons_lo; ob.cons_hi = problem->cons_hi;
    return batch_create_on_device(ob, batch, problem->n, problem->dtype, x0_dev, initial_step_length, device, out);
}

// QuadraticLineSearch.max_increases (legacy :181-188, :138-151) of every instance; 0 = no cap
int32_t dzo_bfgs_batch_set_max_increases(dzo_bfgs_batch_t b, int32_t max_increases) {
    DZO_REQUIRE(b, DZO_ERR_INVALID, "null batch");
    DZO_REQUIRE(max_increases >= 0, DZO_ERR_INVALID, "negative max_increases");
    b->st.ob.max_increases = max_increases;
    return DZO_OK;
}

int32_t dzo_bfgs_batch_count_active(dzo_bfgs_batch_t b, int64_t *active) {
    DZO_REQUIRE(b && active, DZO_ERR_INVALID, "null argument");
    DZO_TRY(batch_count_active_enqueue(b));
    return batch_count_active_finish(b, active);
}

// The same constructor on an explicit device: a host that shards instances over the GPUs of a node
// from ONE process (dzo_comm_init_all) creates one shard per device.  x0_dev must live on `device`.
int32_t dzo_bfgs_batch_create_on(int32_t device, int32_t problem_kind, int64_t batch, int64_t n, int32_t dtype,
                                 const void *x0_dev, double initial_step_length, dzo_bfgs_batch_t *out) {
    int count = 0;
    DZO_HIP(hipGetDeviceCount(&count));
    DZO_REQUIRE(device >= 0 && device < count && device < kMaxDevices, DZO_ERR_INVALID, "device %d out of range [0,%d)", device, count);
    const int prev = ctx().ready ? ctx().device : -1;
    DZO_TRY(dzo_init(device));                          // makes sure the device has a library context
    int32_t rc;
    {
        DeviceScope scope(device);
        rc = dzo_bfgs_batch_create(problem_kind, batch, n, dtype, x0_dev, initial_step_length, out);
    }
    if (prev >= 0 && prev != device) (void)dzo_init(prev);   // the calling thread stays on the device it had selected
    return rc;
}

int32_t dzo_bfgs_batch_step(dzo_bfgs_batch_t b, int32_t steps, int32_t *all_done) {
    DZO_REQUIRE(b, DZO_ERR_INVALID, "null batch");
    DZO_REQUIRE(steps >= 0, DZO_ERR_INVALID, "negative step count");
    DeviceScope scope(b->device);
    if (steps > 0) {
        b->upper_stale = true;
        DZO_TIMED("bfgs_batch_step", b->stream);
        const dim3 grid((unsigned)b->st.batch), block(kBlock);
        hipLaunchKernelGGL(batch_kernel_of(b->dtype, b->rp, b->wide), grid, block, b->lds_bytes, b->stream, b->st, (int)steps,
                           getenv("DZO_TUNE_BATCH_DEBUG") ? atoi(getenv("DZO_TUNE_BATCH_DEBUG")) : 0);
    }
    DZO_HIP(hipGetLastError());
    if (all_done) {
        int64_t active = 0;
        DZO_TRY(dzo_bfgs_batch_count_active(b, &active));
        *all_done = active == 0 ? 1 : 0;
    }
    return DZO_OK;
}

int32_t dzo_bfgs_batch_get_ptr(dzo_bfgs_batch_t b, int32_t what, void **ptr_dev) {
    DZO_REQUIRE(b && ptr_dev, DZO_ERR_INVALID, "null argument");
    DeviceScope scope(b->device);
    if (what == 2 && b->upper_stale) {                   // the step kernel keeps the lower triangle of every H only
        const int64_t n = b->st.n;
        int gx = (int)((n * n + kBlock - 1) / kBlock);
        if (gx > 64) gx = 64;
        DZO_DISPATCH(b->dtype, hipLaunchKernelGGL(batch_mirror_kernel<T>, dim3((unsigned)gx, (unsigned)b->st.batch), dim3(kBlock), 0, b->stream,
                                                  n, (T *)b->st.H));
        DZO_HIP(hipGetLastError());
        b->upper_stale = false;
    }
    DZO_HIP(hipStreamSynchronize(b->stream));
    switch (what) {
    case 0: *ptr_dev = b->st.x; break;
    case 1: *ptr_dev = b->st.g; break;
    case 2: *ptr_dev = b->st.H; break;
    case 3: *ptr_dev = b->st.f; break;
    case 4: *ptr_dev = b->st.has_terminated; break;
    case 5: *ptr_dev = b->st.iteration_count; break;
    case 6: *ptr_dev = b->st.dx; break;
    case 7: *ptr_dev = b->st.dg; break;
    case 8: *ptr_dev = b->st.d; break;
    case 9: *ptr_dev = b->st.last_step_length; break;
    case 10: *ptr_dev = b->st.last_step_type; break;
    case 99: *ptr_dev = b->st.stats; break;             // dev only
    default: set_error("dzo_bfgs_batch_get_ptr: unknown field %d", what); return DZO_ERR_INVALID;
    }
    return DZO_OK;
}

int32_t dzo_bfgs_batch_device(dzo_bfgs_batch_t b, int32_t *device) {
    DZO_REQUIRE(b && device, DZO_ERR_INVALID, "null argument");
    *device = b->device;
    return DZO_OK;
}

}  // extern "C"
