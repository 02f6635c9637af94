// dzo_bfgs.hip -- dense BFGSOptimizer + step! (legacy/DZOptimization.jl:733-994) for gfx950.
//
// H is n x n, column-major, full storage (:746).  The rank-2 update (:878-886) keeps H exactly
// symmetric bit for bit (t_i*d_j + d_i*t_j is commutative in IEEE arithmetic), so H*v is
// computed as per-column dot products t_j = H[:,j].v: coalesced along the column, one
// workgroup per group of 4 columns, no cross-workgroup reduction, deterministic.
//
// update_inverse_hessian! + the following mul! (:953-960) run as three launches:
//   A  symv       t = H*dg                              reads  n^2
//   B  scalars    overlap = d.dg, d' = d/overlap, delta = lambda*overlap + dg.t   (one block)
//   C  update     H[:,j] += delta*(d'_i d'_j) - (t_i d'_j + d'_i t_j)  AND  d_next_j = H+[:,j].g
//                                                       reads n^2, writes n^2
// = 3*n^2 elements, the algorithmic floor of SURVEY.md 8(d) (the reference's pass structure
// moves 4*n^2).  The update expression is evaluated exactly as written at :882-884 (no fma
// contraction), so H matches the oracle bit for bit given equal scalars.
//
// MFMA note: the update is a rank-2 contraction H += [d' t] * [delta*d'-t, -d']^T and maps
// onto v_mfma_f64_16x16x4_f64 with K padded 2->4, but at 0.4 flop/byte the kernel is bound by
// HBM, not by the 2 fma per element; the VALU form below also keeps the reference's rounding
// order, which the MFMA accumulate order would not.  See DESIGN.md.
#include <atomic>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <utility>

#include "dzo_optcore.h"

namespace dzo {

constexpr int kColsPerBlock = 4;     // default column group per block (symv / update)

static int bfgs_cols_knob() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("DZO_TUNE_BFGS_COLS"); v = e ? atoi(e) : kColsPerBlock; if (v != 8 && v != 16) v = 4; }
    return v;
}

// t_j = H[:,j] . v for a group of columns per block
// With `dvec` non-null the kernel also emits, per column group, the partial sums of
// overlap = d.v (:873) and v.t (:876) over its own columns (v = delta_gradient, t = H*v), so the
// update kernel can form the scalars itself and the single-block scalars launch disappears.
template <typename T, bool VEC, int C>
__global__ __launch_bounds__(kBlock) void symv_kernel(int64_t n, const T *__restrict__ H, const T *__restrict__ v,
                                                      T *__restrict__ out, const T *__restrict__ dvec,
                                                      double *__restrict__ part_ov, double *__restrict__ part_vt) {
    constexpr int N = VEC ? Vec16<T>::N : 1;
    __shared__ double lds[kWaves];
    const int64_t groups = (n + C - 1) / C;
    for (int64_t grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const int64_t j0 = grp * C;
        double acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 0;
        for (int64_t i = (int64_t)threadIdx.x * N; i < n; i += (int64_t)kBlock * N) {
            T vv[N];
            if constexpr (VEC) load16(v + i, vv); else vv[0] = v[i];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if (j0 + c < n) {
                    T hv[N];
                    if constexpr (VEC) load16(H + (j0 + c) * n + i, hv); else hv[0] = H[(j0 + c) * n + i];
#pragma unroll
                    for (int q = 0; q < N; ++q) acc[c] = __builtin_fma((double)hv[q], (double)vv[q], acc[c]);
                }
            }
        }
        double pov = 0, pvt = 0;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const double r = block_sum(acc[c], lds);
            if (threadIdx.x == 0 && j0 + c < n) {
                const T tj = (T)r;
                out[j0 + c] = tj;
                if (dvec) {
                    pov = __builtin_fma((double)dvec[j0 + c], (double)v[j0 + c], pov);
                    pvt = __builtin_fma((double)v[j0 + c], (double)tj, pvt);
                }
            }
        }
        if (dvec && threadIdx.x == 0) { part_ov[grp] = pov; part_vt[grp] = pvt; }
    }
}

// one block: overlap = d.dg (:873); d *= inv(overlap) (:874); delta = lambda*overlap + dg.t (:876)
template <typename T>
__global__ __launch_bounds__(kBlock) void bfgs_scalars_kernel(int64_t n, T *__restrict__ d, const T *__restrict__ dg,
                                                              const T *__restrict__ t, T lambda,
                                                              double *__restrict__ scalars) {
    __shared__ double lds[kWaves];
    __shared__ double bc[2];
    double a = 0, b = 0;
    for (int64_t i = threadIdx.x; i < n; i += kBlock) {
        a = __builtin_fma((double)d[i], (double)dg[i], a);
        b = __builtin_fma((double)dg[i], (double)t[i], b);
    }
    const double ra = block_sum(a, lds);
    const double rb = block_sum(b, lds);
    if (threadIdx.x == 0) {
        const T overlap = (T)ra;
        const T delta = lambda * overlap + (T)rb;
        bc[0] = (double)((T)1 / overlap);
        scalars[0] = (double)overlap;
        scalars[1] = (double)delta;
    }
    __syncthreads();
    const T inv = (T)bc[0];
    for (int64_t i = threadIdx.x; i < n; i += kBlock) d[i] = d[i] * inv;
}

// H[:,j] update (:878-886) fused with d_next_j = H_new[:,j] . g (:958-960)
// FUSED: `dp` is the UNSCALED direction; every block first sums the per-group partials of
// overlap and dg.t written by symv_kernel (fixed order, identical in every block), forms
// inv = 1/overlap (:874) and delta (:876) itself, and scales d on the fly -- the same rounded
// values the reference stores before using them.
template <typename T, bool VEC, bool DIRECTION, bool FUSED, int C>
__global__ __launch_bounds__(kBlock) void bfgs_update_kernel(int64_t n, T *__restrict__ H, const T *__restrict__ dp,
                                                             const T *__restrict__ t, const double *__restrict__ scalars,
                                                             const T *__restrict__ g, T *__restrict__ d_next,
                                                             const double *__restrict__ part_ov,
                                                             const double *__restrict__ part_vt, int nparts, T lambda) {
    constexpr int N = VEC ? Vec16<T>::N : 1;
    __shared__ double lds[kWaves];
    T delta, inv = (T)1;
    if constexpr (FUSED) {
        const T overlap = (T)reduce_partials_all(part_ov, nparts, lds);
        const T dgt = (T)reduce_partials_all(part_vt, nparts, lds);
        inv = (T)1 / overlap;
        delta = lambda * overlap + dgt;
    } else {
        delta = (T)scalars[1];
    }
    const int64_t groups = (n + C - 1) / C;
    for (int64_t grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const int64_t j0 = grp * C;
        T sj[C], tj[C];
        double acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int64_t j = j0 + c < n ? j0 + c : n - 1;
            sj[c] = FUSED ? dp[j] * inv : dp[j];             // :879 (:874 applied on the fly when FUSED)
            tj[c] = t[j];                                    // :880
            acc[c] = 0;
        }
        for (int64_t i = (int64_t)threadIdx.x * N; i < n; i += (int64_t)kBlock * N) {
            T di[N], ti[N], gi[N];
            if constexpr (VEC) { load16(dp + i, di); load16(t + i, ti); if (DIRECTION) load16(g + i, gi); }
            else { di[0] = dp[i]; ti[0] = t[i]; if (DIRECTION) gi[0] = g[i]; }
            if constexpr (FUSED) {
#pragma unroll
                for (int q = 0; q < N; ++q) di[q] = di[q] * inv;
            }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if (j0 + c < n) {
                    T *col = H + (j0 + c) * n + i;
                    T hv[N];
                    if constexpr (VEC) load16(col, hv); else hv[0] = col[0];
#pragma unroll
                    for (int q = 0; q < N; ++q) {
                        // :882-884, evaluated in the reference's order (no contraction)
                        hv[q] = hv[q] + (delta * (di[q] * sj[c]) - (ti[q] * sj[c] + di[q] * tj[c]));
                        if (DIRECTION) acc[c] = __builtin_fma((double)hv[q], (double)gi[q], acc[c]);
                    }
                    if constexpr (VEC) store16(col, hv); else col[0] = hv[0];
                }
            }
        }
        if (DIRECTION) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const double r = block_sum(acc[c], lds);
                if (threadIdx.x == 0 && j0 + c < n) d_next[j0 + c] = (T)r;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// MFMA form of the rank-2 update (fp64, n % 16 == 0): H += U V^T with U = [d' t 0 0] and
// V = [delta*d'-t, -d', 0, 0], one v_mfma_f64_16x16x4_f64 per 16x16 tile (K = 2 padded to 4).
// The tile is held TRANSPOSED in the accumulator (C/D column index = lane & 15 runs along the
// memory-contiguous row index of the column-major H), so a load instruction touches four 128-B
// column segments.  Kept as a measured alternative, off by default.  Measured at n = 4096
// (profiles/r01_bench_bfgs_dense.json): 44 us for the 2 n^2 T of the H update alone (6.1 TB/s out
// of the Infinity Cache) against 53 us for the VALU kernel, which in the same pass also forms the
// next direction H_new*g.  The MFMA form cannot fuse that direction without cross-lane sums, so
// step! would need a second symv (+28 us): 101 us against 82 us per update.  It also rounds
// differently from the reference's expression (:882-884), and H_ij / H_ji are no longer
// bit-identical.  See DESIGN.md section 4.
// ---------------------------------------------------------------------------------------------
typedef double mfma_v4f64 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(kBlock) void bfgs_update_mfma_kernel(int64_t n, double *__restrict__ H,
                                                                  const double *__restrict__ dp,
                                                                  const double *__restrict__ t,
                                                                  const double *__restrict__ scalars) {
    const double delta = scalars[1];
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * kWaves + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * kWaves;
    const int64_t tiles = n / 16;                       // per dimension
    const int il = lane & 15, kq = lane >> 4;
    // a wave owns a strip of 16 rows (i-tile) and walks the column tiles 4 at a time
    for (int64_t job = wave; job < tiles * ((tiles + 3) / 4); job += nwaves) {
        const int64_t it = job % tiles, jq = job / tiles;
        const int64_t i0 = it * 16;
        // B[k][col = il] = U[i0 + il][k]
        const double di = dp[i0 + il], ti = t[i0 + il];
        const double bval = kq == 0 ? di : (kq == 1 ? ti : 0.0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t jt = jq * 4 + u;
            if (jt >= tiles) break;
            const int64_t j0 = jt * 16;
            // A[row = il][k] = V[j0 + il][k]
            const double dj = dp[j0 + il], tj = t[j0 + il];
            const double aval = kq == 0 ? (delta * dj - tj) : (kq == 1 ? -dj : 0.0);
            mfma_v4f64 c;
            double *base = H + (i0 + il) + (j0 + kq) * n;      // D[row = kq + 4r][col = il] = H[i0+il][j0+kq+4r]
            c.x = base[0];
            c.y = base[4 * n];
            c.z = base[8 * n];
            c.w = base[12 * n];
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(aval, bval, c, 0, 0, 0);
            base[0] = c.x;
            base[4 * n] = c.y;
            base[8 * n] = c.z;
            base[12 * n] = c.w;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void identity_kernel(int64_t n, T *__restrict__ H) {
    const int64_t total = n * n;
    const int64_t nthreads = (int64_t)gridDim.x * kBlock;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += nthreads)
        H[e] = (e / n == e % n) ? (T)1 : (T)0;              // :712-720
}

// scratch = fma(t_signed, dir, x) with the two bracket flags (:71-80):
//   flags[0] |= any(x != new)   ("point_changed"),  flags[1] |= any(dir != 0)  ("!step_is_zero")
template <typename T>
__global__ __launch_bounds__(kBlock) void phi_point_kernel(int64_t n, T *__restrict__ dst, T t, const T *__restrict__ dir,
                                                           const T *__restrict__ x, int32_t *__restrict__ flags) {
    const int64_t nthreads = (int64_t)gridDim.x * kBlock;
    bool changed = false, nonzero = false;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += nthreads) {
        const T xi = x[i], di = dir[i];
        const T nw = dfma(t, di, xi);                            // t carries the sign: -t for BFGS (:945), +t legacy (:33)
        dst[i] = nw;
        changed |= (xi != nw);
        nonzero |= (di != (T)0);
    }
    if (!flags) return;                                          // only the bracket's first point needs them
    __shared__ int lds_flag;
    block_raise_flag(changed, flags, &lds_flag);
    __syncthreads();
    block_raise_flag(nonzero, flags + 1, &lds_flag);
}

// move (:943-945): dx = x_old, dg = g_old (un-negated backups), x = fma(-t, dir, x)
template <typename T>
__global__ __launch_bounds__(kBlock) void bfgs_move_kernel(int64_t n, T *__restrict__ x, const T *dir, const T *g,
                                                           T t, T *__restrict__ dx, T *__restrict__ dg) {
    const int64_t nthreads = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += nthreads) {
        const T xo = x[i], go = g[i], di = dir[i];
        dx[i] = xo;
        dg[i] = go;
        x[i] = dfma(-t, di, xo);
    }
}

// move + deltas in one launch when the gradient at the new point is already known (it is a
// by-product of the line search, bfgs_dual_search): x = fma(-t, dir, x) (:945), delta_point =
// x_new - x_old (:949), g = g_new (:948), delta_gradient = g_new - g_old (:950)
template <typename T>
__global__ __launch_bounds__(kBlock) void bfgs_move_with_gradient_kernel(int64_t n, T *x, const T *dir, T *g, const T *__restrict__ gnew,
                                                                         T t, T *__restrict__ dx, T *__restrict__ dg) {
    const int64_t nthreads = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += nthreads) {
        const T xo = x[i], go = g[i], di = dir[i], gn = gnew[i];      // (dir may alias g: both read before any store)
        const T xn = dfma(-t, di, xo);
        x[i] = xn;
        dx[i] = xn - xo;
        g[i] = gn;
        dg[i] = gn - go;
    }
}

static inline bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// run `body` with the constexpr column-group width CC selected by DZO_TUNE_BFGS_COLS
#define DZO_BFGS_COLS(body)                                                     \
    switch (bfgs_cols_knob()) {                                                 \
    case 16: { constexpr int CC = 16; body; } break;                            \
    case 8: { constexpr int CC = 8; body; } break;                              \
    default: { constexpr int CC = 4; body; } break;                             \
    }

template <typename T> void launch_symv(hipStream_t s, int64_t n, const T *H, const T *v, T *out) {
    DZO_TIMED("bfgs_symv", s);
    const bool vec = (n % Vec16<T>::N == 0) && al16(H) && al16(v);
    const int cols = bfgs_cols_knob();
    const int64_t groups = (n + cols - 1) / cols;
    const int grid = (int)(groups < 65535 ? groups : 65535);
    DZO_BFGS_COLS(
        if (vec) hipLaunchKernelGGL((symv_kernel<T, true, CC>), dim3(grid), dim3(kBlock), 0, s, n, H, v, out, (const T *)nullptr, (double *)nullptr, (double *)nullptr);
        else hipLaunchKernelGGL((symv_kernel<T, false, CC>), dim3(grid), dim3(kBlock), 0, s, n, H, v, out, (const T *)nullptr, (double *)nullptr, (double *)nullptr));
}

// update_inverse_hessian! + next direction in TWO launches (step! path): the scalars ride on
// the symv epilogue / update prologue.  `d` is left unscaled (step! overwrites it anyway, :958).
// part needs 2 * ceil(n / kColsPerBlock) doubles.
template <typename T>
void launch_bfgs_update_fused(hipStream_t s, int64_t n, T *H, T lambda, const T *d, const T *dg, T *scratch, const T *g,
                              T *d_next, double *part) {
    const bool vec = (n % Vec16<T>::N == 0) && al16(H) && al16(d) && al16(dg) && al16(scratch) && al16(g);
    const int cols = bfgs_cols_knob();
    const int64_t groups = (n + cols - 1) / cols;
    const int grid = (int)(groups < 65535 ? groups : 65535);
    double *part_ov = part, *part_vt = part + groups;
    {
        DZO_TIMED("bfgs_symv", s);
        DZO_BFGS_COLS(
            if (vec) hipLaunchKernelGGL((symv_kernel<T, true, CC>), dim3(grid), dim3(kBlock), 0, s, n, (const T *)H, dg, scratch, d, part_ov, part_vt);
            else hipLaunchKernelGGL((symv_kernel<T, false, CC>), dim3(grid), dim3(kBlock), 0, s, n, (const T *)H, dg, scratch, d, part_ov, part_vt));
    }
    {
        DZO_TIMED("bfgs_update", s);
        DZO_BFGS_COLS(
            if (vec) hipLaunchKernelGGL((bfgs_update_kernel<T, true, true, true, CC>), dim3(grid), dim3(kBlock), 0, s, n, H, d, (const T *)scratch, (const double *)nullptr, g, d_next, (const double *)part_ov, (const double *)part_vt, (int)groups, lambda);
            else hipLaunchKernelGGL((bfgs_update_kernel<T, false, true, true, CC>), dim3(grid), dim3(kBlock), 0, s, n, H, d, (const T *)scratch, (const double *)nullptr, g, d_next, (const double *)part_ov, (const double *)part_vt, (int)groups, lambda));
    }
}

// update_inverse_hessian! (:864-889) + optional next direction; all scalars stay on device
template <typename T>
void launch_bfgs_update(hipStream_t s, int64_t n, T *H, T lambda, T *d, const T *dg, T *scratch, const T *g,
                        T *d_next, double *scalars_dev) {
    launch_symv<T>(s, n, H, dg, scratch);                                                 // :875
    {
        DZO_TIMED("bfgs_scalars", s);
        hipLaunchKernelGGL(bfgs_scalars_kernel<T>, dim3(1), dim3(kBlock), 0, s, n, d, dg, (const T *)scratch, lambda,
                           scalars_dev);                                                  // :873-876
    }
    {
        DZO_TIMED("bfgs_update", s);
        const bool vec = (n % Vec16<T>::N == 0) && al16(H) && al16(d) && al16(scratch) && (!g || al16(g));
        const int cols = bfgs_cols_knob();
        const int64_t groups = (n + cols - 1) / cols;
        const int grid = (int)(groups < 65535 ? groups : 65535);
        const bool dir = g != nullptr && d_next != nullptr;
#define L(V, D) hipLaunchKernelGGL((bfgs_update_kernel<T, V, D, false, CC>), dim3(grid), dim3(kBlock), 0, s, n, H, (const T *)d, (const T *)scratch, (const double *)scalars_dev, g, d_next, (const double *)nullptr, (const double *)nullptr, 0, (T)0)
        DZO_BFGS_COLS(
            if (vec) { if (dir) L(true, true); else L(true, false); }
            else { if (dir) L(false, true); else L(false, false); })
#undef L
    }
}


// ---------------------------------------------------------------------------------------------
// Lower-triangle form of update_inverse_hessian! + the following mul! (step! path; even n with H of 128 MiB
// or more by default).  H is symmetric bit for bit (the update expression :882-884 commutes in i and j), so the
// step reads and writes its LOWER triangle only: 1.5 n^2 T per update instead of 3 n^2 T; the upper
// triangle in memory is stale until somebody asks for H (dzo_bfgs_get_ptr mirrors it first).
//
// Tiling: a workgroup owns a panel of kTriPH rows x a window of kTriCW columns; a thread owns one row
// pair (16-B accesses, coalesced down each column) and every second column of the window.  A symmetric
// product u = H v splits into
//     row part   u_i += H[i,j] v_j   (j <= i)   thread-local           -> rowpart[window][i]
//     col part   u_j += H[i,j] v_i   (i >  j)   summed across the wave -> colpart[panel][j]
// and a small second kernel adds, for every i, the windows left of the diagonal and the panels below it
// in a fixed order.  Loads are unconditional (masked lanes read the first 16 bytes of H) and issued a
// chunk of eight columns ahead -- see the batched kernel, dzo_batch.hip, for the reasons.
// ---------------------------------------------------------------------------------------------
constexpr int kTriPH = 256;          // rows per panel  (= 2 * 128 row pairs)
constexpr int kTriCW = 32;           // columns per window (16 per parity)
constexpr int kTriUJ = 8;

template <typename T> __device__ __forceinline__ void tri_load2(const T *p, T (&hv)[2]) {
    if constexpr (sizeof(T) == 8) { const double2 q = *reinterpret_cast<const double2 *>(p); hv[0] = q.x; hv[1] = q.y; }
    else { const float2 q = *reinterpret_cast<const float2 *>(p); hv[0] = q.x; hv[1] = q.y; }
}
template <typename T> __device__ __forceinline__ void tri_store2(T *p, const T (&hv)[2]) {
    if constexpr (sizeof(T) == 8) { double2 q; q.x = hv[0]; q.y = hv[1]; *reinterpret_cast<double2 *>(p) = q; }
    else { float2 q; q.x = hv[0]; q.y = hv[1]; *reinterpret_cast<float2 *>(p) = q; }
}

// UPDATE = false: u = H v (v = delta_gradient, :875).  UPDATE = true: the rank-2 update of the triangle
// (:878-886) and u = Hnew v (v = gradient, :958-960); `dp` is the UNSCALED direction, the scalars of
// :873-876 are formed from the per-block partials the reduce kernel left behind.
template <typename T, bool UPDATE>
__global__ __launch_bounds__(kBlock) void tri_pass_kernel(int64_t n, T *__restrict__ H, const T *__restrict__ v,
                                                          double *__restrict__ rowpart, double *__restrict__ colpart,
                                                          const T *__restrict__ dp, const T *__restrict__ tvec,
                                                          const double *__restrict__ part_ov, const double *__restrict__ part_vt,
                                                          int nparts, T lambda, T *__restrict__ dummy) {
    // (windows from the right: the blocks right of the diagonal leave at once, the panel's eight diagonal tiles -- the slow
    // form -- start first and the interior ones fill in behind them, instead of the slow tiles being every panel's tail)
    const int P = blockIdx.y, W = (int)gridDim.x - 1 - (int)blockIdx.x;
    if ((int64_t)W * kTriCW >= ((int64_t)P + 1) * kTriPH || (int64_t)W * kTriCW >= n) return;   // window right of the panel's diagonal
    __shared__ double lds[kWaves];
    __shared__ double wp[kWaves][kTriCW];
    __shared__ double rp[kTriPH];
    const int half = threadIdx.x / 128, lane_h = threadIdx.x % 128;
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    const int64_t row = (int64_t)P * kTriPH + 2 * lane_h;
    const int64_t c_first = (int64_t)W * kTriCW;
    T delta = (T)0, inv = (T)1;
    T di[2] = {(T)0, (T)0}, ti[2] = {(T)0, (T)0};
    if constexpr (UPDATE) {
        const T overlap = (T)reduce_partials_all(part_ov, nparts, lds);     // :873
        const T dgt = (T)reduce_partials_all(part_vt, nparts, lds);
        inv = (T)1 / overlap;                                               // :874
        delta = lambda * overlap + dgt;                                     // :876
        if (row < n) { di[0] = dp[row] * inv; di[1] = dp[row + 1] * inv; ti[0] = tvec[row]; ti[1] = tvec[row + 1]; }
    }
    const double v0 = row < n ? (double)v[row] : 0.0, v1 = row < n ? (double)v[row + 1] : 0.0;
    double acc[2] = {0, 0};
    // INTERIOR tiles -- the whole window left of the panel's first row and the whole panel inside the matrix: 968 of the
    // 1096 tiles at n = 4096 -- need none of the predicates below: every element is active and strictly lower.  The
    // generic form spends some thirty instructions per load on the selects and the 64-bit address of `act ? ... : H`
    // (1746 instructions per tile and wave: half the pass's time was instruction issue); here a column is one pointer
    // increment away from the last.  Same operations on the same operands.
    const bool interior = c_first + kTriCW <= (int64_t)P * kTriPH && ((int64_t)P + 1) * kTriPH <= n;
    auto issue = [&](int c0, T (&hv)[kTriUJ][2], auto inner) {
        constexpr bool IN = decltype(inner)::value;
        if constexpr (IN) {
            const T *pcol = H + (c_first + half + 2 * c0) * n + row;
#pragma unroll
            for (int u = 0; u < kTriUJ; ++u) { tri_load2<T>(pcol, hv[u]); pcol += 2 * n; }
        } else {
#pragma unroll
            for (int u = 0; u < kTriUJ; ++u) {
                const int64_t j = c_first + half + 2 * (c0 + u);
                const bool act = c0 + u < kTriCW / 2 && j < n && row < n && row + 1 >= j;
                tri_load2<T>(act ? H + j * n + row : H, hv[u]);
            }
        }
    };
    auto chunk = [&](int c0, const T (&hv)[kTriUJ][2], auto inner) {
        constexpr bool IN = decltype(inner)::value;
        double col[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        T *pcol = H + (c_first + half + 2 * c0) * n + row;                  // (interior form only)
#pragma unroll
        for (int u = 0; u < kTriUJ; ++u) {
            const int64_t j = c_first + half + 2 * (c0 + u);
            const bool inw = IN || (c0 + u < kTriCW / 2 && j < n);
            const bool act = IN || (inw && row < n && row + 1 >= j);
            const bool lo0 = IN || (act && row >= j);                      // (row == j - 1: an upper element, left alone)
            // (a column is the same for the whole wave -- `half` is: its three vector elements come through the scalar cache
            // instead of as 48 more vector loads per thread behind the tile's own)
            const int64_t jc = __builtin_amdgcn_readfirstlane((int)(inw ? j : 0));
            const double vj = inw ? (double)v[jc] : 0.0;
            T nv[2] = {hv[u][0], hv[u][1]};
            if constexpr (UPDATE) {
                const T sj = dp[jc] * inv, tj = tvec[jc];                  // :879-880
                nv[0] = nv[0] + (delta * (di[0] * sj) - (ti[0] * sj + di[0] * tj));   // :882-884
                nv[1] = nv[1] + (delta * (di[1] * sj) - (ti[1] * sj + di[1] * tj));
                if (!lo0) nv[0] = hv[u][0];
                if constexpr (IN) { tri_store2<T>(pcol, nv); pcol += 2 * n; }
                else tri_store2<T>(act ? H + j * n + row : dummy, nv);
            }
            const double h0 = lo0 ? (double)nv[0] : 0.0, h1 = act ? (double)nv[1] : 0.0;
            acc[0] = __builtin_fma(h0, vj, acc[0]);
            acc[1] = __builtin_fma(h1, vj, acc[1]);
            const double m0 = (IN || row > j) ? h0 : 0.0, m1 = (IN || row + 1 > j) ? h1 : 0.0;   // strictly lower: the mirror part
            col[u] = __builtin_fma(m0, v0, col[u]);
            col[u] = __builtin_fma(m1, v1, col[u]);
        }
        const double tot = wave_sum8(col, ln);
        const int cw = 2 * (c0 + wave_sum8_owner(ln)) + half;             // column within the window
        if (ln < 8 && cw < kTriCW) wp[wv][cw] = tot;
    };
    {
        T hvA[kTriUJ][2], hvB[kTriUJ][2];
        if (interior) {                                                     // (uniform)
            issue(0, hvA, std::true_type{});
            issue(kTriUJ, hvB, std::true_type{});
            chunk(0, hvA, std::true_type{});
            chunk(kTriUJ, hvB, std::true_type{});
        } else {
            issue(0, hvA, std::false_type{});
            issue(kTriUJ, hvB, std::false_type{});
            chunk(0, hvA, std::false_type{});
            chunk(kTriUJ, hvB, std::false_type{});
        }
    }
    // row part: the two column parities of a row pair; column part: the two waves of a parity
    if (half == 1) { rp[2 * lane_h] = acc[0]; rp[2 * lane_h + 1] = acc[1]; }
    __syncthreads();
    if (half == 0 && row < n) {
        double *dst = rowpart + (int64_t)W * n + row;
        dst[0] = acc[0] + rp[2 * lane_h];
        dst[1] = acc[1] + rp[2 * lane_h + 1];
    }
    if (threadIdx.x < kTriCW) {
        const int cw = threadIdx.x, hw = cw & 1;                          // parity hw lives in waves 2 hw, 2 hw + 1
        const int64_t j = c_first + cw;
        if (j < n) colpart[(int64_t)P * n + j] = wp[2 * hw][cw] + wp[2 * hw + 1][cw];
    }
}

// u_i = sum of the windows left of (and on) the diagonal + the panels on and below it, fixed order.  A
// block owns kTriRI consecutive i; its 256 threads are kTriRG groups that each take every kTriRG-th window
// (and panel), combined through LDS in group order.  With `dvec` (the symv of update_inverse_hessian!) the
// block also leaves the partial sums of overlap = d.v (:873) and v.u (:876) for the update pass's prologue.
constexpr int kTriRI = 32, kTriRG = kBlock / kTriRI;
template <typename T>
__global__ __launch_bounds__(kBlock) void tri_reduce_kernel(int64_t n, const double *__restrict__ rowpart,
                                                            const double *__restrict__ colpart, T *__restrict__ out,
                                                            const T *__restrict__ dvec, const T *__restrict__ v,
                                                            double *__restrict__ part_ov, double *__restrict__ part_vt) {
    __shared__ double grp[kTriRG][kTriRI];
    __shared__ double pr[2][kTriRI];
    const int li = threadIdx.x % kTriRI, gq = threadIdx.x / kTriRI;
    const int64_t i = (int64_t)blockIdx.x * kTriRI + li;
    const int64_t np = (n + kTriPH - 1) / kTriPH;
    double a = 0;
    if (i < n) {
        // (eight loads in flight, then the adds in the plain loop's order: the loop was a chain of up to sixteen dependent
        // load-add steps, 5.4 us for 2.5 MB)
        const int64_t wmax = i / kTriCW;
        for (int64_t w0 = gq; w0 <= wmax; w0 += 8 * kTriRG) {
            double t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t w = w0 + (int64_t)u * kTriRG;
                t[u] = rowpart[(w <= wmax ? w : wmax) * n + i];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (w0 + (int64_t)u * kTriRG <= wmax) a += t[u];
        }
        for (int64_t p = i / kTriPH + gq; p < np; p += kTriRG) a += colpart[p * n + i];
    }
    grp[gq][li] = a;
    __syncthreads();
    if (gq == 0) {
        double u = 0;
#pragma unroll
        for (int q = 0; q < kTriRG; ++q) u += grp[q][li];
        const T ui = (T)u;
        double pov = 0, pvt = 0;
        if (i < n) {
            out[i] = ui;
            if (dvec) { pov = (double)dvec[i] * (double)v[i]; pvt = (double)v[i] * (double)ui; }
        }
        pr[0][li] = pov; pr[1][li] = pvt;
    }
    __syncthreads();
    if (dvec && threadIdx.x < 2) {
        double r = 0;
        for (int q = 0; q < kTriRI; ++q) r += pr[threadIdx.x][q];
        (threadIdx.x == 0 ? part_ov : part_vt)[blockIdx.x] = r;
    }
}

// upper triangle <- lower triangle, before the host (or a full-storage kernel) looks at H
template <typename T>
__global__ __launch_bounds__(kBlock) void tri_mirror_kernel(int64_t n, T *__restrict__ H) {
    const int64_t total = n * n;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlock) {
        const int64_t col = e / n, row = e % n;
        if (row < col) H[e] = H[row * n + col];
    }
}

// update_inverse_hessian!(H, lambda, d, dg, scratch) + d_next = Hnew g on the lower triangle: four launches.
// part: 2 * ceil(n / kTriRI) doubles; rowpart: ceil(n / kTriCW) * n doubles; colpart: ceil(n / kTriPH) * n.
template <typename T>
void launch_bfgs_update_tri(hipStream_t s, int64_t n, T *H, T lambda, const T *d, const T *dg, T *scratch, const T *g,
                            T *d_next, double *part, double *rowpart, double *colpart, T *dummy) {
    const dim3 grid((unsigned)((n + kTriCW - 1) / kTriCW), (unsigned)((n + kTriPH - 1) / kTriPH));
    const int rblocks = (int)((n + kTriRI - 1) / kTriRI);
    double *part_ov = part, *part_vt = part + rblocks;
    {
        DZO_TIMED("bfgs_symv", s);
        hipLaunchKernelGGL((tri_pass_kernel<T, false>), grid, dim3(kBlock), 0, s, n, H, dg, rowpart, colpart, (const T *)nullptr,
                           (const T *)nullptr, (const double *)nullptr, (const double *)nullptr, 0, (T)0, dummy);
    }
    {
        DZO_TIMED("bfgs_tri_reduce", s);
        hipLaunchKernelGGL(tri_reduce_kernel<T>, dim3(rblocks), dim3(kBlock), 0, s, n, (const double *)rowpart, (const double *)colpart,
                           scratch, d, dg, part_ov, part_vt);
    }
    {
        DZO_TIMED("bfgs_update", s);
        hipLaunchKernelGGL((tri_pass_kernel<T, true>), grid, dim3(kBlock), 0, s, n, H, g, rowpart, colpart, d, (const T *)scratch,
                           (const double *)part_ov, (const double *)part_vt, rblocks, lambda, dummy);
    }
    {
        DZO_TIMED("bfgs_tri_reduce", s);
        hipLaunchKernelGGL(tri_reduce_kernel<T>, dim3(rblocks), dim3(kBlock), 0, s, n, (const double *)rowpart, (const double *)colpart,
                           d_next, (const T *)nullptr, (const T *)nullptr, (double *)nullptr, (double *)nullptr);
    }
}

}  // namespace dzo

namespace dzo {
// sums of squares of two short vectors by one block (fixed order: thread-strided, then the block tree), results
// and a ticket into pinned host memory
template <typename T>
__global__ __launch_bounds__(kBlock) void norm2_pair_kernel(int64_t n, const T *__restrict__ a, const T *__restrict__ b,
                                                            double *__restrict__ out, double ticket) {
    __shared__ double lds[kWaves];
    double sa = 0, sb = 0;
    for (int64_t i = threadIdx.x; i < n; i += kBlock) {
        const double va = (double)a[i], vb = (double)b[i];
        sa = __builtin_fma(va, va, sa);
        sb = __builtin_fma(vb, vb, sb);
    }
    const double ra = block_sum(sa, lds);
    const double rb = block_sum(sb, lds);
    if (threadIdx.x == 0) {
        out[0] = ra; out[1] = rb;
        store_seal(out + 2, seal_bits(ra) ^ seal_bits(rb) ^ seal_bits(ticket));     // (wait_sealed)
        __threadfence_system();
        out[20] = ticket;
        __threadfence_system();
    }
}
struct BfgsSearchDev;
}  // namespace dzo

struct dzo_bfgs_s {
    int64_t n = 0;
    int32_t dtype = DZO_F64;
    hipStream_t stream = nullptr;
    dzo_objective_fn objective = nullptr;       // :734
    dzo_gradient_fn gradient = nullptr;         // :735
    dzo_constraint_fn constraint = nullptr;     // :736
    void *cb_ctx = nullptr;
    dzo_problem_s *problem = nullptr;
    int64_t iteration_count = 0;                // :737
    bool has_terminated = false;                // :738
    void *x = nullptr;                          // :739 (copy of x0, :769)
    double f = 0;                               // :740
    void *g = nullptr, *dx = nullptr, *dg = nullptr;   // :741-743
    double last_step_length = 0;                // :744
    int32_t last_step_type = DZO_STEP_NULL;     // :745
    void *H = nullptr;                          // :746
    void *d = nullptr;                          // :747 next_step_direction = H*g (not negated)
    void *d_alt = nullptr;                      // second buffer the fused update writes d_next to
    void *scratch = nullptr;                    // :748
    void *scratch2 = nullptr, *ref_point2 = nullptr;   // trial point / reference point of the second concurrent search
    void *spec_buf[4] = {nullptr, nullptr, nullptr, nullptr};   // trial points of the speculative evaluations (two per search)
    void *grad_pool = nullptr;                  // 24 vectors: gradients at the evaluated trial points of the last 4 rounds
    const void *best_grad[2] = {nullptr, nullptr};   // gradient at the best point of the last dual search, per direction (or null)
    void *ref_point = nullptr;                  // LineSearchEvaluator.reference_point (:17)
    int32_t max_increases = 0;                  // QuadraticLineSearch.max_increases (:181-188)
    double sign = -1.0;                         // trial point x + sign*t*dir: -1 BFGS (:945), +1 legacy evaluator (:33)
    bool no_hessian = false;                    // legacy GradientDescentOptimizer shares this state without H
    double df = 0;                              // GradientDescentOptimizer.delta_objective_value (:313)
    int64_t evals = 0;
    double *upd_part = nullptr;                 // device: 2*ceil(n/4) partials of the fused update scalars
    bool tri = false;                           // step! keeps the LOWER triangle of H only (launch_bfgs_update_tri)
    bool upper_stale = false;                   // ... and the upper one has not been mirrored since the last update
    double *tri_rowpart = nullptr, *tri_colpart = nullptr;
    void *tri_dummy = nullptr;
    double *ws = nullptr;                       // device: partials + scalars + flags
    double *host = nullptr;                     // pinned
    double *host_dev = nullptr;                 // the same buffer as the device sees it
    dzo::BfgsSearchDev *dsearch = nullptr;      // device: the two line searches' state machines and their posted requests (bfgs_dev_search)
    double ticket = 0;                          // last result published through host[20] (wait_ticket)
    double *partials() const { return ws; }
    double *scalars() const { return ws + dzo::kMaxPartialBlocks + 8; }       // [overlap, delta]
    double *result() const { return ws + dzo::kMaxPartialBlocks + 16; }       // [f]
    int32_t *flags() const { return reinterpret_cast<int32_t *>(ws + dzo::kMaxPartialBlocks + 24); }
    int32_t *phi_flags() const { return reinterpret_cast<int32_t *>(ws + dzo::kMaxPartialBlocks + 32); }   // 18 x int32 {changed, nonzero, differs from ref} per request, kept zero between uses
};

namespace dzo {

static int32_t bfgs_eval(dzo_bfgs_s *o, const void *point, double *f) {
    if (o->objective) {
        DZO_HIP(hipStreamSynchronize(o->stream));
        *f = round_to_dtype(o->dtype, o->objective(o->cb_ctx, point));
        return DZO_OK;
    }
    if (o->problem->kind != DZO_PROBLEM_LSE) {
        // the objective's finish kernel writes f straight into the pinned host buffer: the D->H copy
        // of 8 bytes is otherwise a blit kernel plus a gap on the critical path of every trial
        DZO_TRY(problem_eval_async(o->problem, o->stream, point, o->host_dev));
        DZO_HIP(hipStreamSynchronize(o->stream));
    } else {
        DZO_TRY(problem_eval_async(o->problem, o->stream, point, o->result()));
        DZO_HIP(hipMemcpyAsync(o->host, o->result(), sizeof(double), hipMemcpyDeviceToHost, o->stream));
        DZO_HIP(hipStreamSynchronize(o->stream));
    }
    *f = round_to_dtype(o->dtype, o->host[0]);
    return DZO_OK;
}

static int32_t bfgs_grad(dzo_bfgs_s *o) {
    if (o->gradient) {
        DZO_HIP(hipStreamSynchronize(o->stream));
        o->gradient(o->cb_ctx, o->g, o->x);
        return DZO_OK;
    }
    return problem_grad_async(o->problem, o->stream, o->g, o->x);
}

static int32_t bfgs_norm(dzo_bfgs_s *o, const void *v, double *out) {
    double ss = 0;
    DZO_TRY(dot_blocking(o->stream, o->n, o->dtype, v, v, o->partials(), o->host, &ss));
    *out = o->dtype == DZO_F32 ? (double)sqrtf((float)ss) : sqrt(ss);
    return DZO_OK;
}

// two norms with one host sync (:921 and :928 of the BFGS step): straight into the pinned host scalars, by one
// block in one launch when the vectors are short (dense-BFGS vectors are: n = 4096 at config 2) -- four launches
// and a stream synchronisation were 26 us of a 260-us step
static int32_t bfgs_norm_pair(dzo_bfgs_s *o, const void *a, const void *b, double *na, double *nb) {
    if (o->n <= 65536) {
        o->ticket += 1.0;
        DZO_TIMED("bfgs_norm_pair", o->stream);
        DZO_DISPATCH(o->dtype, hipLaunchKernelGGL(norm2_pair_kernel<T>, dim3(1), dim3(kBlock), 0, o->stream, o->n, (const T *)a, (const T *)b,
                                                  o->host_dev, o->ticket));
        DZO_HIP(hipGetLastError());
        DZO_TRY(wait_ticket(o->stream, o->host + 20, o->ticket));
        DZO_TRY(wait_sealed(o->stream, o->host, 2, o->host + 2, o->ticket));
        const double sa = o->host[0], sb = o->host[1];
        *na = o->dtype == DZO_F32 ? (double)sqrtf((float)sa) : sqrt(sa);
        *nb = o->dtype == DZO_F32 ? (double)sqrtf((float)sb) : sqrt(sb);
        return DZO_OK;
    }
    DZO_DISPATCH(o->dtype, launch_dot<T>(o->stream, o->n, (const T *)a, (const T *)a, o->partials(), o->host_dev));
    DZO_DISPATCH(o->dtype, launch_dot<T>(o->stream, o->n, (const T *)b, (const T *)b, o->upd_part, o->host_dev + 1));
    DZO_HIP(hipGetLastError());
    DZO_HIP(hipStreamSynchronize(o->stream));
    const double sa = o->host[0], sb = o->host[1];
    *na = o->dtype == DZO_F32 ? (double)sqrtf((float)sa) : sqrt(sa);
    *nb = o->dtype == DZO_F32 ? (double)sqrtf((float)sb) : sqrt(sb);
    return DZO_OK;
}

// scratch = x - t*dir, returns the two bracket flags
static int32_t bfgs_point(dzo_bfgs_s *o, const void *dir, double t, bool *changed, bool *nonzero) {
    hipStream_t s = o->stream;
    const bool want_flags = changed || nonzero;
    // the flag words are cleared (a fill kernel + its launch gap) only when somebody reads them
    if (want_flags) DZO_HIP(hipMemsetAsync(o->flags(), 0, 2 * sizeof(int32_t), s));
    {
        DZO_TIMED("bfgs_trial_point", s);
        const int grid = stream_grid(o->n, 1);
        DZO_DISPATCH(o->dtype, hipLaunchKernelGGL(phi_point_kernel<T>, dim3(grid), dim3(kBlock), 0, s, o->n, (T *)o->scratch,
                                                  (T)(o->sign * t), (const T *)dir, (const T *)o->x,
                                                  want_flags ? o->flags() : (int32_t *)nullptr));
    }
    DZO_HIP(hipGetLastError());
    if (changed || nonzero) {
        int32_t *hf = reinterpret_cast<int32_t *>(o->host + 4);
        DZO_HIP(hipMemcpyAsync(hf, o->flags(), 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        DZO_HIP(hipStreamSynchronize(s));
        if (changed) *changed = hf[0] != 0;
        if (nonzero) *nonzero = hf[1] != 0;
    }
    return DZO_OK;
}

// f(P(scratch)) for the point already in scratch (legacy :35-44)
static int32_t bfgs_phi_at_scratch(dzo_bfgs_s *o, double *f, bool *feasible) {
    *feasible = true;
    if (!o->constraint && o->problem && o->problem->cons_on)     // built-in UniformBoxConstraint (:36)
        DZO_TRY(box_clamp_async(o->stream, o->n, o->dtype, o->scratch, o->problem->cons_lo, o->problem->cons_hi));
    if (o->constraint) {
        DZO_HIP(hipStreamSynchronize(o->stream));
        if (!o->constraint(o->cb_ctx, o->scratch)) {           // :36-42
            *feasible = false;
            *f = o->dtype == DZO_F32 ? 3.4028234663852886e38 : 1.7976931348623157e308;   // typemax(T)
            return DZO_OK;
        }
    }
    o->evals += 1;
    return bfgs_eval(o, o->scratch, f);
}

// Built-in dense quadratic: trial point, objective and the bracket's flags in two launches with ONE
// host sync (value and flags land in pinned host memory); the plain sequence is trial point,
// [flags read-back + sync], objective, finish, value read-back + sync, [isequal + read-back + sync].
// `fused_ok` false: this objective / these options have no such path and nothing was done.
static int32_t bfgs_phi_fused(dzo_bfgs_s *o, const void *dir, double t, const void *ref, double *f, bool *changed,
                              bool *nonzero, bool *equal_ref, bool *fused_ok) {
    const bool fast = getenv("DZO_TUNE_BFGS_PHI_FUSED") ? atoi(getenv("DZO_TUNE_BFGS_PHI_FUSED")) != 0 : true;
    *fused_ok = fast && !o->objective && !o->constraint && o->problem &&
                problem_phi_async(o->problem, o->stream, o->x, dir, round_to_dtype(o->dtype, o->sign * t), o->scratch,
                                  o->phi_flags(), o->host_dev, ref);
    if (!*fused_ok) return DZO_OK;
    DZO_HIP(hipGetLastError());
    DZO_HIP(hipStreamSynchronize(o->stream));
    const int32_t *hf = reinterpret_cast<const int32_t *>(o->host + 4);
    if (changed) *changed = hf[0] != 0;
    if (nonzero) *nonzero = hf[1] != 0;
    if (equal_ref) *equal_ref = hf[2] == 0;
    *f = round_to_dtype(o->dtype, o->host[0]);
    return DZO_OK;
}

static int32_t bfgs_phi(dzo_bfgs_s *o, const void *dir, double t, double *f) {
    bool fused_ok = false;
    DZO_TRY(bfgs_phi_fused(o, dir, t, nullptr, f, nullptr, nullptr, nullptr, &fused_ok));
    if (fused_ok) { o->evals += 1; return DZO_OK; }
    bool feasible;
    DZO_TRY(bfgs_point(o, dir, t, nullptr, nullptr));
    return bfgs_phi_at_scratch(o, f, &feasible);
}

static int32_t bfgs_scratch_equals(dzo_bfgs_s *o, const void *other, bool *equal) {
    hipStream_t s = o->stream;
    DZO_HIP(hipMemsetAsync(o->flags(), 0, sizeof(int32_t), s));
    DZO_DISPATCH(o->dtype, launch_isequal<T>(s, o->n, (const T *)o->scratch, (const T *)other, o->flags()));
    int32_t *hf = reinterpret_cast<int32_t *>(o->host + 4);
    DZO_HIP(hipMemcpyAsync(hf, o->flags(), sizeof(int32_t), hipMemcpyDeviceToHost, s));
    DZO_HIP(hipStreamSynchronize(s));
    *equal = hf[0] == 0;
    return DZO_OK;
}

static inline bool finite_t(double v) { return std::isfinite(v); }

// find_three_point_bracket (legacy/DZOptimization.jl:49-172), started at step size t0.
// Mirrors oracle/dzo_oracle_impl.h bfgs_bracket line for line.
static int32_t bfgs_bracket(dzo_bfgs_s *o, const void *dir, double f0, double t0, double *x1, double *f1, double *x2,
                            double *f2) {
    const int32_t dt = o->dtype;
    const size_t bytes = (size_t)o->n * dtype_size(dt);
    *x1 = 0; *f1 = f0; *x2 = 0; *f2 = f0;
    if (!finite_t(f0)) return DZO_OK;                            // :64-66
    if (!(t0 > 0) || !finite_t(t0)) return DZO_OK;
    bool changed = false, nonzero = false;
    double fa = 0;
    bool feasible = true, have_fa = false;
    DZO_TRY(bfgs_phi_fused(o, dir, t0, nullptr, &fa, &changed, &nonzero, nullptr, &have_fa));   // :71-80 (+ :104 speculatively)
    if (!have_fa) DZO_TRY(bfgs_point(o, dir, t0, &changed, &nonzero));
    if (!nonzero) return DZO_OK;                                 // :83-85
    double step = t0;
    bool small = false;
    while (!changed) {                                           // :91-101
        step = round_to_dtype(dt, step + step);
        small = true;
        have_fa = false;                                         // (the speculative value belonged to a point that did not move)
        if (!finite_t(step)) return DZO_OK;
        DZO_TRY(bfgs_point(o, dir, step, &changed, nullptr));
    }
    if (have_fa) o->evals += 1;
    else DZO_TRY(bfgs_phi_at_scratch(o, &fa, &feasible));        // :104,:126
    if (small) {                                                 // :107-123
        if (!feasible) return DZO_OK;
        bool eq;
        DZO_TRY(bfgs_scratch_equals(o, o->x, &eq));
        if (eq) return DZO_OK;
    }
    if (fa <= f0) {                                              // :130
        int32_t increases = 0;
        DZO_HIP(hipMemcpyAsync(o->ref_point, o->scratch, bytes, hipMemcpyDeviceToDevice, o->stream));   // :136
        for (;;) {                                               // :143-156
            const double dbl = round_to_dtype(dt, step + step);
            increases += 1;
            double fb;
            bool eq = false, fused_ok = false;
            DZO_TRY(bfgs_phi_fused(o, dir, dbl, o->ref_point, &fb, nullptr, nullptr, &eq, &fused_ok));   // value and the :150 test together
            if (fused_ok) o->evals += 1;
            else DZO_TRY(bfgs_phi(o, dir, dbl, &fb));
            bool stop = (o->max_increases > 0 && increases >= o->max_increases) || !finite_t(fb) || fb > fa;
            if (!stop) {
                if (!fused_ok) DZO_TRY(bfgs_scratch_equals(o, o->ref_point, &eq));  // :150
                stop = eq;
            }
            if (stop) { *x1 = step; *f1 = fa; *x2 = dbl; *f2 = fb; return DZO_OK; }   // :151
            step = dbl;
            fa = fb;
            DZO_HIP(hipMemcpyAsync(o->ref_point, o->scratch, bytes, hipMemcpyDeviceToDevice, o->stream));   // :155
        }
    } else {                                                     // :157-171
        for (;;) {
            const double hs = round_to_dtype(dt, 0.5 * step);
            double fb;
            DZO_TRY(bfgs_phi(o, dir, hs, &fb));
            if (fb <= f0) { *x1 = hs; *f1 = fb; *x2 = step; *f2 = fa; return DZO_OK; }   // :166
            if (hs == 0.0) return DZO_OK;
            step = hs;
            fa = fb;
        }
    }
}

// QuadraticLineSearch (legacy/DZOptimization.jl:191-216)
static int32_t bfgs_quadratic_search(dzo_bfgs_s *o, const void *dir, double f0, double t0, double *t_best,
                                     double *f_best) {
    const int32_t dt = o->dtype;
    double x1, f1, x2, f2;
    DZO_TRY(bfgs_bracket(o, dir, f0, t0, &x1, &f1, &x2, &f2));   // :195
    double xb = 0, fb = f0;                                      // :196
    if (f1 < fb) { xb = x1; fb = f1; }                           // :197-199
    if (f2 < fb) { xb = x2; fb = f2; }                           // :200-202
    const double d1 = round_to_dtype(dt, f0 - f1), d2 = round_to_dtype(dt, f2 - f1);
    const double sum = round_to_dtype(dt, d1 + d2);              // :203-205
    if (d1 >= 0 && d2 >= 0 && sum > 0) {                         // :206
        const double num = round_to_dtype(dt, round_to_dtype(dt, d1 + d1) + sum);
        const double ratio = round_to_dtype(dt, num / round_to_dtype(dt, sum + sum));   // :207-208
        const double xq = round_to_dtype(dt, ratio * x1);        // :209
        double fq;
        DZO_TRY(bfgs_phi(o, dir, xq, &fq));                      // :210
        if (fq < fb) { xb = xq; fb = fq; }                       // :211-213
    }
    *t_best = xb; *f_best = fb;
    return DZO_OK;
}

// ---------------------------------------------------------------------------------------------
// The two line searches of a dense BFGS step (gradient direction :922-925, quasi-Newton direction
// :929-932) are independent, so they are advanced side by side: each round evaluates the next
// point of BOTH in one pass over A (quadratic_phi2_kernel) with one host sync.  PhiSearch is
// bfgs_bracket + bfgs_quadratic_search turned into a resumable state machine: same decisions,
// same values, same evaluation counts -- only the order in which the two searches' evaluations
// reach the device changes.
// ---------------------------------------------------------------------------------------------
struct PhiSearch {
    const void *dir = nullptr;
    void *scratch = nullptr, *ref_point = nullptr;
    double f0 = 0, t0 = 0;
    enum { FIRST, DOUBLING, SHRINKING, QUADRATIC, DONE, SEQUENTIAL } state = DONE;
    double step = 0, fa = 0, req_t = 0, x1 = 0, f1 = 0, x2 = 0, f2 = 0, xb = 0, fb = 0;
    int32_t increases = 0;
    bool want = false, req_ref = false;
    double t_best = 0, f_best = 0;
    void *cur_point = nullptr;     // buffer holding the trial point of the last consumed evaluation
    void *spec[2] = {nullptr, nullptr};
    // gradient (A*x_t, a by-product of the evaluation) and round of: the last consumed evaluation, the
    // point `step`, the bracket ends, the best point so far
    const void *cur_grad = nullptr, *step_grad = nullptr, *g1 = nullptr, *g2 = nullptr, *best_grad = nullptr;
    int cur_round = 0, step_round = 0, r1 = 0, r2 = 0, best_round = 0;
};

static void phi_search_to_quadratic(dzo_bfgs_s *o, PhiSearch &q) {      // bfgs_quadratic_search after :195
    const int32_t dt = o->dtype;
    q.xb = 0; q.fb = q.f0; q.best_grad = nullptr;                // :196
    if (q.f1 < q.fb) { q.xb = q.x1; q.fb = q.f1; q.best_grad = q.g1; q.best_round = q.r1; }   // :197-199
    if (q.f2 < q.fb) { q.xb = q.x2; q.fb = q.f2; q.best_grad = q.g2; q.best_round = q.r2; }   // :200-202
    const double d1 = round_to_dtype(dt, q.f0 - q.f1), d2 = round_to_dtype(dt, q.f2 - q.f1);
    const double sum = round_to_dtype(dt, d1 + d2);              // :203-205
    if (d1 >= 0 && d2 >= 0 && sum > 0) {                         // :206
        const double num = round_to_dtype(dt, round_to_dtype(dt, d1 + d1) + sum);
        const double ratio = round_to_dtype(dt, num / round_to_dtype(dt, sum + sum));   // :207-208
        q.req_t = round_to_dtype(dt, ratio * q.x1);              // :209
        q.want = true; q.req_ref = false;
        q.state = PhiSearch::QUADRATIC;
    } else {
        q.t_best = q.xb; q.f_best = q.fb;
        q.want = false;
        q.state = PhiSearch::DONE;
    }
}

static void phi_search_bracket_done(dzo_bfgs_s *o, PhiSearch &q, double x1, double f1, double x2, double f2,
                                    const void *g1 = nullptr, int r1 = 0, const void *g2 = nullptr, int r2 = 0) {
    q.x1 = x1; q.f1 = f1; q.x2 = x2; q.f2 = f2; q.g1 = g1; q.r1 = r1; q.g2 = g2; q.r2 = r2;
    phi_search_to_quadratic(o, q);
}

static void phi_search_begin(dzo_bfgs_s *o, PhiSearch &q, const void *dir, double f0, double t0, void *scratch, void *ref_point) {
    q = PhiSearch();
    q.dir = dir; q.f0 = f0; q.t0 = t0; q.scratch = scratch; q.ref_point = ref_point; q.cur_point = scratch;
    if (!finite_t(f0) || !(t0 > 0) || !finite_t(t0)) {           // :64-66 -> bracket (0, f0, 0, f0)
        phi_search_bracket_done(o, q, 0, f0, 0, f0);
        return;
    }
    q.state = PhiSearch::FIRST;
    q.step = t0;
    q.req_t = t0; q.want = true; q.req_ref = false;
}

// feed the result of the pending request; may enqueue a device copy (the :136 / :155 reference point)
static int32_t phi_search_feed(dzo_bfgs_s *o, PhiSearch &q, double f, bool changed, bool nonzero, bool equal_ref) {
    const int32_t dt = o->dtype;
    const size_t bytes = (size_t)o->n * dtype_size(dt);
    q.want = false;
    switch (q.state) {
    case PhiSearch::FIRST:
        if (!nonzero) { phi_search_bracket_done(o, q, 0, q.f0, 0, q.f0); return DZO_OK; }   // :83-85 (the speculative value is dropped)
        if (!changed) { q.state = PhiSearch::SEQUENTIAL; return DZO_OK; }                  // :91-101 tiny-step path: rare, done sequentially
        o->evals += 1;
        q.fa = f;                                                // :104
        q.step_grad = q.cur_grad; q.step_round = q.cur_round;
        if (q.fa <= q.f0) {                                      // :130
            q.increases = 0;
            DZO_HIP(hipMemcpyAsync(q.ref_point, q.cur_point, bytes, hipMemcpyDeviceToDevice, o->stream));   // :136
            q.state = PhiSearch::DOUBLING;
            q.req_t = round_to_dtype(dt, q.step + q.step); q.increases += 1;
            q.want = true; q.req_ref = true;
        } else {
            q.state = PhiSearch::SHRINKING;
            q.req_t = round_to_dtype(dt, 0.5 * q.step);
            q.want = true; q.req_ref = false;
        }
        return DZO_OK;
    case PhiSearch::DOUBLING: {                                  // :143-156
        o->evals += 1;
        const double dbl = q.req_t, fb = f;
        bool stop = (o->max_increases > 0 && q.increases >= o->max_increases) || !finite_t(fb) || fb > q.fa;
        if (!stop) stop = equal_ref;                             // :150
        if (stop) { phi_search_bracket_done(o, q, q.step, q.fa, dbl, fb, q.step_grad, q.step_round, q.cur_grad, q.cur_round); return DZO_OK; }   // :151
        q.step = dbl; q.fa = fb; q.step_grad = q.cur_grad; q.step_round = q.cur_round;
        DZO_HIP(hipMemcpyAsync(q.ref_point, q.cur_point, bytes, hipMemcpyDeviceToDevice, o->stream));       // :155
        q.req_t = round_to_dtype(dt, q.step + q.step); q.increases += 1;
        q.want = true; q.req_ref = true;
        return DZO_OK;
    }
    case PhiSearch::SHRINKING: {                                 // :157-171
        o->evals += 1;
        const double hs = q.req_t, fb = f;
        if (fb <= q.f0) { phi_search_bracket_done(o, q, hs, fb, q.step, q.fa, q.cur_grad, q.cur_round, q.step_grad, q.step_round); return DZO_OK; }   // :166
        if (hs == 0.0) { phi_search_bracket_done(o, q, 0, q.f0, 0, q.f0); return DZO_OK; }
        q.step = hs; q.fa = fb; q.step_grad = q.cur_grad; q.step_round = q.cur_round;
        q.req_t = round_to_dtype(dt, 0.5 * q.step);
        q.want = true; q.req_ref = false;
        return DZO_OK;
    }
    case PhiSearch::QUADRATIC:                                   // :210-213
        o->evals += 1;
        if (f < q.fb) { q.xb = q.req_t; q.fb = f; q.best_grad = q.cur_grad; q.best_round = q.cur_round; }
        q.t_best = q.xb; q.f_best = q.fb;
        q.state = PhiSearch::DONE;
        return DZO_OK;
    default:
        return DZO_OK;
    }
}

// both searches of a step; false in *done when the objective has no two-request kernel (nothing was evaluated)
static int32_t bfgs_dual_search(dzo_bfgs_s *o, const void *dir_a, double t0_a, const void *dir_b, double t0_b,
                                double *t_a, double *f_a, double *t_b, double *f_b, bool *done) {
    const bool enabled = getenv("DZO_TUNE_BFGS_DUAL_SEARCH") ? atoi(getenv("DZO_TUNE_BFGS_DUAL_SEARCH")) != 0 : true;
    *done = false;
    if (!enabled || o->objective || o->constraint || !o->problem || o->problem->kind != DZO_PROBLEM_QUADRATIC ||
        o->problem->l2 != 0.0 || o->problem->cons_on)
        return DZO_OK;
    PhiSearch q[2];
    phi_search_begin(o, q[0], dir_a, o->f, t0_a, o->scratch, o->ref_point);
    phi_search_begin(o, q[1], dir_b, o->f, t0_b, o->scratch2, o->ref_point2);
    for (int r = 0; r < 2; ++r) { q[r].spec[0] = o->spec_buf[2 * r]; q[r].spec[1] = o->spec_buf[2 * r + 1]; }
    const int32_t dt = o->dtype;
    const size_t vbytes = (size_t)((o->n + 63) / 64 * 64) * dtype_size(dt);
    int round = 0;
    o->best_grad[0] = o->best_grad[1] = nullptr;
    while (q[0].want || q[1].want) {
        round += 1;
        // Per search: the evaluation it needs now (slot 0) plus the one or two it will most likely
        // need next (slots 1, 2) -- the first doubling and the first halving after the first point,
        // the next doubling / halving inside those loops.  All of them ride on the same pass over A;
        // a speculative value is used only if the state machine then asks for exactly that point.
        PhiDirHost req[2];
        double spec_t[2][3];
        bool spec_ref[2][3];
        for (int r = 0; r < 2; ++r) {
            PhiSearch &sm = q[r];
            req[r].dir = sm.dir;
            for (int e = 0; e < 3; ++e) { spec_t[r][e] = 0; spec_ref[r][e] = false; }
            if (!sm.want) { req[r].dir = q[1 - r].dir; continue; }
            auto post = [&](int slot, double t, void *out, const void *ref, int ref_req, bool is_ref) {
                req[r].ts[slot] = round_to_dtype(dt, o->sign * t);
                req[r].point_out[slot] = out; req[r].ref[slot] = ref; req[r].ref_req[slot] = ref_req; req[r].active[slot] = true;
                req[r].grad_out[slot] = (char *)o->grad_pool + (size_t)((round % 4) * 6 + r * 3 + slot) * vbytes;
                spec_t[r][slot] = t; spec_ref[r][slot] = is_ref;
            };
            post(0, sm.req_t, sm.scratch, sm.req_ref ? sm.ref_point : nullptr, -1, sm.req_ref);
            if (sm.state == PhiSearch::FIRST) {
                post(1, round_to_dtype(dt, sm.req_t + sm.req_t), sm.spec[0], nullptr, 0, true);      // first doubling (:143), :150 against slot 0
                post(2, round_to_dtype(dt, 0.5 * sm.req_t), sm.spec[1], nullptr, -1, false);         // first halving (:158)
            } else if (sm.state == PhiSearch::DOUBLING) {
                post(1, round_to_dtype(dt, sm.req_t + sm.req_t), sm.spec[0], nullptr, 0, true);
            } else if (sm.state == PhiSearch::SHRINKING) {
                post(1, round_to_dtype(dt, 0.5 * sm.req_t), sm.spec[0], nullptr, -1, false);
            }
        }
        o->ticket += 1.0;
        if (!problem_phi6_async(o->problem, o->stream, o->x, req, o->phi_flags(), o->host_dev, o->ticket)) return DZO_OK;
        DZO_HIP(hipGetLastError());
        DZO_TRY(wait_ticket(o->stream, o->host + 20, o->ticket));
        DZO_TRY(wait_sealed(o->stream, o->host, 17, o->host + 17, o->ticket));
        const int32_t *hf = reinterpret_cast<const int32_t *>(o->host + 8);
        for (int r = 0; r < 2; ++r) {
            PhiSearch &sm = q[r];
            if (!req[r].active[0]) continue;
            bool used[3] = {false, false, false};
            int slot = 0;                                     // the primary request first, then matching speculative ones
            for (;;) {
                used[slot] = true;
                sm.cur_point = req[r].point_out[slot];
                sm.cur_grad = req[r].grad_out[slot]; sm.cur_round = round;
                const int fq = (r * 3 + slot) * 3;
                DZO_TRY(phi_search_feed(o, sm, round_to_dtype(dt, o->host[r * 3 + slot]), hf[fq] != 0, hf[fq + 1] != 0, hf[fq + 2] == 0));
                if (!sm.want) break;
                int next = -1;
                for (int e = 1; e < 3; ++e)
                    if (req[r].active[e] && !used[e] && spec_t[r][e] == sm.req_t && spec_ref[r][e] == sm.req_ref) next = e;
                if (next < 0) break;
                slot = next;
            }
        }
    }
    // the rare tiny-step path (:91-101) is finished with the sequential code
    double tt[2], ff[2];
    for (int r = 0; r < 2; ++r) {
        if (q[r].state == PhiSearch::SEQUENTIAL) DZO_TRY(bfgs_quadratic_search(o, q[r].dir, o->f, q[r].t0, &tt[r], &ff[r]));
        else { tt[r] = q[r].t_best; ff[r] = q[r].f_best; }
    }
    for (int r = 0; r < 2; ++r)        // gradient at the best point, if its pool entry has not been recycled since
        if (q[r].state == PhiSearch::DONE && q[r].t_best != 0.0 && q[r].best_grad && round - q[r].best_round < 4)
            o->best_grad[r] = q[r].best_grad;
    *t_a = tt[0]; *f_a = ff[0]; *t_b = tt[1]; *f_b = ff[1];
    *done = true;
    return DZO_OK;
}

// ---------------------------------------------------------------------------------------------
// The same two searches with the state machines ON THE DEVICE.  A host-driven round costs a round trip
// (kernel -> pinned word -> host decision -> next launch reaches the GPU: 20-25 us against 36 us of
// kernels per round at config 2), and a step has three of them: norms -> first round -> second round ->
// accepted step.  Here the norm kernel's last thread begins both searches and posts the first round's
// requests in device memory (PhiReqDev), every round's finish kernel feeds the six values to the two
// machines and posts the next round's requests, and the host enqueues norm + rounds back to back and
// waits ONCE, for the last enqueued round's summary; a round behind two finished searches does nothing.
// PhiDev is PhiSearch with buffers as indices: a trial point is its request slot (0: scratch, 1, 2: the
// two speculative buffers), a gradient its grad_pool entry.  Same arithmetic (doubles rounded to the
// dtype where the host code rounds), same decisions, same evaluation counts: tested bit for bit against
// the host-driven search.
// ---------------------------------------------------------------------------------------------
struct PhiDev {
    double f0, t0, step, fa, req_t, x1, f1, x2, f2, xb, fb, t_best, f_best;
    double spec_t[3];
    int32_t state, increases, want, req_ref;
    int32_t spec_ref[3], active[3];
    int32_t step_grad, g1, g2, best_grad;          // grad_pool entries (-1: none)
    int32_t step_round, r1, r2, best_round;
    int32_t ref_buf;                               // the trial-point buffer (= request slot) that holds the reference point (:136 / :155)
};
struct BfgsSearchDev {
    PhiDev q[2];
    PhiReqDev req;
    double norm[2];
    int64_t evals;
};
enum { kPhiFirst = 0, kPhiDoubling, kPhiShrinking, kPhiQuadratic, kPhiDone, kPhiSequential };
constexpr int kSumBase = 24;                       // summary of the searches in the pinned host doubles, from here
constexpr int kHostDoubles = 64;

__device__ __forceinline__ double dev_rt(int32_t dt, double v) { return dt == DZO_F32 ? (double)(float)v : v; }   // round_to_dtype

__device__ __forceinline__ void phi_dev_to_quadratic(int32_t dt, PhiDev &q) {                     // phi_search_to_quadratic
    q.xb = 0; q.fb = q.f0; q.best_grad = -1; q.best_round = 0;
    if (q.f1 < q.fb) { q.xb = q.x1; q.fb = q.f1; q.best_grad = q.g1; q.best_round = q.r1; }
    if (q.f2 < q.fb) { q.xb = q.x2; q.fb = q.f2; q.best_grad = q.g2; q.best_round = q.r2; }
    const double d1 = dev_rt(dt, q.f0 - q.f1), d2 = dev_rt(dt, q.f2 - q.f1);
    const double sum = dev_rt(dt, d1 + d2);
    if (d1 >= 0 && d2 >= 0 && sum > 0) {
        const double num = dev_rt(dt, dev_rt(dt, d1 + d1) + sum);
        const double ratio = dev_rt(dt, num / dev_rt(dt, sum + sum));
        q.req_t = dev_rt(dt, ratio * q.x1);
        q.want = 1; q.req_ref = 0;
        q.state = kPhiQuadratic;
    } else {
        q.t_best = q.xb; q.f_best = q.fb;
        q.want = 0;
        q.state = kPhiDone;
    }
}

__device__ __forceinline__ void phi_dev_bracket_done(int32_t dt, PhiDev &q, double x1, double f1, double x2, double f2, int g1 = -1, int r1 = 0,
                                            int g2 = -1, int r2 = 0) {
    q.x1 = x1; q.f1 = f1; q.x2 = x2; q.f2 = f2; q.g1 = g1; q.r1 = r1; q.g2 = g2; q.r2 = r2;
    phi_dev_to_quadratic(dt, q);
}

__device__ __forceinline__ void phi_dev_begin(int32_t dt, PhiDev &q, double f0, double t0) {      // phi_search_begin
    q.f0 = f0; q.t0 = t0;
    q.step = q.fa = q.req_t = q.x1 = q.f1 = q.x2 = q.f2 = q.xb = q.fb = q.t_best = q.f_best = 0;
    q.increases = 0; q.want = 0; q.req_ref = 0;
    q.step_grad = q.g1 = q.g2 = q.best_grad = -1;
    q.step_round = q.r1 = q.r2 = q.best_round = 0;
    q.ref_buf = 0;
    q.state = kPhiDone;
    for (int e = 0; e < 3; ++e) { q.spec_t[e] = 0; q.spec_ref[e] = 0; q.active[e] = 0; }
    if (!isfinite(f0) || !(t0 > 0) || !isfinite(t0)) { phi_dev_bracket_done(dt, q, 0, f0, 0, f0); return; }
    q.state = kPhiFirst;
    q.step = t0;
    q.req_t = t0; q.want = 1; q.req_ref = 0;
}

// phi_search_feed; slot: where the consumed evaluation's trial point lies, cur_grad / cur_round: its gradient
__device__ __forceinline__ void phi_dev_feed(int32_t dt, int32_t max_increases, PhiDev &q, double f, bool changed, bool nonzero, bool equal_ref,
                                    int slot, int cur_grad, int cur_round, int64_t &evals) {
    q.want = 0;
    switch (q.state) {
    case kPhiFirst:
        if (!nonzero) { phi_dev_bracket_done(dt, q, 0, q.f0, 0, q.f0); return; }
        if (!changed) { q.state = kPhiSequential; return; }
        evals += 1;
        q.fa = f;
        q.step_grad = cur_grad; q.step_round = cur_round;
        if (q.fa <= q.f0) {
            q.increases = 0;
            q.ref_buf = slot;
            q.state = kPhiDoubling;
            q.req_t = dev_rt(dt, q.step + q.step); q.increases += 1;
            q.want = 1; q.req_ref = 1;
        } else {
            q.state = kPhiShrinking;
            q.req_t = dev_rt(dt, 0.5 * q.step);
            q.want = 1; q.req_ref = 0;
        }
        return;
    case kPhiDoubling: {
        evals += 1;
        const double dbl = q.req_t, fb = f;
        bool stop = (max_increases > 0 && q.increases >= max_increases) || !isfinite(fb) || fb > q.fa;
        if (!stop) stop = equal_ref;
        if (stop) { phi_dev_bracket_done(dt, q, q.step, q.fa, dbl, fb, q.step_grad, q.step_round, cur_grad, cur_round); return; }
        q.step = dbl; q.fa = fb; q.step_grad = cur_grad; q.step_round = cur_round;
        q.ref_buf = slot;
        q.req_t = dev_rt(dt, q.step + q.step); q.increases += 1;
        q.want = 1; q.req_ref = 1;
        return;
    }
    case kPhiShrinking: {
        evals += 1;
        const double hs = q.req_t, fb = f;
        if (fb <= q.f0) { phi_dev_bracket_done(dt, q, hs, fb, q.step, q.fa, cur_grad, cur_round, q.step_grad, q.step_round); return; }
        if (hs == 0.0) { phi_dev_bracket_done(dt, q, 0, q.f0, 0, q.f0); return; }
        q.step = hs; q.fa = fb; q.step_grad = cur_grad; q.step_round = cur_round;
        q.req_t = dev_rt(dt, 0.5 * q.step);
        q.want = 1; q.req_ref = 0;
        return;
    }
    case kPhiQuadratic:
        evals += 1;
        if (f < q.fb) { q.xb = q.req_t; q.fb = f; q.best_grad = cur_grad; q.best_round = cur_round; }
        q.t_best = q.xb; q.f_best = q.fb;
        q.state = kPhiDone;
        return;
    default:
        return;
    }
}

// the requests of the next launch (the loop body of bfgs_dual_search that fills req[])
__device__ __forceinline__ void phi_dev_post(int32_t dt, double sign, PhiDev &q, PhiReqDev &R, int side) {
    for (int e = 0; e < 3; ++e) {
        q.spec_t[e] = 0; q.spec_ref[e] = 0; q.active[e] = 0;
        R.ts[side][e] = 0; R.active[side][e] = 0; R.ref_req[side][e] = -1; R.use_ref[side][e] = 0;
    }
    if (!q.want) return;
    auto post = [&](int slot, double t, int use_ref, int ref_req, int is_ref) {
        R.ts[side][slot] = dev_rt(dt, sign * t);
        R.active[side][slot] = 1; R.use_ref[side][slot] = use_ref; R.ref_req[side][slot] = ref_req;
        q.spec_t[slot] = t; q.spec_ref[slot] = is_ref; q.active[slot] = 1;
    };
    post(0, q.req_t, q.req_ref ? 1 + q.ref_buf : 0, -1, q.req_ref);
    if (q.state == kPhiFirst) {
        post(1, dev_rt(dt, q.req_t + q.req_t), 0, 0, 1);
        post(2, dev_rt(dt, 0.5 * q.req_t), 0, -1, 0);
    } else if (q.state == kPhiDoubling) {
        post(1, dev_rt(dt, q.req_t + q.req_t), 0, 0, 1);
    } else if (q.state == kPhiShrinking) {
        post(1, dev_rt(dt, 0.5 * q.req_t), 0, -1, 0);
    }
}

// what the host needs after its one wait: 5 doubles per search + the evaluation count
//   [0] want | state << 8 | (best_grad + 1) << 16 | best_round << 32 (as an integer-valued double: < 2^53)
//   [1] t_best  [2] f_best  [3] t0  [4] the direction's norm
constexpr int kSumStride = 5;
__device__ __forceinline__ unsigned long long phi_dev_summary(const PhiDev &q, double norm, double *o) {      // returns its share of the seal
    const int64_t packed = (int64_t)q.want | ((int64_t)q.state << 8) | ((int64_t)(q.best_grad + 1) << 16) | ((int64_t)q.best_round << 32);
    const double v[kSumStride] = {(double)packed, q.t_best, q.f_best, q.t0, norm};
    unsigned long long seal = 0;
    // (system-scope stores: written through to the host's memory at once, no fence needed -- the host accepts the summary
    // only when its seal matches, wait_sealed)
    unsigned long long *ob = reinterpret_cast<unsigned long long *>(o);
#pragma unroll
    for (int i = 0; i < kSumStride; ++i) { __hip_atomic_store(ob + i, seal_bits(v[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); seal ^= seal_bits(v[i]); }
    return seal;
}

// BfgsSearchDev lives in device memory between kernels; inside a kernel the whole block moves it to LDS (one memory
// latency instead of one per field: a single thread walking the structure in device memory took 13-20 us per round)
// and thread 0 works on register copies.
constexpr int kSearchWords = (int)(sizeof(BfgsSearchDev) / 4);
static_assert(sizeof(BfgsSearchDev) % 4 == 0, "word copies");
__device__ __forceinline__ void search_to_lds(const BfgsSearchDev *S, BfgsSearchDev *L) {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(S);
    uint32_t *dst = reinterpret_cast<uint32_t *>(L);
    for (int i = threadIdx.x; i < kSearchWords; i += kBlock) dst[i] = src[i];
}
__device__ __forceinline__ void search_from_lds(BfgsSearchDev *S, const BfgsSearchDev *L) {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(L);
    uint32_t *dst = reinterpret_cast<uint32_t *>(S);
    for (int i = threadIdx.x; i < kSearchWords; i += kBlock) dst[i] = src[i];
}

// :921 and :928 (the two norms) and the begin of both searches
template <typename T>
__global__ __launch_bounds__(kBlock) void norm2_pair_begin_kernel(int64_t n, const T *__restrict__ a, const T *__restrict__ b,
                                                                  BfgsSearchDev *__restrict__ S, double f0, double step_length,
                                                                  int32_t dt, double sign) {
    __shared__ double lds[2 * kWaves];
    __shared__ BfgsSearchDev L;
    double sa = 0, sb = 0;
    int64_t i = threadIdx.x;
    for (; i + 7 * kBlock < n; i += 8 * kBlock) {                // (norm2_pair_kernel's sums: sixteen loads in flight, then the
        T ta[8], tb[8];                                          // fmas in the plain loop's order -- one block, latency-bound)
#pragma unroll
        for (int u = 0; u < 8; ++u) { ta[u] = a[i + (int64_t)u * kBlock]; tb[u] = b[i + (int64_t)u * kBlock]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double va = (double)ta[u], vb = (double)tb[u];
            sa = __builtin_fma(va, va, sa);
            sb = __builtin_fma(vb, vb, sb);
        }
    }
    for (; i < n; i += kBlock) {
        const double va = (double)a[i], vb = (double)b[i];
        sa = __builtin_fma(va, va, sa);
        sb = __builtin_fma(vb, vb, sb);
    }
    const double sums[2] = {sa, sb};
    double red[2];
    block_sum_multi<2>(sums, lds, red);                          // (bit for bit what two block_sum calls give)
    const double ra = red[0], rb = red[1];
    if (threadIdx.x == 0) {
        const double na = dt == DZO_F32 ? (double)sqrtf((float)ra) : sqrt(ra);
        const double nb = dt == DZO_F32 ? (double)sqrtf((float)rb) : sqrt(rb);
        phi_dev_begin(dt, L.q[0], f0, dev_rt(dt, step_length / na));      // (in LDS: private copies of the structures end up in scratch memory)
        phi_dev_begin(dt, L.q[1], f0, dev_rt(dt, step_length / nb));
        phi_dev_post(dt, sign, L.q[0], L.req, 0);
        phi_dev_post(dt, sign, L.q[1], L.req, 1);
        L.norm[0] = na; L.norm[1] = nb;
        L.evals = 0;
    }
    __syncthreads();
    search_from_lds(S, &L);
}

template <typename V> __device__ __forceinline__ V sel3(int i, V a, V b, V c) { return i == 0 ? a : (i == 1 ? b : c); }

// finish_phi6_kernel + the host loop body of bfgs_dual_search behind it: the round's six sums, both machines fed --
// the primary request, then the speculative ones that match what the machine asks for next --, the next round's
// requests posted, the summary and the ticket published.  The reference point of :136 / :155 is not copied anywhere:
// it is the trial point the machine consumed last, which stays in its request's buffer until the next launch -- and
// that launch reads a reference element before it writes any trial-point element of the same index (PhiDev::ref_buf).
__global__ __launch_bounds__(kBlock) void finish_phi6_advance_kernel(const double *__restrict__ partials, int64_t count, double scale,
                                                                     double *__restrict__ out, int32_t *__restrict__ flags, double ticket,
                                                                     BfgsSearchDev *__restrict__ S, int round, int32_t dt,
                                                                     int32_t max_increases, double sign, int publish) {
    __shared__ double lds6[6 * kWaves];
    __shared__ BfgsSearchDev L;
    __shared__ int32_t hf_s[18];
    const bool live = S->q[0].want || S->q[1].want;             // (uniform; nothing was evaluated otherwise)
    search_to_lds(S, &L);                                       // (complete at the barriers inside block_sum_multi, like hf_s)
    if (threadIdx.x < 18) { hf_s[threadIdx.x] = flags[threadIdx.x]; flags[threadIdx.x] = 0; }    // (re-armed for the next round)
    double v[6] = {0, 0, 0, 0, 0, 0};
    if (live) {
        int64_t i = threadIdx.x;
        for (; i + 7 * kBlock < count; i += 8 * kBlock) {       // (finish_phi6_kernel's sums, same order)
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
    }
    double sres[6];
    block_sum_multi<6>(v, lds6, sres);
    // The two searches are independent machines (their own PhiDev, their own side of the request table): wave 0's first
    // lane drives the one along the gradient, wave 1's the one along the BFGS direction -- side by side instead of one
    // after the other on a single lane walking structures in LDS, which was most of this kernel's 7.7 us.
    __shared__ double s_sum[6];
    __shared__ int64_t s_evals[2];
    if (threadIdx.x == 0) {
#pragma unroll
        for (int r = 0; r < 6; ++r) s_sum[r] = sres[r];
    }
    __syncthreads();
    if (live && (threadIdx.x == 0 || threadIdx.x == 64)) {
        const int r = threadIdx.x >> 6;
        PhiDev &sm = L.q[r];                                     // (worked on in LDS: private copies end up in scratch memory, 4x slower)
        int64_t evals = 0;
        if (sm.active[0]) {
            const double f3[3] = {dev_rt(dt, scale * s_sum[r * 3 + 0]), dev_rt(dt, scale * s_sum[r * 3 + 1]), dev_rt(dt, scale * s_sum[r * 3 + 2])};
            int used = 0, slot = 0;
            for (int turn = 0; turn < 3; ++turn) {           // (at most the three requests of this direction)
                used |= 1 << slot;
                const int changed = sel3(slot, hf_s[r * 9 + 0], hf_s[r * 9 + 3], hf_s[r * 9 + 6]);
                const int nonzero = sel3(slot, hf_s[r * 9 + 1], hf_s[r * 9 + 4], hf_s[r * 9 + 7]);
                const int differs = sel3(slot, hf_s[r * 9 + 2], hf_s[r * 9 + 5], hf_s[r * 9 + 8]);
                phi_dev_feed(dt, max_increases, sm, sel3(slot, f3[0], f3[1], f3[2]), changed != 0, nonzero != 0, differs == 0,
                             slot, (round % 4) * 6 + r * 3 + slot, round, evals);
                if (!sm.want) break;
                int next = -1;
#pragma unroll
                for (int e = 1; e < 3; ++e)
                    if (sm.active[e] && !(used & (1 << e)) && sm.spec_t[e] == sm.req_t && sm.spec_ref[e] == sm.req_ref) next = e;
                if (next < 0) break;
                slot = next;
            }
        }
        phi_dev_post(dt, sign, sm, L.req, r);
        s_evals[r] = evals;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        PhiDev *q = L.q;
        int64_t evals = L.evals;
        if (live) {
            evals += s_evals[0] + s_evals[1];
            L.evals = evals;
        }
        if (publish) {                                           // (the host waits for the last enqueued round only)
            unsigned long long seal = seal_bits(ticket);         // (wait_sealed: the host checks it before it reads the summary)
            seal ^= phi_dev_summary(q[0], L.norm[0], out + kSumBase);
            seal ^= phi_dev_summary(q[1], L.norm[1], out + kSumBase + kSumStride);
            unsigned long long *ob = reinterpret_cast<unsigned long long *>(out);
            __hip_atomic_store(ob + kSumBase + 2 * kSumStride, seal_bits((double)evals), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            seal ^= seal_bits((double)evals);
            __hip_atomic_store(ob + kSumBase + 2 * kSumStride + 1, seal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    __syncthreads();
    if (live) search_from_lds(S, &L);
}

static int dev_search_rounds() {
    const int r = getenv("DZO_TUNE_BFGS_DEV_ROUNDS") ? atoi(getenv("DZO_TUNE_BFGS_DEV_ROUNDS")) : 2;
    return r < 1 ? 1 : (r > 8 ? 8 : r);
}

// norms + both searches, one host wait.  *done = false: not applicable (nothing was enqueued).
static int32_t bfgs_dev_search(dzo_bfgs_s *o, double step_length, double *grad_norm, double *bfgs_norm_v, double *t_a, double *f_a,
                               double *t_b, double *f_b, bool *done) {
    const bool enabled = getenv("DZO_TUNE_BFGS_DEV_SEARCH") ? atoi(getenv("DZO_TUNE_BFGS_DEV_SEARCH")) != 0 : true;      // (read per call: tests switch it)
    const bool dual = getenv("DZO_TUNE_BFGS_DUAL_SEARCH") ? atoi(getenv("DZO_TUNE_BFGS_DUAL_SEARCH")) != 0 : true;
    *done = false;
    if (!enabled || !dual || !o->dsearch || o->objective || o->constraint || !o->problem || o->problem->kind != DZO_PROBLEM_QUADRATIC ||
        o->problem->l2 != 0.0 || o->problem->cons_on || o->n > 65536 || o->problem->scratch_doubles < 6 * o->n)
        return DZO_OK;
    const int32_t dt = o->dtype;
    const size_t es = dtype_size(dt);
    const size_t vbytes = (size_t)((o->n + 63) / 64 * 64) * es;
    hipStream_t s = o->stream;
    const void *dirs[2] = {o->g, o->d};
    {
        DZO_TIMED("bfgs_norm_pair", s);
        DZO_DISPATCH(dt, hipLaunchKernelGGL(norm2_pair_begin_kernel<T>, dim3(1), dim3(kBlock), 0, s, o->n, (const T *)o->g, (const T *)o->d,
                                            o->dsearch, o->f, step_length, dt, o->sign));
    }
    void *pts[2][3] = {{o->scratch, o->spec_buf[0], o->spec_buf[1]}, {o->scratch2, o->spec_buf[2], o->spec_buf[3]}};   // trial-point buffer of request slot e
    int round = 0;
    auto enqueue_round = [&](bool publish) -> int32_t {
        round += 1;
        PhiDirHost req[2];
        for (int r = 0; r < 2; ++r) {
            req[r].dir = dirs[r];
            for (int e = 0; e < 3; ++e) {
                req[r].point_out[e] = pts[r][e];
                req[r].grad_out[e] = (char *)o->grad_pool + (size_t)((round % 4) * 6 + r * 3 + e) * vbytes;
                req[r].active[e] = true;                         // (the kernel takes these three from the device's requests)
            }
        }
        o->ticket += 1.0;
        DZO_REQUIRE(problem_phi6_async(o->problem, s, o->x, req, o->phi_flags(), o->host_dev, o->ticket, &o->dsearch->req), DZO_ERR_HIP,
                    "the two-direction objective kernel refused a launch it had accepted");
        {
            DZO_TIMED("bfgs_search_advance", s);
            hipLaunchKernelGGL(finish_phi6_advance_kernel, dim3(1), dim3(kBlock), 0, s, (const double *)o->problem->scratch, o->n, 0.5,
                               o->host_dev, o->phi_flags(), o->ticket, o->dsearch, round, dt, o->max_increases, o->sign, publish ? 1 : 0);
        }
        DZO_HIP(hipGetLastError());
        return DZO_OK;
    };
    const double *sum = o->host + kSumBase;
    for (int batch = dev_search_rounds();; batch = 1) {
        for (int i = 0; i < batch; ++i) DZO_TRY(enqueue_round(i + 1 == batch));
        DZO_TRY(wait_sealed(s, o->host + kSumBase, 2 * kSumStride + 1, o->host + kSumBase + 2 * kSumStride + 1, o->ticket));
        if ((((int64_t)sum[0]) & 0xFF) == 0 && (((int64_t)sum[kSumStride]) & 0xFF) == 0) break;   // neither search wants another evaluation
    }
    *grad_norm = sum[4]; *bfgs_norm_v = sum[kSumStride + 4];
    o->evals += (int64_t)sum[2 * kSumStride];
    o->best_grad[0] = o->best_grad[1] = nullptr;
    double tt[2], ff[2];
    for (int r = 0; r < 2; ++r) {
        const double *q = sum + kSumStride * r;
        const int64_t packed = (int64_t)q[0];
        const int state = (int)((packed >> 8) & 0xFF);
        const double t0 = q[3];
        if (state == kPhiSequential) {                           // the rare tiny-step path (:91-101): the sequential code
            DZO_TRY(bfgs_quadratic_search(o, dirs[r], o->f, t0, &tt[r], &ff[r]));
            continue;
        }
        tt[r] = q[1]; ff[r] = q[2];
        const int bg = (int)((packed >> 16) & 0xFFFF) - 1, br = (int)(packed >> 32);
        if (state == kPhiDone && tt[r] != 0.0 && bg >= 0 && round - br < 4)
            o->best_grad[r] = (char *)o->grad_pool + (size_t)bg * vbytes;
    }
    *t_a = tt[0]; *f_a = ff[0]; *t_b = tt[1]; *f_b = ff[1];
    *done = true;
    return DZO_OK;
}

// make the stored matrix whole again (upper <- lower) before anything reads it as a full matrix
static int32_t bfgs_mirror(dzo_bfgs_s *o) {
    if (!o->upper_stale) return DZO_OK;
    DZO_TIMED("bfgs_mirror", o->stream);
    const int grid = stream_grid(o->n * o->n, 4);
    DZO_DISPATCH(o->dtype, hipLaunchKernelGGL(tri_mirror_kernel<T>, dim3(grid), dim3(kBlock), 0, o->stream, o->n, (T *)o->H));
    DZO_HIP(hipGetLastError());
    o->upper_stale = false;
    return DZO_OK;
}

static int32_t bfgs_identity(dzo_bfgs_s *o) {
    DZO_TIMED("bfgs_identity", o->stream);
    const int grid = stream_grid(o->n * o->n, 4);
    DZO_DISPATCH(o->dtype, hipLaunchKernelGGL(identity_kernel<T>, dim3(grid), dim3(kBlock), 0, o->stream, o->n, (T *)o->H));
    DZO_HIP(hipGetLastError());
    o->upper_stale = false;                                      // (both triangles written)
    return DZO_OK;
}

// :943-950 / :971-978
static int32_t bfgs_move(dzo_bfgs_s *o, double t, const void *dir, const void *grad_at_new_point = nullptr) {
    hipStream_t s = o->stream;
    if (grad_at_new_point && !o->gradient && !o->constraint && !(o->problem && o->problem->cons_on)) {
        DZO_TIMED("bfgs_move", s);
        const int grid = stream_grid(o->n, 1);
        DZO_DISPATCH(o->dtype, hipLaunchKernelGGL(bfgs_move_with_gradient_kernel<T>, dim3(grid), dim3(kBlock), 0, s, o->n, (T *)o->x,
                                                  (const T *)dir, (T *)o->g, (const T *)grad_at_new_point, (T)t, (T *)o->dx, (T *)o->dg));
        DZO_HIP(hipGetLastError());
        return DZO_OK;
    }
    {
        DZO_TIMED("bfgs_move", s);
        const int grid = stream_grid(o->n, 1);
        DZO_DISPATCH(o->dtype, hipLaunchKernelGGL(bfgs_move_kernel<T>, dim3(grid), dim3(kBlock), 0, s, o->n, (T *)o->x,
                                                  (const T *)dir, (const T *)o->g, (T)t, (T *)o->dx, (T *)o->dg));
    }
    DZO_HIP(hipGetLastError());
    if (!o->constraint && o->problem && o->problem->cons_on)     // :946 built-in box
        DZO_TRY(box_clamp_async(s, o->n, o->dtype, o->x, o->problem->cons_lo, o->problem->cons_hi));
    if (o->constraint) {                                         // :946-947
        DZO_HIP(hipStreamSynchronize(s));
        DZO_REQUIRE(o->constraint(o->cb_ctx, o->x) != 0, DZO_ERR_ASSERT,
                    "@assert constraint_success (legacy/DZOptimization.jl:947)");
    }
    if (grad_at_new_point && !o->gradient && !o->constraint && !(o->problem && o->problem->cons_on))
        DZO_HIP(hipMemcpyAsync(o->g, grad_at_new_point, (size_t)o->n * dtype_size(o->dtype), hipMemcpyDeviceToDevice, s));   // :948, already computed by the line search
    else
        DZO_TRY(bfgs_grad(o));                                   // :948
    // :949-950  (-x_old) + x_new == x_new - x_old exactly
    DZO_DISPATCH(o->dtype, launch_axpby<T>(s, o->n, (T)1, (const T *)o->x, (T)-1, (T *)o->dx));
    DZO_DISPATCH(o->dtype, launch_axpby<T>(s, o->n, (T)1, (const T *)o->g, (T)-1, (T *)o->dg));
    DZO_HIP(hipGetLastError());
    return DZO_OK;
}

static int32_t bfgs_step(dzo_bfgs_s *o) {
    if (o->has_terminated) return DZO_OK;                        // :893
    problem_view_sync(o->problem);
    const int32_t dt = o->dtype;
    const size_t bytes = (size_t)o->n * dtype_size(dt);
    const double step_length = o->last_step_length;              // :918
    double grad_norm, bfgs_norm_v;
    double t_g, f_g, t_b, f_b;
    bool dual = false;
    DZO_TRY(bfgs_dev_search(o, step_length, &grad_norm, &bfgs_norm_v, &t_g, &f_g, &t_b, &f_b, &dual));   // :921-932 with one host wait
    if (!dual) DZO_TRY(bfgs_norm_pair(o, o->g, o->d, &grad_norm, &bfgs_norm_v));   // :921, :928
    if (!dual) DZO_TRY(bfgs_dual_search(o, o->g, round_to_dtype(dt, step_length / grad_norm), o->d, round_to_dtype(dt, step_length / bfgs_norm_v),
                             &t_g, &f_g, &t_b, &f_b, &dual));    // :922-925 and :929-932 side by side
    if (!dual) {
        o->best_grad[0] = o->best_grad[1] = nullptr;
        DZO_TRY(bfgs_quadratic_search(o, o->g, o->f, round_to_dtype(dt, step_length / grad_norm), &t_g, &f_g));   // :922-925
        DZO_TRY(bfgs_quadratic_search(o, o->d, o->f, round_to_dtype(dt, step_length / bfgs_norm_v), &t_b, &f_b)); // :929-932
    }
    if (f_b < o->f && !(f_b > f_g)) {                            // :934
        o->f = f_b;                                              // :937
        o->last_step_length = round_to_dtype(dt, t_b * bfgs_norm_v);   // :938
        o->last_step_type = DZO_STEP_BFGS;                       // :939
        o->iteration_count += 1;                                 // :940
        DZO_TRY(bfgs_move(o, t_b, o->d, o->best_grad[1]));       // :943-950
        // :953-960 update_inverse_hessian!(H, -t_b, d, dg, scratch) fused with d = H*g
        if (o->tri) {
            DZO_DISPATCH(dt, launch_bfgs_update_tri<T>(o->stream, o->n, (T *)o->H, (T)(-t_b), (const T *)o->d, (const T *)o->dg,
                                                       (T *)o->scratch, (const T *)o->g, (T *)o->d_alt, o->upd_part, o->tri_rowpart,
                                                       o->tri_colpart, (T *)o->tri_dummy));
            o->upper_stale = true;
        } else if (o->n >= 65535LL * kColsPerBlock) {
            DZO_DISPATCH(dt, launch_bfgs_update<T>(o->stream, o->n, (T *)o->H, (T)(-t_b), (T *)o->d, (const T *)o->dg,
                                                   (T *)o->scratch, (const T *)o->g, (T *)o->d_alt, o->scalars()));
        } else {
            DZO_DISPATCH(dt, launch_bfgs_update_fused<T>(o->stream, o->n, (T *)o->H, (T)(-t_b), (const T *)o->d, (const T *)o->dg,
                                                         (T *)o->scratch, (const T *)o->g, (T *)o->d_alt, o->upd_part));
        }
        DZO_HIP(hipGetLastError());
        std::swap(o->d, o->d_alt);
    } else if (f_g < o->f) {                                     // :962
        o->f = f_g;                                              // :965
        o->last_step_length = round_to_dtype(dt, t_g * grad_norm);   // :966
        o->last_step_type = DZO_STEP_GRADIENT_DESCENT;           // :967
        o->iteration_count += 1;                                 // :968
        DZO_TRY(bfgs_move(o, t_g, o->g, o->best_grad[0]));       // :971-978
        DZO_TRY(bfgs_identity(o));                               // :981
        DZO_HIP(hipMemcpyAsync(o->d, o->g, bytes, hipMemcpyDeviceToDevice, o->stream));   // :984-986
    } else {
        o->has_terminated = true;                                // :989
    }
    // (no stream sync here: include/dzo.h -- step functions return once the host-side decisions are
    // made; the getters and the next step's first read-back synchronise)
    return DZO_OK;
}

static int32_t bfgs_alloc(dzo_bfgs_s *o) {
    const size_t es = dtype_size(o->dtype);
    const size_t vbytes = (size_t)((o->n + 63) / 64 * 64) * es;
    DZO_HIP(hipStreamCreateWithFlags(&o->stream, hipStreamNonBlocking));
    void **vecs[] = {&o->x, &o->g, &o->dx, &o->dg, &o->d, &o->d_alt, &o->scratch, &o->ref_point, &o->scratch2, &o->ref_point2,
                     &o->spec_buf[0], &o->spec_buf[1], &o->spec_buf[2], &o->spec_buf[3]};
    for (void **v : vecs) {
        hipError_t e = hipMalloc(v, vbytes);
        if (e != hipSuccess) { set_error("out of device memory allocating BFGS vectors"); (void)hipGetLastError(); return DZO_ERR_NOMEM; }
        DZO_HIP(hipMemset(*v, 0, vbytes));                      // :777-778 zero deltas
    }
    if (!o->no_hessian) {
        hipError_t e = hipMalloc(&o->H, (size_t)o->n * (size_t)o->n * es);
        if (e != hipSuccess) {
            set_error("out of device memory allocating the %lld x %lld inverse Hessian", (long long)o->n, (long long)o->n);
            (void)hipGetLastError(); return DZO_ERR_NOMEM;
        }
    }
    {
        // default: from H = 128 MiB up (config 2: n = 4096 fp64).  Measured on MI355X: n = 4096 (H fits the
        // Infinity Cache) update + direction 68 us against 78 us for the full-storage kernels, the step rate
        // the same; n = 8192 (512 MiB, HBM) 196 us against 332 us, step! 1446 against 1218 per second.
        // Below that size the two extra launches cost more than the halved traffic saves.
        const char *e = getenv("DZO_TUNE_BFGS_TRI_MIN_N");
        const bool big = e ? o->n >= atoll(e) : (size_t)o->n * (size_t)o->n * es >= (128u << 20);
        o->tri = !o->no_hessian && o->n % 2 == 0 && big && o->n < 65535LL * kTriCW;
        if (o->tri) {
            const size_t nw = (size_t)((o->n + kTriCW - 1) / kTriCW), npn = (size_t)((o->n + kTriPH - 1) / kTriPH);
            if (hipMalloc((void **)&o->tri_rowpart, nw * (size_t)o->n * sizeof(double)) != hipSuccess ||
                hipMalloc((void **)&o->tri_colpart, npn * (size_t)o->n * sizeof(double)) != hipSuccess ||
                hipMalloc(&o->tri_dummy, 64) != hipSuccess) {
                (void)hipGetLastError();
                o->tri = false;                                  // not enough memory for the partial sums: full-storage kernels
            }
        }
    }
    DZO_HIP(hipMalloc(&o->grad_pool, 24 * vbytes));
    DZO_HIP(hipMalloc((void **)&o->upd_part, sizeof(double) * (size_t)(2 * ((o->n + kColsPerBlock - 1) / kColsPerBlock) + 8)));   // (>= 2 ceil(n / kTriRI) too)
    DZO_HIP(hipMalloc((void **)&o->ws, sizeof(double) * (kMaxPartialBlocks + 48)));
    DZO_HIP(hipMemset(o->ws, 0, sizeof(double) * (kMaxPartialBlocks + 48)));
    DZO_HIP(hipHostMalloc((void **)&o->host, sizeof(double) * kHostDoubles, hipHostMallocMapped | hipHostMallocCoherent));
    DZO_HIP(hipHostGetDevicePointer((void **)&o->host_dev, o->host, 0));
    for (int i = 0; i < kHostDoubles; ++i) o->host[i] = 0;
    DZO_HIP(hipMalloc((void **)&o->dsearch, sizeof(BfgsSearchDev)));
    DZO_HIP(hipMemset(o->dsearch, 0, sizeof(BfgsSearchDev)));
    DZO_HIP(hipDeviceSynchronize());
    return DZO_OK;
}

template <typename S, typename D>
__global__ __launch_bounds__(kBlock) void cast_kernel(int64_t n, const S *__restrict__ src, D *__restrict__ dst) {
    const int64_t nthreads = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += nthreads) dst[i] = (D)src[i];   // T.(v)
}

static int32_t cast_async(hipStream_t s, int64_t n, int32_t src_dtype, const void *src, int32_t dst_dtype, void *dst) {
    const int grid = stream_grid(n, 4);
    if (src_dtype == dst_dtype) {
        DZO_HIP(hipMemcpyAsync(dst, src, (size_t)n * dtype_size(src_dtype), hipMemcpyDeviceToDevice, s));
    } else if (src_dtype == DZO_F64) {
        hipLaunchKernelGGL((cast_kernel<double, float>), dim3(grid), dim3(kBlock), 0, s, n, (const double *)src, (float *)dst);
    } else {
        hipLaunchKernelGGL((cast_kernel<float, double>), dim3(grid), dim3(kBlock), 0, s, n, (const float *)src, (double *)dst);
    }
    DZO_HIP(hipGetLastError());
    return DZO_OK;
}

// BFGSOptimizer(::Type{T}, objective, gradient!, constraint!, opt)  legacy/DZOptimization.jl:819-862:
// re-types a running optimizer; `o` already carries the new dtype and the new callbacks/problem.
static int32_t bfgs_convert(dzo_bfgs_s *src, dzo_bfgs_s *o) {
    DZO_TRY(bfgs_mirror(src));
    DZO_HIP(hipStreamSynchronize(src->stream));
    DZO_TRY(bfgs_alloc(o));
    hipStream_t s = o->stream;
    const int64_t n = o->n;
    DZO_TRY(cast_async(s, n, src->dtype, src->x, o->dtype, o->x));                  // :825 T.(current_point)
    if (o->constraint) {
        DZO_HIP(hipStreamSynchronize(s));
        DZO_REQUIRE(o->constraint(o->cb_ctx, o->x) != 0, DZO_ERR_ASSERT, "@assert constraint_success (legacy/DZOptimization.jl:826-827)");
    } else if (o->problem && o->problem->cons_on) {
        DZO_TRY(box_clamp_async(s, n, o->dtype, o->x, o->problem->cons_lo, o->problem->cons_hi));
    }
    DZO_TRY(bfgs_eval(o, o->x, &o->f));                                              // :828
    DZO_REQUIRE(!(o->f != o->f), DZO_ERR_ASSERT, "@assert !isnan(initial_objective_value) (legacy/DZOptimization.jl:829)");
    DZO_TRY(bfgs_grad(o));                                                           // :830-831
    DZO_TRY(cast_async(s, n * n, src->dtype, src->H, o->dtype, o->H));              // :832
    DZO_DISPATCH(o->dtype, launch_symv<T>(s, n, (const T *)o->H, (const T *)o->g, (T *)o->d));   // :833-836 mul!
    DZO_HIP(hipGetLastError());
    DZO_TRY(cast_async(s, n, src->dtype, src->dx, o->dtype, o->dx));                // :853
    DZO_TRY(cast_async(s, n, src->dtype, src->dg, o->dtype, o->dg));                // :854
    o->iteration_count = src->iteration_count;                                       // :848
    o->has_terminated = false;                                                       // :849
    o->last_step_length = round_to_dtype(o->dtype, src->last_step_length);           // :855
    o->last_step_type = src->last_step_type;                                         // :856
    o->max_increases = src->max_increases;
    DZO_HIP(hipStreamSynchronize(s));
    return DZO_OK;
}

static int32_t bfgs_create_common(dzo_bfgs_s *o, const void *x0_dev, double initial_step_length) {
    const size_t es = dtype_size(o->dtype);
    // (the legacy optimizers predate KernelAbstractions; the same-device rule of src/DZOptimization.jl:363-364 is
    // applied to the one array they are given)
    DZO_TRY(require_same_backend("BFGSOptimizer", "src/DZOptimization.jl:363-364", x0_dev, "initial_point", nullptr, ""));
    DZO_TRY(bfgs_alloc(o));
    DZO_HIP(hipMemcpy(o->x, x0_dev, (size_t)o->n * es, hipMemcpyDeviceToDevice));   // :769 copy
    DZO_HIP(hipDeviceSynchronize());   // null-stream memset/D2D copies are asynchronous to the host and to our non-blocking streams
    if (o->constraint)
        DZO_REQUIRE(o->constraint(o->cb_ctx, o->x) != 0, DZO_ERR_ASSERT, "@assert constraint_success (legacy/DZOptimization.jl:770-771)");
    else if (o->problem && o->problem->cons_on)
        DZO_TRY(box_clamp_async(o->stream, o->n, o->dtype, o->x, o->problem->cons_lo, o->problem->cons_hi));
    DZO_TRY(bfgs_eval(o, o->x, &o->f));                          // :772
    DZO_REQUIRE(!(o->f != o->f), DZO_ERR_ASSERT, "@assert !isnan(initial_objective_value) (legacy/DZOptimization.jl:773)");
    DZO_TRY(bfgs_grad(o));                                       // :775-776
    o->last_step_length = round_to_dtype(o->dtype, initial_step_length);   // :779
    o->last_step_type = DZO_STEP_NULL;                           // :780
    DZO_TRY(bfgs_identity(o));                                   // :781-783
    DZO_HIP(hipMemcpyAsync(o->d, o->g, (size_t)o->n * es, hipMemcpyDeviceToDevice, o->stream));   // :784
    DZO_HIP(hipStreamSynchronize(o->stream));
    return DZO_OK;
}

// ---------------------------------------------------------------------------------------------
// Legacy GradientDescentOptimizer (legacy/DZOptimization.jl:305-449) with QuadraticLineSearch
// (:181-216) as line_search_function!: same search code, sign = +1, d = next_step_direction.
// ---------------------------------------------------------------------------------------------
static int32_t gd_inv_norm(dzo_bfgs_s *o, const void *v, double *out) {     // Kernels.jl:141 rsqrt(norm2(x))
    double ss = 0;
    DZO_TRY(dot_blocking(o->stream, o->n, o->dtype, v, v, o->partials(), o->host, &ss));
    *out = o->dtype == DZO_F32 ? (double)(1.0f / sqrtf((float)ss)) : 1.0 / sqrt(ss);
    return DZO_OK;
}

static int32_t gd_create_common(dzo_bfgs_s *o, const void *x0_dev, double initial_step_length) {
    const size_t es = dtype_size(o->dtype);
    o->no_hessian = true;
    o->sign = 1.0;
    DZO_TRY(require_same_backend("GradientDescentOptimizer", "src/DZOptimization.jl:363-364", x0_dev, "initial_point", nullptr, ""));
    DZO_TRY(bfgs_alloc(o));
    DZO_HIP(hipMemcpy(o->x, x0_dev, (size_t)o->n * es, hipMemcpyDeviceToDevice));   // :339 collect
    DZO_HIP(hipDeviceSynchronize());
    if (o->constraint)
        DZO_REQUIRE(o->constraint(o->cb_ctx, o->x) != 0, DZO_ERR_ASSERT, "@assert constraint_function!(current_point) (legacy/DZOptimization.jl:340)");
    else if (o->problem && o->problem->cons_on)
        DZO_TRY(box_clamp_async(o->stream, o->n, o->dtype, o->x, o->problem->cons_lo, o->problem->cons_hi));
    DZO_TRY(bfgs_eval(o, o->x, &o->f));                          // :343
    DZO_TRY(bfgs_grad(o));                                       // :347-348
    o->last_step_length = 0;                                     // :351
    double ign = 0;
    DZO_TRY(gd_inv_norm(o, o->g, &ign));                         // :352
    if (std::isfinite(ign)) {                                    // :354-357  d = g * (-step * inv_norm)
        const double sc = round_to_dtype(o->dtype, -initial_step_length * ign);
        DZO_DISPATCH(o->dtype, launch_scal_oop<T>(o->stream, o->n, (T *)o->d, (T)sc, (const T *)o->g));
        DZO_HIP(hipGetLastError());
    }
    o->has_terminated = !std::isfinite(o->f) || !std::isfinite(ign);   // :364-366
    DZO_HIP(hipStreamSynchronize(o->stream));
    return DZO_OK;
}

static int32_t gd_step(dzo_bfgs_s *o) {
    if (o->has_terminated) return DZO_OK;                        // :402
    problem_view_sync(o->problem);
    const int32_t dt = o->dtype;
    const size_t bytes = (size_t)o->n * dtype_size(dt);
    hipStream_t s = o->stream;
    double t, fv;
    DZO_TRY(bfgs_quadratic_search(o, o->d, o->f, 1.0, &t, &fv)); // :405-407 (bracket starts at step size 1, :89)
    if (t == 0.0 || !(fv < o->f)) { o->has_terminated = true; return DZO_OK; }   // :410-414
    o->iteration_count += 1;                                     // :415
    DZO_HIP(hipMemcpyAsync(o->dx, o->x, bytes, hipMemcpyDeviceToDevice, s));      // :418
    DZO_DISPATCH(dt, launch_axpy<T>(s, o->n, (T)t, (const T *)o->d, (T *)o->x));  // :419
    if (o->constraint) {                                         // :420
        DZO_HIP(hipStreamSynchronize(s));
        DZO_REQUIRE(o->constraint(o->cb_ctx, o->x) != 0, DZO_ERR_ASSERT, "@assert constraint_function!(current_point) (legacy/DZOptimization.jl:420)");
    } else if (o->problem && o->problem->cons_on) {
        DZO_TRY(box_clamp_async(s, o->n, dt, o->x, o->problem->cons_lo, o->problem->cons_hi));
    }
    DZO_DISPATCH(dt, launch_axpby<T>(s, o->n, (T)1, (const T *)o->x, (T)-1, (T *)o->dx));   // :423 delta!(dx, x): dx = x - dx
    double ss = 0;
    DZO_TRY(dot_blocking(s, o->n, dt, o->dx, o->dx, o->partials(), o->host, &ss));
    const double step_length = dt == DZO_F32 ? (double)sqrtf((float)ss) : sqrt(ss);        // :424
    o->last_step_length = step_length;                           // :425
    o->df = round_to_dtype(dt, fv - o->f);                       // :428-429
    o->f = fv;                                                   // :430
    DZO_HIP(hipMemcpyAsync(o->dg, o->g, bytes, hipMemcpyDeviceToDevice, s));      // :433
    DZO_TRY(bfgs_grad(o));                                       // :434
    DZO_DISPATCH(dt, launch_axpby<T>(s, o->n, (T)1, (const T *)o->g, (T)-1, (T *)o->dg));   // :435
    double ign = 0;
    DZO_TRY(gd_inv_norm(o, o->g, &ign));                         // :438
    if (!std::isfinite(ign)) { o->has_terminated = true; return DZO_OK; }          // :439-442
    const double sc = round_to_dtype(dt, -step_length * ign);
    DZO_DISPATCH(dt, launch_scal_oop<T>(s, o->n, (T *)o->d, (T)sc, (const T *)o->g));       // :445-446
    DZO_HIP(hipGetLastError());
    DZO_HIP(hipStreamSynchronize(s));
    return DZO_OK;
}

}  // namespace dzo

using namespace dzo;

extern "C" {

int32_t dzo_gd_create_problem(dzo_problem_t problem, const void *x0_dev, double initial_step_length, dzo_bfgs_t *out) {
    DZO_TRY(require_init());
    DZO_REQUIRE(problem && x0_dev && out, DZO_ERR_INVALID, "null argument");
    dzo_bfgs_s *o = new dzo_bfgs_s();
    o->n = problem->n; o->dtype = problem->dtype;
    int32_t rc = problem_view_create(problem, &o->problem);
    if (rc == DZO_OK) rc = gd_create_common(o, x0_dev, initial_step_length);
    if (rc != DZO_OK) { dzo_bfgs_destroy(o); return rc; }
    *out = o;
    return DZO_OK;
}

int32_t dzo_gd_create_callbacks(dzo_constraint_fn constraint, dzo_objective_fn objective, dzo_gradient_fn gradient,
                                void *cb_ctx, int64_t n, int32_t dtype, const void *x0_dev, double initial_step_length,
                                dzo_bfgs_t *out) {
    DZO_TRY(require_init());
    DZO_REQUIRE(objective && gradient && x0_dev && out && n >= 1, DZO_ERR_INVALID, "bad argument");
    DZO_REQUIRE(dtype == DZO_F32 || dtype == DZO_F64, DZO_ERR_INVALID, "bad dtype %d", dtype);
    dzo_bfgs_s *o = new dzo_bfgs_s();
    o->n = n; o->dtype = dtype; o->objective = objective; o->gradient = gradient; o->constraint = constraint;
    o->cb_ctx = cb_ctx;
    int32_t rc = gd_create_common(o, x0_dev, initial_step_length);
    if (rc != DZO_OK) { dzo_bfgs_destroy(o); return rc; }
    *out = o;
    return DZO_OK;
}

int32_t dzo_gd_step(dzo_bfgs_t o) {
    DZO_REQUIRE(o && o->no_hessian, DZO_ERR_INVALID, "not a GradientDescentOptimizer handle");
    return gd_step(o);
}

int32_t dzo_bfgs_destroy(dzo_bfgs_t o) {
    if (!o) return DZO_OK;
    if (o->stream) (void)hipStreamSynchronize(o->stream);
    void *ptrs[] = {o->x, o->g, o->dx, o->dg, o->d, o->d_alt, o->scratch, o->ref_point, o->scratch2, o->ref_point2, o->spec_buf[0], o->spec_buf[1], o->spec_buf[2],
                    o->spec_buf[3], o->grad_pool, o->H, o->ws, o->upd_part, o->tri_rowpart, o->tri_colpart, o->tri_dummy};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (o->host) (void)hipHostFree(o->host);
    if (o->dsearch) (void)hipFree(o->dsearch);
    problem_view_destroy(o->problem);
    if (o->stream) (void)hipStreamDestroy(o->stream);
    delete o;
    return DZO_OK;
}

int32_t dzo_bfgs_create_callbacks(dzo_objective_fn objective, dzo_gradient_fn gradient, dzo_constraint_fn constraint,
                                  void *cb_ctx, int64_t n, int32_t dtype, const void *x0_dev, double initial_step_length,
                                  dzo_bfgs_t *out) {
    DZO_TRY(require_init());
    DZO_REQUIRE(objective && gradient && x0_dev && out && n >= 1, DZO_ERR_INVALID, "bad argument");
    DZO_REQUIRE(dtype == DZO_F32 || dtype == DZO_F64, DZO_ERR_INVALID, "bad dtype %d", dtype);
    dzo_bfgs_s *o = new dzo_bfgs_s();
    o->n = n; o->dtype = dtype; o->objective = objective; o->gradient = gradient; o->constraint = constraint;
    o->cb_ctx = cb_ctx;
    int32_t rc = bfgs_create_common(o, x0_dev, initial_step_length);
    if (rc != DZO_OK) { dzo_bfgs_destroy(o); return rc; }
    *out = o;
    return DZO_OK;
}

int32_t dzo_bfgs_create_problem(dzo_problem_t problem, const void *x0_dev, double initial_step_length, dzo_bfgs_t *out) {
    DZO_TRY(require_init());
    DZO_REQUIRE(problem && x0_dev && out, DZO_ERR_INVALID, "null argument");
    dzo_bfgs_s *o = new dzo_bfgs_s();
    o->n = problem->n; o->dtype = problem->dtype;
    int32_t rc = problem_view_create(problem, &o->problem);        // private partial-sum workspace per optimizer
    if (rc == DZO_OK) rc = bfgs_create_common(o, x0_dev, initial_step_length);
    if (rc != DZO_OK) { dzo_bfgs_destroy(o); return rc; }
    *out = o;
    return DZO_OK;
}

int32_t dzo_bfgs_convert_problem(dzo_bfgs_t src, dzo_problem_t problem, dzo_bfgs_t *out) {
    DZO_TRY(require_init());
    DZO_REQUIRE(src && problem && out, DZO_ERR_INVALID, "null argument");
    DZO_REQUIRE(problem->n == src->n, DZO_ERR_INVALID, "problem size does not match the optimizer");
    dzo_bfgs_s *o = new dzo_bfgs_s();
    o->n = src->n; o->dtype = problem->dtype;
    int32_t rc = problem_view_create(problem, &o->problem);
    if (rc == DZO_OK) rc = bfgs_convert(src, o);
    if (rc != DZO_OK) { dzo_bfgs_destroy(o); return rc; }
    *out = o;
    return DZO_OK;
}

int32_t dzo_bfgs_convert_callbacks(dzo_bfgs_t src, int32_t dtype, dzo_objective_fn objective, dzo_gradient_fn gradient,
                                   dzo_constraint_fn constraint, void *cb_ctx, dzo_bfgs_t *out) {
    DZO_TRY(require_init());
    DZO_REQUIRE(src && objective && gradient && out, DZO_ERR_INVALID, "null argument");
    DZO_REQUIRE(dtype == DZO_F32 || dtype == DZO_F64, DZO_ERR_INVALID, "bad dtype %d", dtype);
    dzo_bfgs_s *o = new dzo_bfgs_s();
    o->n = src->n; o->dtype = dtype; o->objective = objective; o->gradient = gradient; o->constraint = constraint;
    o->cb_ctx = cb_ctx;
    int32_t rc = bfgs_convert(src, o);
    if (rc != DZO_OK) { dzo_bfgs_destroy(o); return rc; }
    *out = o;
    return DZO_OK;
}

int32_t dzo_bfgs_step(dzo_bfgs_t o) {
    DZO_REQUIRE(o, DZO_ERR_INVALID, "null optimizer");
    DZO_REQUIRE(!o->no_hessian, DZO_ERR_INVALID, "GradientDescentOptimizer handle: use dzo_gd_step");
    return bfgs_step(o);
}

int32_t dzo_bfgs_update(int64_t n, int32_t dtype, void *H_dev, double step_length, void *d_dev, const void *dg_dev,
                        void *scratch_dev, const void *g_dev, void *d_next_dev) {
    DZO_TRY(require_init());
    DZO_REQUIRE(n >= 1 && H_dev && d_dev && dg_dev && scratch_dev, DZO_ERR_INVALID, "bad argument");
    DZO_REQUIRE((g_dev == nullptr) == (d_next_dev == nullptr), DZO_ERR_INVALID, "g and d_next must be given together");
    DZO_REQUIRE(d_next_dev != d_dev || !d_next_dev, DZO_ERR_INVALID, "d_next must not alias d");
    hipStream_t s = ctx().stream;
    DZO_DISPATCH(dtype, launch_bfgs_update<T>(s, n, (T *)H_dev, (T)step_length, (T *)d_dev, (const T *)dg_dev,
                                              (T *)scratch_dev, (const T *)g_dev, (T *)d_next_dev, ctx().scratch + kMaxPartialBlocks));
    DZO_HIP(hipGetLastError());
    DZO_HIP(hipStreamSynchronize(s));
    return DZO_OK;
}

int32_t dzo_bfgs_update_mfma(int64_t n, int32_t dtype, void *H_dev, double step_length, void *d_dev, const void *dg_dev,
                             void *scratch_dev) {
    DZO_TRY(require_init());
    DZO_REQUIRE(n >= 16 && H_dev && d_dev && dg_dev && scratch_dev, DZO_ERR_INVALID, "bad argument");
    DZO_REQUIRE(dtype == DZO_F64 && n % 16 == 0, DZO_ERR_UNSUPPORTED,
                "the MFMA update is fp64 only (v_mfma_f64_16x16x4_f64) and needs n %% 16 == 0");
    hipStream_t s = ctx().stream;
    double *scal = ctx().scratch + kMaxPartialBlocks;
    launch_symv<double>(s, n, (const double *)H_dev, (const double *)dg_dev, (double *)scratch_dev);       // :875
    {
        DZO_TIMED("bfgs_scalars", s);
        hipLaunchKernelGGL(bfgs_scalars_kernel<double>, dim3(1), dim3(kBlock), 0, s, n, (double *)d_dev,
                           (const double *)dg_dev, (const double *)scratch_dev, step_length, scal);        // :873-876
    }
    {
        DZO_TIMED("bfgs_update_mfma", s);
        const int64_t tiles = n / 16;
        const int64_t jobs = tiles * ((tiles + 3) / 4);
        int64_t blocks = (jobs + kWaves - 1) / kWaves;
        if (blocks > (int64_t)ctx().cus * 16) blocks = (int64_t)ctx().cus * 16;
        hipLaunchKernelGGL(bfgs_update_mfma_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, s, n, (double *)H_dev,
                           (const double *)d_dev, (const double *)scratch_dev, (const double *)scal);
    }
    DZO_HIP(hipGetLastError());
    DZO_HIP(hipStreamSynchronize(s));
    return DZO_OK;
}

int32_t dzo_symv(int64_t n, int32_t dtype, const void *H_dev, const void *v_dev, void *out_dev) {
    DZO_TRY(require_init());
    DZO_REQUIRE(n >= 1 && H_dev && v_dev && out_dev, DZO_ERR_INVALID, "bad argument");
    DZO_DISPATCH(dtype, launch_symv<T>(ctx().stream, n, (const T *)H_dev, (const T *)v_dev, (T *)out_dev));
    DZO_HIP(hipGetLastError());
    DZO_HIP(hipStreamSynchronize(ctx().stream));
    return DZO_OK;
}

int32_t dzo_bfgs_line_search(dzo_bfgs_t o, int32_t use_gradient_direction, double t0, double *t_best, double *f_best) {
    DZO_REQUIRE(o && t_best && f_best, DZO_ERR_INVALID, "null argument");
    problem_view_sync(o->problem);
    return bfgs_quadratic_search(o, use_gradient_direction ? o->g : o->d, o->f, t0, t_best, f_best);
}

int32_t dzo_bfgs_reset(dzo_bfgs_t o) {
    DZO_REQUIRE(o, DZO_ERR_INVALID, "null optimizer");
    DZO_REQUIRE(o->H, DZO_ERR_STATE, "this handle has no inverse Hessian (gradient-descent optimizer)");
    DZO_TRY(bfgs_identity(o));                                                               // :981 (identity_matrix! :712-720)
    DZO_HIP(hipMemcpyAsync(o->d, o->g, (size_t)o->n * dtype_size(o->dtype), hipMemcpyDeviceToDevice, o->stream));   // :984-986
    return DZO_OK;
}

int32_t dzo_bfgs_set_max_increases(dzo_bfgs_t o, int32_t max_increases) {
    DZO_REQUIRE(o, DZO_ERR_INVALID, "null optimizer");
    o->max_increases = max_increases;
    return DZO_OK;
}

int32_t dzo_bfgs_get_i(dzo_bfgs_t o, int32_t what, int64_t *value) {
    DZO_REQUIRE(o && value, DZO_ERR_INVALID, "null argument");
    switch (what) {
    case 0: *value = o->has_terminated ? 1 : 0; break;
    case 1: *value = o->iteration_count; break;
    case 2: *value = o->n; break;
    case 3: *value = o->last_step_type; break;
    case 4: *value = o->evals; break;
    default: set_error("dzo_bfgs_get_i: unknown field %d", what); return DZO_ERR_INVALID;
    }
    return DZO_OK;
}

int32_t dzo_bfgs_get_s(dzo_bfgs_t o, int32_t what, double *value) {
    DZO_REQUIRE(o && value, DZO_ERR_INVALID, "null argument");
    DZO_REQUIRE(what >= 0 && what <= 2, DZO_ERR_INVALID, "unknown field %d", what);
    *value = what == 0 ? o->f : (what == 1 ? o->last_step_length : o->df);
    return DZO_OK;
}

// Install host-side state (checkpoint / resume, and per-step parity tests that upload the CPU
// reference's state before every step).  The device arrays are written through dzo_bfgs_get_ptr.
int32_t dzo_bfgs_set_s(dzo_bfgs_t o, int32_t what, double value) {
    DZO_REQUIRE(o, DZO_ERR_INVALID, "null optimizer");
    DZO_REQUIRE(what >= 0 && what <= 2, DZO_ERR_INVALID, "unknown field %d", what);
    const double v = round_to_dtype(o->dtype, value);
    if (what == 0) {
        DZO_REQUIRE(!(v != v), DZO_ERR_ASSERT, "@assert !isnan(objective value) (legacy/DZOptimization.jl:773)");
        o->f = v;
    } else if (what == 1) {
        o->last_step_length = v;
    } else {
        o->df = v;
    }
    return DZO_OK;
}

int32_t dzo_bfgs_set_i(dzo_bfgs_t o, int32_t what, int64_t value) {
    DZO_REQUIRE(o, DZO_ERR_INVALID, "null optimizer");
    switch (what) {
    case 0: o->has_terminated = value != 0; break;
    case 1: DZO_REQUIRE(value >= 0, DZO_ERR_INVALID, "negative iteration_count"); o->iteration_count = value; break;
    case 3:
        DZO_REQUIRE(value == DZO_STEP_NULL || value == DZO_STEP_GRADIENT_DESCENT || value == DZO_STEP_BFGS, DZO_ERR_INVALID,
                    "unknown step type %lld", (long long)value);
        o->last_step_type = (int32_t)value;
        break;
    case 4: DZO_REQUIRE(value >= 0, DZO_ERR_INVALID, "negative evaluation count"); o->evals = value; break;
    default: set_error("dzo_bfgs_set_i: field %d is not settable", what); return DZO_ERR_INVALID;
    }
    return DZO_OK;
}

int32_t dzo_bfgs_get_ptr(dzo_bfgs_t o, int32_t what, void **ptr_dev) {
    DZO_REQUIRE(o && ptr_dev, DZO_ERR_INVALID, "null argument");
    if (what == 5) DZO_TRY(bfgs_mirror(o));                  // step! keeps the lower triangle only
    DZO_HIP(hipStreamSynchronize(o->stream));
    switch (what) {
    case 0: *ptr_dev = o->x; break;
    case 1: *ptr_dev = o->dx; break;
    case 2: *ptr_dev = o->g; break;
    case 3: *ptr_dev = o->dg; break;
    case 4: *ptr_dev = o->d; break;
    case 5: *ptr_dev = o->H; break;
    case 6: *ptr_dev = o->scratch; break;
    default: set_error("dzo_bfgs_get_ptr: unknown field %d", what); return DZO_ERR_INVALID;
    }
    return DZO_OK;
}

}  // extern "C"
