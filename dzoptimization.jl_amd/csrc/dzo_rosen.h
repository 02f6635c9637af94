// dzo_rosen.h -- elementwise pieces of the chained Rosenbrock objective
//   f = sum_{i<n-1} 100 (x[i+1] - x[i]^2)^2 + (1 - x[i])^2        (BASELINE configs[2])
// shared by the stand-alone objective / gradient kernels (dzo_problems.hip) and the L-BFGS
// single-pass step kernel (dzo_lbfgs.hip).  Operation order mirrors oracle/dzo_oracle_impl.h.
#pragma once
#include "dzo_common.h"

namespace dzo {

template <typename T> __device__ __forceinline__ double rosen_term(T xi, T xn) {
    T t1 = (T)1 - xi;
    T t2 = dfma(-xi, xi, xn);
    return (double)dfma((T)100 * t2, t2, t1 * t1);
}

template <typename T> __device__ __forceinline__ T rosen_grad_elem(int64_t i, int64_t n, T xp, T xi, T xn) {
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

// The index tests of rosen_grad_elem as per-element COEFFICIENTS (a kernel forms them once per wave-row and runs the same
// straight-line stencil on every row).  Interior element: c400 = -400, c2a = 2, c2b = -2, c200 = 200 -- the operations of
// rosen_grad_elem, with -2 (1 - xi) as ONE fma: 2 xi - 2 is the same real number, and rounding commutes with the factor
// 2, so round(2 xi - 2) == -2 round(1 - xi) bit for bit (no overflow / subnormals anywhere near xi = 1).  An element
// without a next one gets c400 = c2a = c2b = 0: its first term is fma(+-0, t2, +0) = +0, the (T)0 rosen_grad_elem starts
// from; an element without a previous one gets c200 = 0 and fma(0, t2p, gi) = gi.  Same bits as rosen_grad_elem for
// FINITE neighbours, whatever stands in for a missing one (tests/test_gpu_lbfgs.py compares the pass with the kernel
// that calls rosen_grad_elem, first and last element included).
template <typename T> struct RosenCoef { T c400, c2a, c2b, c200; };
template <typename T> __device__ __forceinline__ RosenCoef<T> rosen_coef(int64_t i, int64_t n) {
    RosenCoef<T> c;
    const bool nx = i + 1 < n, pv = i > 0 && i < n;           // (i >= n: the phantom padding of a ragged last vector -- all zero, gradient +0)
    c.c400 = nx ? (T)-400 : (T)0; c.c2a = nx ? (T)2 : (T)0; c.c2b = nx ? (T)-2 : (T)0; c.c200 = pv ? (T)200 : (T)0;
    return c;
}
template <typename T> __device__ __forceinline__ T rosen_grad_coef(const RosenCoef<T> &c, T xp, T xi, T xn) {
    const T t2 = dfma(-xi, xi, xn);
    const T m2 = dfma(c.c2a, xi, c.c2b);
    const T gi = dfma(c.c400 * xi, t2, m2);
    const T t2p = dfma(-xp, xp, xi);
    return dfma(c.c200, t2p, gi);
}

// ------------------------------------------------------------------------------------------------------------------
// Chained QUADRATIC (DZO_PROBLEM_QUADRATIC_CHAIN; north_star: "synthetic quadratic/Rosenbrock-N problems" -- the dense
// quadratic of config 2 cannot be stored at n = 1e7, this one is its large-n member):
//   f = sum_{i<n-1} 1/2 (x[i+1] - x[i])^2 + sum_{i<n} lambda/2 (x[i] - 1)^2        convex, tridiagonal Hessian,
//   g_i = (x_i - x_{i+1}) [i+1 < n] + (x_i - x_{i-1}) [i > 0] + lambda (x_i - 1)    minimiser x = 1, condition (4 + lambda) / lambda
// Like the chained Rosenbrock it is a radius-1 stencil, so the L-BFGS point pass serves it through the same code
// (ChainObj below).  The index tests are per-element coefficients here too; operation order = oracle/dzo_oracle_impl.h.
template <typename T> struct QChainCoef { T cR, cL, cD, hk, hm; };
template <typename T> __device__ __forceinline__ QChainCoef<T> qchain_coef(int64_t i, int64_t n, T lambda) {
    QChainCoef<T> c;
    const bool real = i >= 0 && i < n, nx = real && i + 1 < n, pv = real && i > 0;
    c.cR = nx ? (T)1 : (T)0; c.cL = pv ? (T)1 : (T)0; c.cD = real ? lambda : (T)0;
    c.hk = nx ? (T)0.5 : (T)0; c.hm = real ? (T)0.5 * lambda : (T)0;
    return c;
}
template <typename T> __device__ __forceinline__ T qchain_grad_coef(const QChainCoef<T> &c, T xp, T xi, T xn) {
    // (the innermost term as fma(cR, ., +0), not a product: an element without the term -- the last one, or the phantom padding of a
    // ragged n, whose neighbour registers may hold anything finite -- then contributes +0 whatever the neighbour's sign, so a
    // phantom's gradient is +0 bit for bit, as ring_compare_kernel expects; same value as the oracle's cR * (x - x_r) otherwise)
    return dfma(c.cD, xi - (T)1, dfma(c.cL, xi - xp, dfma(c.cR, xi - xn, (T)0)));
}
// the terms attributed to element i: its diagonal term and the pair term with its right neighbour
template <typename T> __device__ __forceinline__ double qchain_term(const QChainCoef<T> &c, T xi, T xn) {
    const T d = xi - (T)1;
    const T p = xn - xi;
    return (double)dfma(c.hk * p, p, (c.hm * d) * d);
}

// What the point pass needs of a chained objective (radius-1 stencil): per-element coefficients formed once per
// wave-row, the gradient of an element from its two neighbours, the objective terms attributed to an element and
// whether an element carries terms at all.  OBJ = 0: chained Rosenbrock, 1: chained quadratic.
template <typename T, int OBJ> struct ChainObj;
template <typename T> struct ChainObj<T, 0> {
    using Coef = RosenCoef<T>;
    static __device__ __forceinline__ Coef coef(int64_t i, int64_t n, T) { return rosen_coef<T>(i, n); }
    static __device__ __forceinline__ T grad(const Coef &c, T xp, T xi, T xn) { return rosen_grad_coef<T>(c, xp, xi, xn); }
    static __device__ __forceinline__ bool has_term(int64_t i, int64_t n) { return i + 1 < n; }
    static __device__ __forceinline__ double term(const Coef &, T xi, T xn) { return rosen_term<T>(xi, xn); }
};
template <typename T> struct ChainObj<T, 1> {
    using Coef = QChainCoef<T>;
    static __device__ __forceinline__ Coef coef(int64_t i, int64_t n, T lambda) { return qchain_coef<T>(i, n, lambda); }
    static __device__ __forceinline__ T grad(const Coef &c, T xp, T xi, T xn) { return qchain_grad_coef<T>(c, xp, xi, xn); }
    static __device__ __forceinline__ bool has_term(int64_t i, int64_t n) { return i < n; }
    static __device__ __forceinline__ double term(const Coef &c, T xi, T xn) { return qchain_term<T>(c, xi, xn); }
};

}  // namespace dzo
