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

}  // namespace dzo
