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

// the same for an element that has both neighbours (0 < i < n - 1): identical operations, no index tests
template <typename T> __device__ __forceinline__ T rosen_grad_interior(T xp, T xi, T xn) {
    const T t2 = dfma(-xi, xi, xn);
    // -2 (1 - xi) as ONE fma: 2 xi - 2 is the same real number, and rounding commutes with the factor 2, so
    // round(2 xi - 2) == -2 round(1 - xi) bit for bit (no overflow / subnormals anywhere near xi = 1)
    const T m2 = dfma((T)2, xi, (T)-2);
    const T gi = dfma((T)-400 * xi, t2, m2);
    const T t2p = dfma(-xp, xp, xi);
    return dfma((T)200, t2p, gi);
}

}  // namespace dzo
