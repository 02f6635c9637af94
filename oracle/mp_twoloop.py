"""High-precision (mpmath) twin of the oracle's two-loop recursion and dense update.

TEST INFRASTRUCTURE ONLY.  Arbiter for GPU-vs-oracle differences: both must be within
tolerance of THIS evaluation of the same inputs (SURVEY.md 8(c) pin (2)).  Pure-Python loops,
so small cases only (n up to a few thousand).

Follows src/DZOptimization.jl:430-451 (two-loop) and legacy/DZOptimization.jl:864-889
(update_inverse_hessian!) operation for operation, in exact-ish arithmetic.
"""
from __future__ import annotations

import mpmath as mp
import numpy as np

mp.mp.dps = 60


def _vec(a):
    return [mp.mpf(float(v)) for v in a]


def _dot(a, b):
    return mp.fsum(x * y for x, y in zip(a, b))


def two_loop(g, S, Y, rho=None):
    """Returns (d, alpha) as float64 arrays rounded once from 60-digit values.

    rho defaults to the exact s_i.y_i; pass the stored rho to arbitrate a run that used
    rounded rho values (src/DZOptimization.jl:505 stores the rounded dot)."""
    k = len(S)
    q = _vec(g)
    Sm = [_vec(s) for s in S]
    Ym = [_vec(y) for y in Y]
    rh = [_dot(Sm[i], Ym[i]) for i in range(k)] if rho is None else [mp.mpf(float(r)) for r in rho]
    alpha = [mp.mpf(0)] * k
    for i in range(k):                                   # :439-442
        alpha[i] = _dot(Sm[i], q) / rh[i]
        q = [qe - alpha[i] * ye for qe, ye in zip(q, Ym[i])]
    if k > 0:                                            # :443-445
        scale = -rh[0] / _dot(Ym[0], Ym[0])
        q = [scale * qe for qe in q]
    for i in reversed(range(k)):                         # :446-449
        beta = _dot(Ym[i], q) / rh[i]
        c = alpha[i] + beta
        q = [qe - c * se for qe, se in zip(q, Sm[i])]
    return (np.array([float(v) for v in q]), np.array([float(a) for a in alpha]))


def dense_inverse_from_pairs(g, S, Y):
    """-H_k g with H_k built by the textbook inverse update from H0 = gamma*I, oldest pair
    first (identity (ii) of SURVEY.md section 4).  Returns d as float64."""
    k = len(S)
    n = len(g)
    Sm = [_vec(s) for s in S]
    Ym = [_vec(y) for y in Y]
    gamma = _dot(Sm[0], Ym[0]) / _dot(Ym[0], Ym[0]) if k else mp.mpf(1)
    H = mp.eye(n) * gamma
    for i in reversed(range(k)):
        s = mp.matrix(Sm[i])
        y = mp.matrix(Ym[i])
        r = 1 / (s.T * y)[0]
        V = mp.eye(n) - r * (s * y.T)
        H = V * H * V.T + r * (s * s.T)
    d = -(H * mp.matrix(_vec(g)))
    return np.array([float(v) for v in d])


def bfgs_update(H, lam, d, y):
    """H+ per legacy/DZOptimization.jl:873-886 in 60-digit arithmetic; H is n x n array."""
    n = len(d)
    Hm = mp.matrix(H.tolist())
    dv = mp.matrix(_vec(d))
    yv = mp.matrix(_vec(y))
    overlap = (dv.T * yv)[0]
    dp = dv / overlap
    t = Hm * yv
    delta = mp.mpf(float(lam)) * overlap + (yv.T * t)[0]
    Hn = Hm + delta * (dp * dp.T) - (t * dp.T + dp * t.T)
    return np.array([[float(Hn[i, j]) for j in range(n)] for i in range(n)])
