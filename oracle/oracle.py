"""ctypes loader for the CPU oracle (oracle/dzo_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (dzoptimization.jl_amd) never imports this module.

PARITY UNPINNED BY THE REFERENCE (no upstream tests / fixtures, no Julia here); see the
header of dzo_oracle.c for what pins it instead.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("DZO_ORACLE_LIB") or os.path.join(_HERE, "libdzo_oracle.so")   # (DZO_ORACLE_LIB: the sanitizer build)

ROSENBROCK2D, ROSENBROCK_CHAIN, QUADRATIC, LSE, QUADRATIC_CHAIN = 0, 1, 2, 3, 4
DOT_SEQUENTIAL, DOT_EIGHT_LANE, DOT_WIDE = 0, 1, 2


def build(force: bool = False) -> str:
    """Compile libdzo_oracle.so with gcc (oracle/Makefile)."""
    src_newer = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in ("dzo_oracle.c", "dzo_oracle_impl.h")
    )
    if force or src_newer:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B" if force else "-s"])
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _declare(_lib)
    return _lib


def _ct(dtype):
    return C.c_double if np.dtype(dtype) == np.float64 else C.c_float


def _suf(dtype):
    return "_f64" if np.dtype(dtype) == np.float64 else "_f32"


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _problem_fields(ct):
    return [("kind", C.c_int32), ("n", C.c_int64), ("A", C.c_void_p), ("c", C.c_void_p), ("lam", ct),
            ("l2", ct), ("bg_on", C.c_int32), ("bg_lo", ct), ("bg_hi", ct),
            ("cons_on", C.c_int32), ("cons_lo", ct), ("cons_hi", ct)]


class _ProblemF64(C.Structure):
    _fields_ = _problem_fields(C.c_double)


class _ProblemF32(C.Structure):
    _fields_ = _problem_fields(C.c_float)


def _declare(L):
    L.orc_set_dot_mode.argtypes = [C.c_int]
    L.orc_set_threads.argtypes = [C.c_int]
    L.orc_pcg_fill_f64.argtypes = [C.c_void_p, C.c_int64, C.c_uint64]
    L.orc_pcg_fill_f32.argtypes = [C.c_void_p, C.c_int64, C.c_uint64]
    L.orc_pcg_raw_u32.argtypes = [C.c_void_p, C.c_int64, C.c_uint64]
    for suf, ct in (("_f64", C.c_double), ("_f32", C.c_float)):
        vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int32

        def f(name, res, args):
            fn = getattr(L, name + suf)
            fn.restype = res
            fn.argtypes = args

        f("orc_dot", ct, [vp, vp, i64])
        f("orc_norm2", ct, [vp, i64])
        f("orc_norm", ct, [vp, i64])
        f("orc_axpy", None, [ct, vp, vp, i64])
        f("orc_axpy_oop", None, [vp, ct, vp, vp, i64])
        f("orc_axpby", None, [ct, vp, ct, vp, i64])
        f("orc_scal", None, [vp, ct, i64])
        f("orc_isequal", C.c_int, [vp, vp, i64])
        f("orc_problem_eval", ct, [vp, vp])
        f("orc_problem_grad", None, [vp, vp, vp])
        f("orc_box_clamp", None, [vp, ct, ct, i64])
        f("orc_lbfgs_direction", None, [vp, vp, vp, vp, vp, vp, i32, i64])
        f("orc_lbfgs_create_problem", vp, [vp, vp, vp, ct, i32])
        f("orc_lbfgs_create_full", vp, [vp, vp, vp, vp, vp, ct, vp, ct, i32, i64])
        f("orc_lbfgs_destroy", None, [vp])
        f("orc_lbfgs_step", None, [vp])
        f("orc_lbfgs_get_i", i64, [vp, C.c_int])
        f("orc_lbfgs_get_s", ct, [vp, C.c_int])
        f("orc_lbfgs_set_f", None, [vp, ct, i32])
        f("orc_lbfgs_get_v", vp, [vp, C.c_int, C.c_int])
        f("orc_lbfgs_set_max_halvings", None, [vp, i64])
        f("orc_lbfgs_set_history", None, [vp, i32, vp, vp, vp, i64])
        f("orc_lbfgs_set_safeguards", None, [vp, i32, i32])
        f("orc_lbfgs_set_line_search", None, [vp, i32, ct, ct, i32])
        f("orc_adgd_create_problem", vp, [vp, vp, vp, ct])
        f("orc_adgd_destroy", None, [vp])
        f("orc_adgd_step", None, [vp])
        f("orc_adgd_get_i", i64, [vp, C.c_int])
        f("orc_adgd_get_s", ct, [vp, C.c_int])
        f("orc_adgd_get_v", vp, [vp, C.c_int])
        f("orc_bfgs_create_problem", vp, [vp, vp, ct])
        f("orc_bfgs_destroy", None, [vp])
        f("orc_bfgs_step", None, [vp])
        f("orc_bfgs_steps", None, [vp, i32])
        f("orc_bfgs_get_i", i64, [vp, C.c_int])
        f("orc_bfgs_get_s", ct, [vp, C.c_int])
        f("orc_bfgs_get_v", vp, [vp, C.c_int])
        f("orc_bfgs_set_max_increases", None, [vp, i32])
        f("orc_bfgs_set_s", None, [vp, C.c_int, ct])
        f("orc_bfgs_set_i", None, [vp, C.c_int, i64])
        f("orc_bfgs_update", None, [vp, ct, vp, vp, vp, i64])
        f("orc_symv", None, [vp, vp, vp, i64])
        f("orc_bfgs_line_search", None, [vp, C.c_int, ct, vp, vp])
        f("orc_line_search_eval", ct,
          [vp, vp, vp, vp, i64, vp, ct, vp, ct, ct, C.c_int, vp, vp, vp, vp])
        f("orc_line_search_eval_problem", ct, [vp, vp, ct, vp, ct, ct, C.c_int, vp, vp, vp, vp])
        f("orc_gd_create_problem", vp, [vp, vp, ct])
        f("orc_gd_destroy", None, [vp])
        f("orc_gd_step", None, [vp])
        f("orc_gd_base", vp, [vp])
        f("orc_gd_delta_f", ct, [vp])


def set_dot_mode(mode: int) -> None:
    lib().orc_set_dot_mode(mode)


def set_threads(t: int) -> None:
    lib().orc_set_threads(t)


def pcg_fill(n: int, seed: int, dtype=np.float64) -> np.ndarray:
    """legacy/PCG.jl:15-22 random_fill!: uniform [0,1) = 2^-32 * u32."""
    x = np.empty(n, dtype=dtype)
    getattr(lib(), "orc_pcg_fill" + _suf(dtype))(_ptr(x), n, seed)
    return x


def pcg_raw(n: int, seed: int) -> np.ndarray:
    x = np.empty(n, dtype=np.uint32)
    lib().orc_pcg_raw_u32(_ptr(x), n, seed)
    return x


def dot(a, b):
    return getattr(lib(), "orc_dot" + _suf(a.dtype))(_ptr(a), _ptr(b), a.size)


def norm(a):
    return getattr(lib(), "orc_norm" + _suf(a.dtype))(_ptr(a), a.size)


def axpy(alpha, x, y):
    getattr(lib(), "orc_axpy" + _suf(x.dtype))(alpha, _ptr(x), _ptr(y), x.size)


def axpby(alpha, x, beta, y):
    getattr(lib(), "orc_axpby" + _suf(x.dtype))(alpha, _ptr(x), beta, _ptr(y), x.size)


def scal(x, alpha):
    getattr(lib(), "orc_scal" + _suf(x.dtype))(_ptr(x), alpha, x.size)


def box_clamp(x, lo, hi):
    """UniformBoxConstraint call (legacy/DZOptimization.jl:264-272), in place."""
    getattr(lib(), "orc_box_clamp" + _suf(x.dtype))(_ptr(x), lo, hi, x.size)
    return x


def isequal(a, b) -> bool:
    return bool(getattr(lib(), "orc_isequal" + _suf(a.dtype))(_ptr(a), _ptr(b), a.size))


class Problem:
    """Synthetic objective; the reference's user callbacks."""

    def __init__(self, kind, n, dtype=np.float64, A=None, c=None, lam=0.0, l2=0.0, box_gradient=None,
                 box_constraint=None):
        """l2: L2RegularizationWrapper/L2GradientWrapper lambda; box_gradient=(lo, hi):
        UniformBoxGradientWrapper; box_constraint=(lo, hi): UniformBoxConstraint as the
        optimizer's constraint_function! (legacy/DZOptimization.jl:219-296)."""
        self.kind, self.n, self.dtype = kind, int(n), np.dtype(dtype)
        self.A = None if A is None else np.asfortranarray(A, dtype=dtype)
        self.c = None if c is None else np.ascontiguousarray(c, dtype=dtype)
        cls = _ProblemF64 if self.dtype == np.float64 else _ProblemF32
        bg = box_gradient or (0.0, 0.0)
        bc = box_constraint or (0.0, 0.0)
        self.struct = cls(kind, self.n,
                          None if self.A is None else self.A.ctypes.data,
                          None if self.c is None else self.c.ctypes.data, lam,
                          l2, int(box_gradient is not None), bg[0], bg[1],
                          int(box_constraint is not None), bc[0], bc[1])
        self.lam = lam

    @property
    def ref(self):
        return C.addressof(self.struct)

    def eval(self, x):
        return getattr(lib(), "orc_problem_eval" + _suf(self.dtype))(self.ref, _ptr(x))

    def grad(self, x):
        g = np.empty_like(x)
        getattr(lib(), "orc_problem_grad" + _suf(self.dtype))(self.ref, _ptr(g), _ptr(x))
        return g


def _view(ptr, n, dtype):
    if not ptr:
        return None
    ct = _ct(dtype)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(n,))


def lbfgs_direction(g, S, Y, rho):
    """compute_lbfgs_step_direction! (src/DZOptimization.jl:430-451) on explicit arrays.

    S, Y: (k, n) arrays, row 0 = newest pair.  Returns (d, alpha)."""
    dtype = g.dtype
    k, n = (0, g.size) if len(S) == 0 else S.shape
    S = np.ascontiguousarray(S, dtype=dtype).reshape(k, n)
    Y = np.ascontiguousarray(Y, dtype=dtype).reshape(k, n)
    rho = np.ascontiguousarray(rho, dtype=dtype)
    d = np.empty_like(g)
    alpha = np.zeros(max(k, 1), dtype=dtype)
    PA = C.c_void_p * max(k, 1)
    sp = PA(*[S[i].ctypes.data for i in range(k)])
    yp = PA(*[Y[i].ctypes.data for i in range(k)])
    getattr(lib(), "orc_lbfgs_direction" + _suf(dtype))(
        _ptr(d), _ptr(g), sp, yp, _ptr(alpha), _ptr(rho), k, n)
    return d, alpha[:k]


class LBFGS:
    """LBFGSOptimizer (src/DZOptimization.jl:321-427) + step! (:454-509) on the CPU."""

    def __init__(self, problem: Problem, x0, initial_step_length, history_length):
        self.problem = problem
        self.dtype = problem.dtype
        self.suf = _suf(self.dtype)
        self.x = np.ascontiguousarray(x0, dtype=self.dtype)  # aliased, :393
        self.g = np.empty_like(self.x)
        self.h = getattr(lib(), "orc_lbfgs_create_problem" + self.suf)(
            problem.ref, _ptr(self.x), _ptr(self.g), initial_step_length, history_length)
        if not self.h:
            raise ValueError("orc_lbfgs_create failed (assertion in reference constructor)")
        self.n = self.x.size

    def close(self):
        if self.h:
            getattr(lib(), "orc_lbfgs_destroy" + self.suf)(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def step(self):
        getattr(lib(), "orc_lbfgs_step" + self.suf)(self.h)
        return self

    def _i(self, w):
        return getattr(lib(), "orc_lbfgs_get_i" + self.suf)(self.h, w)

    def _v(self, w, idx=0, n=None):
        return _view(getattr(lib(), "orc_lbfgs_get_v" + self.suf)(self.h, w, idx),
                     self.n if n is None else n, self.dtype)

    is_stuck = property(lambda s: bool(s._i(0)))
    iteration_count = property(lambda s: s._i(1))
    history_length = property(lambda s: s._i(3))
    history_count = property(lambda s: s._i(4))
    last_trials = property(lambda s: s._i(7))
    current_objective_value = property(
        lambda s: getattr(lib(), "orc_lbfgs_get_s" + s.suf)(s.h, 0))
    delta_objective_value = property(
        lambda s: getattr(lib(), "orc_lbfgs_get_s" + s.suf)(s.h, 1))
    current_point = property(lambda s: s._v(0))
    delta_point = property(lambda s: s._v(1))
    current_gradient = property(lambda s: s._v(2))
    delta_gradient = property(lambda s: s._v(3))
    step_direction = property(lambda s: s._v(4))

    def S(self, i):
        return self._v(5, i)

    def Y(self, i):
        return self._v(6, i)

    @property
    def rho_history(self):
        return self._v(8, n=max(self.history_length, 1))[: self._i(6)].copy()

    @property
    def alpha_history(self):
        return self._v(7, n=max(self.history_length, 1))[: self._i(5)].copy()

    def history_arrays(self):
        k = self.history_count
        S = np.stack([self.S(i) for i in range(k)]) if k else np.zeros((0, self.n), self.dtype)
        Y = np.stack([self.Y(i) for i in range(k)]) if k else np.zeros((0, self.n), self.dtype)
        return S, Y

    def set_history(self, S, Y, rho=None, iteration_count=None):
        S = np.ascontiguousarray(S, dtype=self.dtype)
        Y = np.ascontiguousarray(Y, dtype=self.dtype)
        k = S.shape[0]
        assert k <= self.history_length
        rp = None if rho is None else _ptr(np.ascontiguousarray(rho, dtype=self.dtype))
        getattr(lib(), "orc_lbfgs_set_history" + self.suf)(
            self.h, k, _ptr(S), _ptr(Y), rp, k if iteration_count is None else iteration_count)

    def set_max_halvings(self, v):
        getattr(lib(), "orc_lbfgs_set_max_halvings" + self.suf)(self.h, v)

    def install_state(self, x, g, f, S, Y, rho, iteration_count):
        """Follow another optimizer: its point, gradient, objective value and (s, y) history become this one's."""
        self.current_point[:] = x
        self.current_gradient[:] = g
        getattr(lib(), "orc_lbfgs_set_f" + self.suf)(self.h, f, 0)
        self.set_history(S, Y, rho, iteration_count)

    # optional safeguards (off = the live reference), SURVEY.md 8(f) rows 2 and 4
    def set_safeguards(self, descent_check=False, steepest_descent_fallback=False):
        getattr(lib(), "orc_lbfgs_set_safeguards" + self.suf)(self.h, int(descent_check), int(steepest_descent_fallback))

    def set_line_search(self, kind, c1=0.0, c2=0.0, max_evals=0):
        """kind 0 = take_backtracking_step! (reference), 1 = strong Wolfe on the
        LineSearchEvaluator quotients; zero c1 / c2 / max_evals keep the defaults (1e-4, 0.9, 40)."""
        getattr(lib(), "orc_lbfgs_set_line_search" + self.suf)(self.h, int(kind), c1, c2, int(max_evals))

    last_step_length = property(lambda s: getattr(lib(), "orc_lbfgs_get_s" + s.suf)(s.h, 2))
    history_resets = property(lambda s: s._i(8))
    descent_resets = property(lambda s: s._i(9))
    last_step_kind = property(lambda s: s._i(10))


class AdGD:
    """AdGDOptimizer (src/DZOptimization.jl:179-312)."""

    def __init__(self, problem: Problem, x0, initial_step_length):
        self.problem, self.dtype = problem, problem.dtype
        self.suf = _suf(self.dtype)
        self.x = np.ascontiguousarray(x0, dtype=self.dtype)
        self.g = np.empty_like(self.x)
        self.n = self.x.size
        self.h = getattr(lib(), "orc_adgd_create_problem" + self.suf)(
            problem.ref, _ptr(self.x), _ptr(self.g), initial_step_length)
        if not self.h:
            raise ValueError("orc_adgd_create failed")

    def close(self):
        if self.h:
            getattr(lib(), "orc_adgd_destroy" + self.suf)(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def step(self):
        getattr(lib(), "orc_adgd_step" + self.suf)(self.h)
        return self

    def _i(self, w):
        return getattr(lib(), "orc_adgd_get_i" + self.suf)(self.h, w)

    def _s(self, w):
        return getattr(lib(), "orc_adgd_get_s" + self.suf)(self.h, w)

    def _v(self, w):
        return _view(getattr(lib(), "orc_adgd_get_v" + self.suf)(self.h, w), self.n, self.dtype)

    is_stuck = property(lambda s: bool(s._i(0)))
    iteration_count = property(lambda s: s._i(1))
    current_objective_value = property(lambda s: s._s(0))
    delta_objective_value = property(lambda s: s._s(1))
    current_step_size = property(lambda s: s._s(2))
    previous_step_size = property(lambda s: s._s(3))
    current_point = property(lambda s: s._v(0))
    delta_point = property(lambda s: s._v(1))
    current_gradient = property(lambda s: s._v(2))
    delta_gradient = property(lambda s: s._v(3))


class BFGS:
    """BFGSOptimizer (legacy/DZOptimization.jl:733-994, spec by reading)."""

    def __init__(self, problem: Problem, x0, initial_step_length):
        self.problem, self.dtype = problem, problem.dtype
        self.suf = _suf(self.dtype)
        x0 = np.ascontiguousarray(x0, dtype=self.dtype)
        self.n = x0.size
        self.h = getattr(lib(), "orc_bfgs_create_problem" + self.suf)(
            problem.ref, _ptr(x0), initial_step_length)
        if not self.h:
            raise ValueError("orc_bfgs_create failed")

    def close(self):
        if self.h:
            getattr(lib(), "orc_bfgs_destroy" + self.suf)(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def step(self):
        getattr(lib(), "orc_bfgs_step" + self.suf)(self.h)
        return self

    def steps(self, k):
        """k step! calls inside one C call (releases the interpreter lock for the whole run)."""
        getattr(lib(), "orc_bfgs_steps" + self.suf)(self.h, int(k))
        return self

    def _i(self, w):
        return getattr(lib(), "orc_bfgs_get_i" + self.suf)(self.h, w)

    def _s(self, w):
        return getattr(lib(), "orc_bfgs_get_s" + self.suf)(self.h, w)

    def _v(self, w, n=None):
        return _view(getattr(lib(), "orc_bfgs_get_v" + self.suf)(self.h, w),
                     self.n if n is None else n, self.dtype)

    has_terminated = property(lambda s: bool(s._i(0)))
    has_converged = has_terminated  # README.md:38
    iteration_count = property(lambda s: s._i(1))
    last_step_type = property(lambda s: s._i(3))
    objective_evaluations = property(lambda s: s._i(4))
    current_objective_value = property(lambda s: s._s(0))
    last_step_length = property(lambda s: s._s(1))
    current_point = property(lambda s: s._v(0))
    delta_point = property(lambda s: s._v(1))
    current_gradient = property(lambda s: s._v(2))
    delta_gradient = property(lambda s: s._v(3))
    next_step_direction = property(lambda s: s._v(4))

    @property
    def approximate_inverse_hessian(self):
        return self._v(5, n=self.n * self.n).reshape(self.n, self.n, order="F")

    def set_max_increases(self, v):
        getattr(lib(), "orc_bfgs_set_max_increases" + self.suf)(self.h, int(v))

    def install_state(self, x, g, H, d, f, last_step_length, iteration_count=0, last_step_type=0, dx=None, dg=None):
        """Overwrite the optimizer's state (all fields are public in the reference, legacy :733-751).
        H: (n, n) symmetric or column-major."""
        self.current_point[:] = x
        self.current_gradient[:] = g
        self._v(5, n=self.n * self.n)[:] = np.asarray(H, dtype=self.dtype).reshape(-1, order="F")
        self.next_step_direction[:] = d
        if dx is not None:
            self.delta_point[:] = dx
        if dg is not None:
            self.delta_gradient[:] = dg
        L = lib()
        getattr(L, "orc_bfgs_set_s" + self.suf)(self.h, 0, f)
        getattr(L, "orc_bfgs_set_s" + self.suf)(self.h, 1, last_step_length)
        getattr(L, "orc_bfgs_set_i" + self.suf)(self.h, 0, 0)
        getattr(L, "orc_bfgs_set_i" + self.suf)(self.h, 1, int(iteration_count))
        getattr(L, "orc_bfgs_set_i" + self.suf)(self.h, 3, int(last_step_type))

    def line_search(self, use_gradient_dir: bool, t0: float):
        ct = _ct(self.dtype)
        t, f = ct(), ct()
        getattr(lib(), "orc_bfgs_line_search" + self.suf)(
            self.h, int(use_gradient_dir), t0, C.byref(t), C.byref(f))
        return t.value, f.value


def line_search_eval(problem, x, f_old, direction, overlap, step_size, compute_gradient):
    """LineSearchEvaluator call (src/DZOptimization.jl:65-92).
    Returns (f_new, improvement_ratio, slope_ratio, trial_point, trial_gradient)."""
    ct = _ct(problem.dtype)
    tp, tg = np.empty_like(x), np.empty_like(x)
    ir, sr = ct(), ct()
    f = getattr(lib(), "orc_line_search_eval_problem" + _suf(problem.dtype))(
        problem.ref, _ptr(x), f_old, _ptr(direction), overlap, step_size, int(compute_gradient), _ptr(tp), _ptr(tg),
        C.byref(ir), C.byref(sr))
    return f, ir.value, sr.value, tp, tg


class GradientDescent(BFGS):
    """Legacy GradientDescentOptimizer with QuadraticLineSearch (legacy/DZOptimization.jl:305-449)."""

    def __init__(self, problem: Problem, x0, initial_step_length):
        self.problem, self.dtype = problem, problem.dtype
        self.suf = _suf(self.dtype)
        x0 = np.ascontiguousarray(x0, dtype=self.dtype)
        self.n = x0.size
        self.gd = getattr(lib(), "orc_gd_create_problem" + self.suf)(problem.ref, _ptr(x0), initial_step_length)
        self.h = getattr(lib(), "orc_gd_base" + self.suf)(self.gd)     # BFGS accessors read the shared fields

    def close(self):
        if getattr(self, "gd", None):
            getattr(lib(), "orc_gd_destroy" + self.suf)(self.gd)
            self.gd = self.h = None

    def step(self):
        getattr(lib(), "orc_gd_step" + self.suf)(self.gd)
        return self

    delta_objective_value = property(lambda s: getattr(lib(), "orc_gd_delta_f" + s.suf)(s.gd))


def bfgs_rate_worker(args):
    """(n, first_seed, instances, steps) -> (step! calls done, seconds): dense BFGS on chained Rosenbrock
    instances, single-threaded; run in worker PROCESSES by bench.py's batched CPU baseline (one optimizer
    per core, the reference's "run multiple optimizers in parallel")."""
    import time
    n, first_seed, instances, steps = args
    set_threads(1)
    prob = Problem(ROSENBROCK_CHAIN, n)
    refs = [BFGS(prob, pcg_fill(n, first_seed + b), 1.0) for b in range(instances)]
    t0 = time.perf_counter()
    for r in refs:
        r.steps(steps)
    dt = time.perf_counter() - t0
    return sum(r.iteration_count for r in refs), dt


def bfgs_update(H, step_length, d, dg):
    """update_inverse_hessian! (legacy/DZOptimization.jl:864-889). H (F-order) and d are
    modified in place; returns the scratch vector t = H*dg."""
    assert H.flags.f_contiguous
    t = np.empty_like(d)
    getattr(lib(), "orc_bfgs_update" + _suf(d.dtype))(
        _ptr(H), step_length, _ptr(d), _ptr(dg), _ptr(t), d.size)
    return t


def symv(H, v):
    t = np.empty_like(v)
    getattr(lib(), "orc_symv" + _suf(v.dtype))(_ptr(t), _ptr(H), _ptr(v), v.size)
    return t


# ---------------------------------------------------------------------------------------------
# Synthetic workload definitions shared by tests and bench; inputs only.
# ---------------------------------------------------------------------------------------------

def rosenbrock_chain_x0(n, dtype=np.float64, seed=5):
    """C3 start: -1.2 (odd i, 1-based), 1.0 (even i) plus 0.01*(u-1/2), u = PCG32(seed)."""
    u = pcg_fill(n, seed, np.float64)
    base = np.where(np.arange(n) % 2 == 0, -1.2, 1.0)
    return (base + 0.01 * (u - 0.5)).astype(dtype)


def frozen_two_loop_state(n, k, dtype=np.float64):
    """K1-in-isolation inputs: g, s_i, y_i = u - 1/2 (seeds 10, 100+i, 200+i), y_i += s_i."""
    g = (pcg_fill(n, 10, np.float64) - 0.5).astype(dtype)
    S = np.empty((k, n), dtype=dtype)
    Y = np.empty((k, n), dtype=dtype)
    for i in range(k):
        s = pcg_fill(n, 100 + i, np.float64) - 0.5
        y = pcg_fill(n, 200 + i, np.float64) - 0.5 + s
        S[i] = s.astype(dtype)
        Y[i] = y.astype(dtype)
    return g, S, Y


def quadratic_matrix(n, dtype=np.float64, r=8):
    """C2: A = D + U U'/r, D = diag(1 + 99 u) (seed 2), U n x r entries u - 1/2 (seed 3)."""
    dvec = 1.0 + 99.0 * pcg_fill(n, 2, np.float64)
    U = (pcg_fill(n * r, 3, np.float64) - 0.5).reshape(n, r, order="F")
    A = (U @ U.T) / r
    A[np.diag_indices(n)] += dvec
    A = 0.5 * (A + A.T)
    return np.asfortranarray(A.astype(dtype))
