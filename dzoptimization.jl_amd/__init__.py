"""dzoptimization.jl_amd -- MI355X-native BFGS / L-BFGS ``step!()`` (host-side mirror).

This package is the Python twin of the Julia host module in ``julia/DZOptimizationAMD.jl``:
both are thin bindings over the C ABI of ``include/dzo.h`` (``libdzo_hip.so``, hand-written
HIP for gfx950).  It mirrors the reference's optimizer interface -- constructor argument
order, public state fields, ``step!`` -> :func:`step_` / ``opt.step()``, and the termination
flag under all three historical names (``is_stuck`` ≡ ``has_terminated`` ≡ ``has_converged``;
SURVEY.md 1.2) -- so that the parity tests read like the reference's own usage
(README.md:33-41, src/DZOptimization.jl:347-427).

The directory name contains a dot, so it cannot be imported by name; use the loader::

    from dzo_loader import dzo        # repo root
    opt = dzo.LBFGSOptimizer(None, problem, None, x0, 1.0, 20)

There is NO CPU fallback: importing works anywhere (so ABI/symbol tests can run without a
GPU), but every compute entry point raises :class:`DzoError` unless ``libdzo_hip.so`` is
built and a HIP device is present.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DZO_LIB_PATH") or os.path.join(_HERE, "libdzo_hip.so")   # (override: A/B of two builds)
INCLUDE_DIR = os.path.join(os.path.dirname(_HERE), "include")

F32, F64 = 0, 1
ROSENBROCK2D, ROSENBROCK_CHAIN, QUADRATIC, LSE, QUADRATIC_CHAIN = 0, 1, 2, 3, 4
TWOLOOP_CHAIN, TWOLOOP_GRAM = 0, 1
LINE_SEARCH_BACKTRACKING, LINE_SEARCH_WOLFE = 0, 1
STEP_NULL, STEP_GRADIENT_DESCENT, STEP_BFGS = 0, 1, 2

_ERR = {1: "invalid argument", 2: "HIP runtime error", 3: "reference @assert", 4: "out of memory",
        5: "unsupported", 6: "bad call sequence"}


class DzoError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"dzo error {code} ({_ERR.get(code, '?')}): {msg}")
        self.code = code


class AssertionFailed(DzoError, AssertionError):
    """A reference ``@assert`` would have fired (Julia raises AssertionError)."""


def build(force: bool = False) -> str:
    """Compile ``libdzo_hip.so`` for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(_HERE, "csrc", f) for f in os.listdir(os.path.join(_HERE, "csrc"))]
    srcs.append(os.path.join(INCLUDE_DIR, "dzo.h"))
    stale = (not os.path.exists(LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        jobs = str(min(8, os.cpu_count() or 1))
        subprocess.check_call(["make", "-C", _HERE, "-j", jobs] + (["-B"] if force else []))
    return LIB_PATH


_lib = None
_inited = False


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch wheels bundle their own libamdhip64.so (same
    SONAME as /opt/rocm's); if libdzo_hip.so bound to the system copy and torch later loaded
    its own, the second runtime finds no device.  Bind to torch's copy whenever torch is
    installed -- without importing torch -- so the load order never matters."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def lib() -> C.CDLL:
    """Load the C-ABI library (no device needed for loading)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DzoError(2, f"{LIB_PATH} is not built; run __graft_entry__.build() "
                              "(there is no CPU fallback)")
        _preload_hip_runtime()
        _lib = C.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


def _check(rc):
    if rc != 0:
        msg = lib().dzo_last_error().decode(errors="replace")
        raise (AssertionFailed if rc == 3 else DzoError)(rc, msg)


def init(device: int = 0) -> None:
    """Select the HIP device.  Fails loudly when there is none."""
    global _inited
    _check(lib().dzo_init(device))
    _inited = True


def _need_init():
    if not _inited:
        init(int(os.environ.get("LOCAL_RANK", "0")) if "DZO_DEVICE" not in os.environ
             else int(os.environ["DZO_DEVICE"]))


def device_info():
    _need_init()
    name = C.create_string_buffer(128)
    cus, hbm = C.c_int32(), C.c_int64()
    _check(lib().dzo_device_info(name, 128, C.byref(cus), C.byref(hbm)))
    return {"name": name.value.decode(), "compute_units": cus.value, "hbm_bytes": hbm.value}


def device_count() -> int:
    """HIP devices visible to the process (no ``init`` needed, no context created)."""
    v = C.c_int32()
    _check(lib().dzo_device_count(C.byref(v)))
    return v.value


def synchronize():
    _check(lib().dzo_synchronize())


def calibrate_read_bandwidth(nbytes, repeats=20):
    """GB/s of a plain streaming read over ``nbytes`` (Infinity-Cache resident up to ~200 MiB, HBM beyond)."""
    _need_init()
    r = C.c_double()
    _check(lib().dzo_calibrate_read_bandwidth(int(nbytes), int(repeats), C.byref(r)))
    return r.value


def selftest_fast_div(seed, pairs, mode):
    """(checked, mismatches, first) of ``dzo_selftest_fast_div``: the recurrence's division-free quotient against ``a / b``."""
    _need_init()
    chk, bad = C.c_int64(), C.c_int64()
    first = (C.c_double * 4)()
    _check(lib().dzo_selftest_fast_div(int(seed), int(pairs), int(mode), C.byref(chk), C.byref(bad), first))
    return chk.value, bad.value, list(first)


def calibrate_read_bandwidth_of(array, repeats=20):
    """GB/s of the same streaming read over the bytes of a DeviceArray, whatever it holds."""
    _need_init()
    r = C.c_double()
    _check(lib().dzo_calibrate_read_bandwidth_of(array.ptr, int(array.size * array.dtype.itemsize), int(repeats), C.byref(r)))
    return r.value


# ------------------------------------------------------------------------------ ABI table
_vp, _i32, _i64, _dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
CONSTRAINT_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p)
OBJECTIVE_FN = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_void_p)
GRADIENT_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p)
_P = C.POINTER

# name -> argtypes; every function returns int32 except dzo_last_error / dzo_version
ABI = {
    "dzo_init": [_i32], "dzo_shutdown": [], "dzo_device_info": [C.c_char_p, _i32, _P(_i32), _P(_i64)],
    "dzo_device_count": [_P(_i32)],
    "dzo_synchronize": [],
    "dzo_profile_enable": [_i32], "dzo_profile_reset": [], "dzo_unsealed_first_reads": [_P(_i64)], "dzo_profile_count": [_P(_i32)],
    "dzo_profile_get": [_i32, C.c_char_p, _i32, _P(_i64), _P(_dbl)],
    "dzo_calibrate_read_bandwidth": [_i64, _i32, _P(_dbl)],
    "dzo_selftest_fast_div": [C.c_uint64, _i64, _i32, _P(_i64), _P(_i64), _P(_dbl)],
    "dzo_calibrate_read_bandwidth_of": [_vp, _i64, _i32, _P(_dbl)],
    "dzo_malloc": [_P(_vp), _i64], "dzo_free": [_vp], "dzo_memcpy_h2d": [_vp, _vp, _i64],
    "dzo_memcpy_d2h": [_vp, _vp, _i64], "dzo_memcpy_d2d": [_vp, _vp, _i64],
    "dzo_axpy": [_i64, _i32, _dbl, _vp, _vp], "dzo_axpby": [_i64, _i32, _dbl, _vp, _dbl, _vp],
    "dzo_scal": [_i64, _i32, _dbl, _vp], "dzo_copy": [_i64, _i32, _vp, _vp],
    "dzo_fill": [_i64, _i32, _dbl, _vp], "dzo_dot": [_i64, _i32, _vp, _vp, _P(_dbl)],
    "dzo_nrm2": [_i64, _i32, _vp, _P(_dbl)], "dzo_isequal": [_i64, _i32, _vp, _vp, _P(_i32)],
    "dzo_trial_point": [_i64, _i32, _vp, _dbl, _vp, _vp],
    "dzo_norm2": [_i64, _i32, _vp, _P(_dbl)], "dzo_inv_norm": [_i64, _i32, _vp, _P(_dbl)],
    "dzo_negate": [_i64, _i32, _vp], "dzo_scal_oop": [_i64, _i32, _vp, _dbl, _vp],
    "dzo_problem_create": [_i32, _i64, _i32, _vp, _vp, _dbl, _P(_vp)], "dzo_problem_destroy": [_vp],
    "dzo_problem_eval": [_vp, _vp, _P(_dbl)], "dzo_problem_grad": [_vp, _vp, _vp],
    "dzo_problem_objective_cb": [_vp, _vp], "dzo_problem_gradient_cb": [_vp, _vp, _vp], "dzo_problem_constraint_cb": [_vp, _vp],
    "dzo_problem_set_l2": [_vp, _dbl], "dzo_problem_set_box_gradient": [_vp, _i32, _dbl, _dbl],
    "dzo_problem_set_box_constraint": [_vp, _i32, _dbl, _dbl], "dzo_box_clamp": [_i64, _i32, _vp, _dbl, _dbl],
    "dzo_lbfgs_create": [_i64, _i32, _i32, _vp, _vp, _dbl, _dbl, _P(_vp)],
    "dzo_lbfgs_create_callbacks": [CONSTRAINT_FN, OBJECTIVE_FN, GRADIENT_FN, _vp, _i64, _i32, _i32, _vp,
                                   _dbl, _P(_vp)],
    "dzo_lbfgs_create_problem": [_vp, _i32, _vp, _dbl, _P(_vp)], "dzo_lbfgs_destroy": [_vp],
    "dzo_lbfgs_set_callbacks": [_vp, CONSTRAINT_FN, OBJECTIVE_FN, GRADIENT_FN, _vp],
    "dzo_lbfgs_set_problem": [_vp, _vp], "dzo_lbfgs_set_two_loop_mode": [_vp, _i32],
    "dzo_lbfgs_set_max_halvings": [_vp, _i64],
    "dzo_lbfgs_set_safeguards": [_vp, _i32, _i32], "dzo_lbfgs_set_line_search": [_vp, _i32, _dbl, _dbl, _i32],
    "dzo_lbfgs_step": [_vp], "dzo_lbfgs_direction": [_vp],
    "dzo_lbfgs_begin_search": [_vp], "dzo_lbfgs_trial": [_vp, _dbl, _P(_i32)],
    "dzo_lbfgs_accept": [_vp, _dbl], "dzo_lbfgs_reject": [_vp], "dzo_lbfgs_pre_gradient": [_vp],
    "dzo_lbfgs_post_gradient": [_vp], "dzo_lbfgs_get_i": [_vp, _i32, _P(_i64)],
    "dzo_lbfgs_get_s": [_vp, _i32, _P(_dbl)], "dzo_lbfgs_set_s": [_vp, _i32, _dbl],
    "dzo_lbfgs_set_stuck": [_vp, _i32], "dzo_lbfgs_get_ptr": [_vp, _i32, _i32, _P(_vp)], "dzo_lbfgs_read": [_vp, _i32, _i32, _vp],
    "dzo_lbfgs_get_rho": [_vp, _P(_dbl), _i32, _P(_i32)],
    "dzo_lbfgs_get_alpha": [_vp, _P(_dbl), _i32, _P(_i32)],
    "dzo_lbfgs_set_history": [_vp, _i32, _vp, _vp, _P(_dbl), _i64], "dzo_lbfgs_stream": [_vp, _P(_vp)],
    "dzo_line_search_eval": [CONSTRAINT_FN, OBJECTIVE_FN, GRADIENT_FN, _vp, _i64, _i32, _vp, _dbl, _vp, _dbl,
                             _dbl, _i32, _vp, _vp, _P(_dbl), _P(_dbl), _P(_dbl)],
    "dzo_adgd_create": [_i64, _i32, _vp, _vp, _dbl, _dbl, _P(_vp)],
    "dzo_adgd_create_problem": [_vp, _vp, _dbl, _P(_vp)], "dzo_adgd_destroy": [_vp],
    "dzo_adgd_set_callbacks": [_vp, CONSTRAINT_FN, OBJECTIVE_FN, GRADIENT_FN, _vp], "dzo_adgd_step": [_vp],
    "dzo_adgd_get_i": [_vp, _i32, _P(_i64)], "dzo_adgd_get_s": [_vp, _i32, _P(_dbl)],
    "dzo_adgd_get_ptr": [_vp, _i32, _P(_vp)], "dzo_adgd_read": [_vp, _i32, _vp],
    "dzo_bfgs_create_callbacks": [OBJECTIVE_FN, GRADIENT_FN, CONSTRAINT_FN, _vp, _i64, _i32, _vp, _dbl,
                                  _P(_vp)],
    "dzo_bfgs_create_problem": [_vp, _vp, _dbl, _P(_vp)], "dzo_bfgs_destroy": [_vp], "dzo_bfgs_step": [_vp],
    "dzo_bfgs_convert_problem": [_vp, _vp, _P(_vp)],
    "dzo_bfgs_convert_callbacks": [_vp, _i32, OBJECTIVE_FN, GRADIENT_FN, CONSTRAINT_FN, _vp, _P(_vp)],
    "dzo_bfgs_update": [_i64, _i32, _vp, _dbl, _vp, _vp, _vp, _vp, _vp],
    "dzo_symv": [_i64, _i32, _vp, _vp, _vp],
    "dzo_bfgs_update_mfma": [_i64, _i32, _vp, _dbl, _vp, _vp, _vp],
    "dzo_bfgs_line_search": [_vp, _i32, _dbl, _P(_dbl), _P(_dbl)], "dzo_bfgs_set_max_increases": [_vp, _i32], "dzo_bfgs_reset": [_vp],
    "dzo_bfgs_get_i": [_vp, _i32, _P(_i64)], "dzo_bfgs_get_s": [_vp, _i32, _P(_dbl)],
    "dzo_bfgs_get_ptr": [_vp, _i32, _P(_vp)],
    "dzo_bfgs_set_s": [_vp, _i32, _dbl], "dzo_bfgs_set_i": [_vp, _i32, _i64],
    "dzo_gd_create_callbacks": [CONSTRAINT_FN, OBJECTIVE_FN, GRADIENT_FN, _vp, _i64, _i32, _vp, _dbl, _P(_vp)],
    "dzo_gd_create_problem": [_vp, _vp, _dbl, _P(_vp)], "dzo_gd_step": [_vp],
    "dzo_bfgs_batch_create": [_i32, _i64, _i64, _i32, _vp, _dbl, _P(_vp)], "dzo_bfgs_batch_destroy": [_vp],
    "dzo_bfgs_batch_step": [_vp, _i32, _P(_i32)], "dzo_bfgs_batch_get_ptr": [_vp, _i32, _P(_vp)],
    "dzo_bfgs_batch_count_active": [_vp, _P(_i64)],
    "dzo_bfgs_batch_create_on": [_i32, _i32, _i64, _i64, _i32, _vp, _dbl, _P(_vp)], "dzo_bfgs_batch_device": [_vp, _P(_i32)],
    "dzo_bfgs_batch_create_problem": [_vp, _i64, _vp, _dbl, _i32, _P(_vp)],
    "dzo_bfgs_batch_create_problem_matrices": [_vp, _i64, _vp, _i64, _vp, _dbl, _i32, _P(_vp)], "dzo_bfgs_batch_set_max_increases": [_vp, _i32],
    "dzo_comm_unique_id": [_vp], "dzo_comm_init_rank": [_vp, _i32, _i32, _P(_vp)],
    "dzo_comm_init_all": [_P(_i32), _i32, _P(_vp)], "dzo_comm_destroy": [_vp],
    "dzo_comm_info": [_vp, _P(_i32), _P(_i32), _P(_i32), _P(_i64)],
    "dzo_flag_allreduce_min": [_vp, _P(_i32), _P(_i32)],
    "dzo_flag_allreduce_min_n": [_vp, _P(_i32), _i32, _P(_i32)],
    "dzo_bfgs_batch_all_done": [_vp, _P(_vp), _i32, _P(_i32)],
}


def _declare(L):
    for name, args in ABI.items():
        fn = getattr(L, name)           # AttributeError here = symbol missing from the library
        fn.argtypes = args
        fn.restype = C.c_int32
    L.dzo_problem_objective_cb.restype = C.c_double
    L.dzo_problem_gradient_cb.restype = None
    L.dzo_last_error.restype = C.c_char_p
    L.dzo_last_error.argtypes = []
    L.dzo_version.restype = C.c_int32
    L.dzo_version.argtypes = []


def _dt(dtype) -> int:
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return F64
    if dtype == np.float32:
        return F32
    raise TypeError(f"unsupported element type {dtype}")


def _np(dt: int):
    return np.float64 if dt == F64 else np.float32


# ------------------------------------------------------------------------------ device arrays
class DeviceArray:
    """Dense device vector/matrix (what ``A <: AbstractArray{T}`` is for the reference).

    Owns HBM allocated through the C ABI, or wraps a raw device pointer (``owner=False``) such
    as a ``torch.Tensor.data_ptr()`` or an optimizer field."""

    def __init__(self, shape, dtype=np.float64, ptr=None, owner=True):
        _need_init()
        self.shape = (int(shape),) if np.isscalar(shape) else tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.size = int(np.prod(self.shape)) if self.shape else 1
        self.nbytes = self.size * self.dtype.itemsize
        self._owner = owner and ptr is None
        if ptr is None:
            p = C.c_void_p()
            _check(lib().dzo_malloc(C.byref(p), max(self.nbytes, 16)))
            ptr = p.value
        self.ptr = int(ptr)

    @classmethod
    def from_host(cls, a, dtype=None):
        a = np.ascontiguousarray(a, dtype=dtype)
        d = cls(a.shape, a.dtype)
        d.upload(a)
        return d

    @classmethod
    def zeros(cls, shape, dtype=np.float64):
        d = cls(shape, dtype)
        _check(lib().dzo_fill(d.size, _dt(dtype), 0.0, d.ptr))
        return d

    def upload(self, a):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        assert a.size == self.size
        _check(lib().dzo_memcpy_h2d(self.ptr, a.ctypes.data, self.nbytes))
        return self

    def to_host(self):
        out = np.empty(self.shape, dtype=self.dtype)
        _check(lib().dzo_memcpy_d2h(out.ctypes.data, self.ptr, self.nbytes))
        return out

    def copy(self):
        d = DeviceArray(self.shape, self.dtype)
        _check(lib().dzo_memcpy_d2d(d.ptr, self.ptr, self.nbytes))
        return d

    def view(self, offset_elems, shape):
        return DeviceArray(shape, self.dtype, ptr=self.ptr + offset_elems * self.dtype.itemsize, owner=False)

    def free(self):
        if self._owner and self.ptr and _lib is not None:
            lib().dzo_free(self.ptr)
        self.ptr = 0
        self._owner = False

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def __len__(self):
        return self.shape[0]


def _as_dev(a, dtype=None):
    return a if isinstance(a, DeviceArray) else DeviceArray.from_host(a, dtype)


# L1 primitives with the reference's call shapes (LinearAlgebra / Base)
def axpy_(alpha, x, y):
    _check(lib().dzo_axpy(x.size, _dt(x.dtype), alpha, x.ptr, y.ptr)); return y


def axpby_(alpha, x, beta, y):
    _check(lib().dzo_axpby(x.size, _dt(x.dtype), alpha, x.ptr, beta, y.ptr)); return y


def rmul_(x, alpha):
    _check(lib().dzo_scal(x.size, _dt(x.dtype), alpha, x.ptr)); return x


def copy_(dst, src):
    _check(lib().dzo_copy(src.size, _dt(src.dtype), src.ptr, dst.ptr)); return dst


def fill_(x, value):
    _check(lib().dzo_fill(x.size, _dt(x.dtype), value, x.ptr)); return x


def dot(x, y):
    r = C.c_double()
    _check(lib().dzo_dot(x.size, _dt(x.dtype), x.ptr, y.ptr, C.byref(r))); return r.value


def norm(x):
    r = C.c_double()
    _check(lib().dzo_nrm2(x.size, _dt(x.dtype), x.ptr, C.byref(r))); return r.value


def isequal(a, b):
    r = C.c_int32()
    _check(lib().dzo_isequal(a.size, _dt(a.dtype), a.ptr, b.ptr, C.byref(r))); return bool(r.value)


# legacy/Kernels.jl primitives without a LinearAlgebra twin
def norm2(x):
    """``norm2(x)``: the sum of squares, not its root (legacy/Kernels.jl:49-55,139)."""
    r = C.c_double()
    _check(lib().dzo_norm2(x.size, _dt(x.dtype), x.ptr, C.byref(r))); return r.value


def inv_norm(x):
    """``inv_norm(x) = rsqrt(norm2(x))`` (legacy/Kernels.jl:141)."""
    r = C.c_double()
    _check(lib().dzo_inv_norm(x.size, _dt(x.dtype), x.ptr, C.byref(r))); return r.value


def negate_(x):
    """``negate!(x)`` (legacy/Kernels.jl:76-83)."""
    _check(lib().dzo_negate(x.size, _dt(x.dtype), x.ptr)); return x


def scale_(dst, alpha, x=None):
    """``scale!(x, alpha)`` in place (legacy/Kernels.jl:87-94) or ``scale!(dst, alpha, x)`` out of
    place (:96-104)."""
    if x is None:
        return rmul_(dst, alpha)
    _check(lib().dzo_scal_oop(x.size, _dt(x.dtype), dst.ptr, alpha, x.ptr)); return dst


def box_clamp_(x, lower_bound, upper_bound):
    """``UniformBoxConstraint(lo, hi)(x)`` (legacy/DZOptimization.jl:264-272); returns True."""
    _check(lib().dzo_box_clamp(x.size, _dt(x.dtype), x.ptr, lower_bound, upper_bound)); return True


def trial_point_(dst, t, d, x):
    _check(lib().dzo_trial_point(x.size, _dt(x.dtype), dst.ptr, t, d.ptr, x.ptr)); return dst


# ------------------------------------------------------------------------------ profiling
def profile_enable(level=2):
    """0/False off, 1 = the roofline kernels only (cheap), 2/True = every kernel."""
    _check(lib().dzo_profile_enable(2 if level is True else int(level)))


def profile_reset():
    _check(lib().dzo_profile_reset())


def unsealed_first_reads() -> int:
    """Diagnostic (include/dzo.h): how often a host wait found a kernel's published result not yet matching its seal."""
    v = C.c_int64(0)
    _check(lib().dzo_unsealed_first_reads(C.byref(v)))
    return int(v.value)


def profile_table():
    """{kernel name: (launches, total_ms)} measured with HIP events on the launching stream."""
    n = C.c_int32()
    _check(lib().dzo_profile_count(C.byref(n)))
    out = {}
    for i in range(n.value):
        name = C.create_string_buffer(96)
        launches, ms = C.c_int64(), C.c_double()
        _check(lib().dzo_profile_get(i, name, 96, C.byref(launches), C.byref(ms)))
        if launches.value:
            out[name.value.decode()] = (launches.value, ms.value)
    return out


# ------------------------------------------------------------------------------ problems
class Problem:
    """Built-in device objective; callable like the reference's callbacks:
    ``p(x)`` = objective_function(x), ``p.gradient_(g, x)`` = gradient_function!(g, x)."""

    def __init__(self, kind, n, dtype=np.float64, A=None, c=None, lam=0.0, l2=0.0, box_gradient=None,
                 box_constraint=None):
        """``l2``: L2RegularizationWrapper + L2GradientWrapper lambda; ``box_gradient=(lo, hi)``:
        UniformBoxGradientWrapper; ``box_constraint=(lo, hi)``: UniformBoxConstraint as the
        constraint_function! of optimizers built from this problem (legacy/DZOptimization.jl:219-296)."""
        _need_init()
        self.kind, self.n, self.dtype = kind, int(n), np.dtype(dtype)
        # device layout is column-major (legacy/DZOptimization.jl:746): C-order of A' == F-order of A
        self.A = None if A is None else (A if isinstance(A, DeviceArray) else _as_dev(np.ascontiguousarray(np.asarray(A).T), dtype))
        self.c = None if c is None else _as_dev(c, dtype)
        h = C.c_void_p()
        _check(lib().dzo_problem_create(kind, self.n, _dt(dtype), self.A.ptr if self.A else None,
                                        self.c.ptr if self.c else None, lam, C.byref(h)))
        self.h = h
        if l2:
            _check(lib().dzo_problem_set_l2(h, l2))
        if box_gradient is not None:
            _check(lib().dzo_problem_set_box_gradient(h, 1, box_gradient[0], box_gradient[1]))
        if box_constraint is not None:
            _check(lib().dzo_problem_set_box_constraint(h, 1, box_constraint[0], box_constraint[1]))

    def __call__(self, x):
        f = C.c_double()
        _check(lib().dzo_problem_eval(self.h, x.ptr, C.byref(f)))
        return f.value

    def gradient_(self, g, x):
        _check(lib().dzo_problem_grad(self.h, g.ptr, x.ptr))
        return g

    def native_callbacks(self, with_constraint=False):
        """This objective in the shape of the reference's three callbacks (src/DZOptimization.jl:323-325): C function
        pointers exported by the library (``dzo_problem_*_cb``, ctx = the problem handle).  Pass the result as
        ``objective_function`` of an optimizer (the other two callback arguments are then ignored): it runs its
        general, callback-driven path with no Python frame in the loop -- what a Julia host's closures over
        ``dzo_problem_eval`` / ``dzo_problem_grad`` amount to."""
        return NativeCallbacks(self, with_constraint)

    def __del__(self):
        try:
            if self.h and _lib is not None:
                lib().dzo_problem_destroy(self.h)
                self.h = None
        except Exception:
            pass


class NativeCallbacks:
    """See :meth:`Problem.native_callbacks`."""

    def __init__(self, problem, with_constraint=False):
        self.problem = problem
        L = lib()
        self.cf = C.cast(L.dzo_problem_constraint_cb, CONSTRAINT_FN) if with_constraint else C.cast(None, CONSTRAINT_FN)
        self.of = C.cast(L.dzo_problem_objective_cb, OBJECTIVE_FN)
        self.gf = C.cast(L.dzo_problem_gradient_cb, GRADIENT_FN)
        self.ctx = problem.h


def _wrap_callbacks(constraint, objective, gradient, n, dtype):
    """Python callables taking DeviceArray views -> C function pointers."""
    def mk(ptr):
        return DeviceArray(n, dtype, ptr=ptr, owner=False)
    cf = CONSTRAINT_FN(lambda ctx, x: int(bool(constraint(mk(x))))) if constraint else C.cast(None, CONSTRAINT_FN)
    of = OBJECTIVE_FN(lambda ctx, x: float(objective(mk(x))))

    def _g(ctx, g, x):
        gradient(mk(g), mk(x))
    gf = GRADIENT_FN(_g)
    return cf, of, gf


class LineSearchEvaluator:
    """``LineSearchEvaluator(constraint_function!, objective_function, gradient_function!,
    initial_point, initial_objective_value, initial_gradient, step_direction, overlap)``
    (src/DZOptimization.jl:29-62); calling it with ``(step_size, compute_gradient)`` evaluates the
    trial point ``x + t*d``, its objective, the Armijo quotient ``improvement_ratio`` (:84) and,
    optionally, the trial gradient and the curvature quotient ``slope_ratio`` (:85-90)."""

    def __init__(self, constraint_function_, objective_function, gradient_function_, initial_point,
                 initial_objective_value, initial_gradient, step_direction, overlap):
        _need_init()
        if isinstance(objective_function, Problem):
            p = objective_function
            objective_function, gradient_function_ = p, p.gradient_
        self.current_point, self.current_gradient, self.step_direction = initial_point, initial_gradient, step_direction
        self.n, self.dtype = initial_point.size, initial_point.dtype
        assert initial_gradient.size == self.n and step_direction.size == self.n        # :40-42
        self.current_objective_value, self.overlap = float(initial_objective_value), float(overlap)
        self.trial_point = DeviceArray(self.n, self.dtype)                              # :48
        self.trial_gradient = DeviceArray(self.n, self.dtype)                           # :52
        self.trial_objective_value = self.improvement_ratio = self.slope_ratio = float("nan")
        self._cbs = _wrap_callbacks(constraint_function_, objective_function,
                                    gradient_function_ if gradient_function_ is not None else (lambda g, x: None),
                                    self.n, self.dtype)
        self._has_gradient = gradient_function_ is not None

    def __call__(self, step_size, compute_gradient):
        f, ir, sr = C.c_double(), C.c_double(), C.c_double(self.slope_ratio)
        if compute_gradient and not self._has_gradient:
            raise AssertionFailed(3, "@assert !isnothing(lse.gradient_function!) (src/DZOptimization.jl:86)")
        _check(lib().dzo_line_search_eval(self._cbs[0], self._cbs[1], self._cbs[2], None, self.n, _dt(self.dtype),
                                          self.current_point.ptr, self.current_objective_value, self.step_direction.ptr,
                                          self.overlap, step_size, int(bool(compute_gradient)), self.trial_point.ptr,
                                          self.trial_gradient.ptr, C.byref(f), C.byref(ir), C.byref(sr)))
        self.trial_objective_value, self.improvement_ratio = f.value, ir.value
        if compute_gradient or f.value >= 1e38:
            self.slope_ratio = sr.value
        return f.value


class _FieldView(DeviceArray):
    """A vector field of a live optimizer (``opt.current_point`` ...).  ``to_host()`` goes through ``dzo_<opt>_read``: a copy to
    the host without a pointer hand-out, so the step behind a monitoring read does not have to check the aliased arrays for
    host writes.  Everything that needs the device pointer (``ptr``, ``upload``, ``copy``, passing the field to a kernel) asks
    ``dzo_<opt>_get_ptr`` at that moment -- the hand-out after which the host may have written (src/DZOptimization.jl:393)."""

    def __init__(self, opt, what, idx):
        self._opt, self._what, self._idx = opt, what, idx
        self.shape = (int(opt.n),)
        self.dtype = np.dtype(opt.dtype)
        self.size = int(opt.n)
        self.nbytes = self.size * self.dtype.itemsize
        self._owner = False

    @property
    def ptr(self):
        return int(self._opt._get_ptr(self._what, self._idx))

    def free(self):
        pass

    def to_host(self):
        out = np.empty(self.shape, dtype=self.dtype)
        o = self._opt
        fn = getattr(lib(), f"dzo_{o._prefix}_read")
        _check(fn(o.h, self._what, self._idx, out.ctypes.data) if o._ptr_idx else fn(o.h, self._what, out.ctypes.data))
        return out


class _OptBase:
    _prefix = ""
    _ptr_idx = True

    def _i(self, what):
        v = C.c_int64()
        _check(getattr(lib(), f"dzo_{self._prefix}_get_i")(self.h, what, C.byref(v)))
        return v.value

    def _s(self, what):
        v = C.c_double()
        _check(getattr(lib(), f"dzo_{self._prefix}_get_s")(self.h, what, C.byref(v)))
        return v.value

    _has_read = False                # the C ABI has dzo_<prefix>_read (a copy to the host without a pointer hand-out)

    def _get_ptr(self, what, idx=0):
        p = C.c_void_p()
        fn = getattr(lib(), f"dzo_{self._prefix}_get_ptr")
        _check(fn(self.h, what, idx, C.byref(p)) if self._ptr_idx else fn(self.h, what, C.byref(p)))
        return p.value

    def _p(self, what, idx=0, shape=None):
        if self._has_read and shape is None:
            return _FieldView(self, what, idx)
        return DeviceArray(self.n if shape is None else shape, self.dtype, ptr=self._get_ptr(what, idx), owner=False)

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            getattr(lib(), f"dzo_{self._prefix}_destroy")(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    iteration_count = property(lambda s: s._i(1))
    current_objective_value = property(lambda s: s._s(0))
    current_point = property(lambda s: s._p(0))
    delta_point = property(lambda s: s._p(1))
    current_gradient = property(lambda s: s._p(2))
    delta_gradient = property(lambda s: s._p(3))


class LBFGSOptimizer(_OptBase):
    """``LBFGSOptimizer(constraint_function!, objective_function, gradient_function!,
    initial_point, initial_step_length, history_length)`` (src/DZOptimization.jl:400-407), or
    the full form with ``initial_objective_value`` and ``initial_gradient`` inserted
    (:347-356).  ``objective_function`` may be a :class:`Problem` (then ``gradient_function!``
    is ignored and may be ``None``).  The optimizer ALIASES ``initial_point`` (:393)."""

    _prefix = "lbfgs"
    _has_read = True

    def __init__(self, constraint_function_, objective_function, gradient_function_, initial_point, *rest):
        _need_init()
        if len(rest) == 2:
            f0, g0 = None, None
            initial_step_length, history_length = rest
        elif len(rest) == 4:
            f0, g0, initial_step_length, history_length = rest
        else:
            raise TypeError("LBFGSOptimizer(c, f, g!, x0, [f0, g0,] step, m)")
        self.current_point_array = _as_dev(initial_point)
        x = self.current_point_array
        self.n, self.dtype = x.size, x.dtype
        self.history_length = int(history_length)
        self._keep = [constraint_function_, objective_function, gradient_function_, g0]
        h = C.c_void_p()
        if isinstance(objective_function, Problem) and constraint_function_ is None and g0 is None:
            self.problem = objective_function
            _check(lib().dzo_lbfgs_create_problem(self.problem.h, self.history_length, x.ptr,
                                                  initial_step_length, C.byref(h)))
        elif isinstance(objective_function, NativeCallbacks):
            nc = objective_function
            assert g0 is None, "native callbacks: use the short constructor (:400-407)"
            _check(lib().dzo_lbfgs_create_callbacks(nc.cf, nc.of, nc.gf, nc.ctx, self.n, self.history_length,
                                                    _dt(self.dtype), x.ptr, initial_step_length, C.byref(h)))
        else:
            if isinstance(objective_function, Problem):
                p = objective_function
                objective_function, gradient_function_ = p, p.gradient_
            cbs = _wrap_callbacks(constraint_function_, objective_function, gradient_function_, self.n, self.dtype)
            self._keep.append(cbs)
            if g0 is None:
                _check(lib().dzo_lbfgs_create_callbacks(cbs[0], cbs[1], cbs[2], None, self.n, self.history_length,
                                                        _dt(self.dtype), x.ptr, initial_step_length, C.byref(h)))
            else:
                _check(lib().dzo_lbfgs_create(self.n, self.history_length, _dt(self.dtype), x.ptr, g0.ptr, f0,
                                              initial_step_length, C.byref(h)))
                _check(lib().dzo_lbfgs_set_callbacks(h, cbs[0], cbs[1], cbs[2], None))
        self.h = h

    def step(self):
        """``step!(opt)`` (src/DZOptimization.jl:454-509)."""
        _check(lib().dzo_lbfgs_step(self.h))
        return self

    is_stuck = property(lambda s: bool(s._i(0)))
    has_terminated = is_stuck       # legacy/DZOptimization.jl:478
    has_converged = is_stuck        # README.md:38
    history_count = property(lambda s: s._i(4))
    last_trials = property(lambda s: s._i(5))
    delta_objective_value = property(lambda s: s._s(1))
    step_direction = property(lambda s: s._p(4))

    @property
    def delta_point_history(self):
        return [self._p(5, i) for i in range(self.history_count)]

    @property
    def delta_gradient_history(self):
        return [self._p(6, i) for i in range(self.history_count)]

    def _hist(self, fn):
        buf = (C.c_double * 64)()
        cnt = C.c_int32()
        _check(fn(self.h, buf, 64, C.byref(cnt)))
        return np.array(buf[: cnt.value])

    rho_history = property(lambda s: s._hist(lib().dzo_lbfgs_get_rho))
    alpha_history = property(lambda s: s._hist(lib().dzo_lbfgs_get_alpha))

    # split entry points (include/dzo.h)
    def set_two_loop_mode(self, mode):
        _check(lib().dzo_lbfgs_set_two_loop_mode(self.h, mode))

    def set_max_halvings(self, v):
        _check(lib().dzo_lbfgs_set_max_halvings(self.h, v))

    # optional safeguards (off = the live reference); include/dzo.h, SURVEY.md 8(f) rows 2 and 4
    def set_safeguards(self, descent_check=False, steepest_descent_fallback=False):
        """Legacy descent check (legacy/DZOptimization.jl:682-692) and steepest-descent
        fallback with history reset (:588-610)."""
        _check(lib().dzo_lbfgs_set_safeguards(self.h, int(descent_check), int(steepest_descent_fallback)))

    def set_line_search(self, kind, c1=0.0, c2=0.0, max_evals=0):
        """LINE_SEARCH_BACKTRACKING (reference) or LINE_SEARCH_WOLFE (strong Wolfe on the
        LineSearchEvaluator quotients, src/DZOptimization.jl:65-92)."""
        _check(lib().dzo_lbfgs_set_line_search(self.h, int(kind), c1, c2, int(max_evals)))

    last_step_length = property(lambda s: s._s(2))
    history_resets = property(lambda s: s._i(8))
    descent_resets = property(lambda s: s._i(9))
    last_step_kind = property(lambda s: s._i(10))
    single_pass_steps = property(lambda s: s._i(11))          # informational
    single_pass_rejections = property(lambda s: s._i(12))
    single_pass_retries = property(lambda s: s._i(13))    # rejected first trials continued by a second pass at t/2
    ring_layout = property(lambda s: s._i(14))            # 0 slabs, 1 tile-major pairs, 2 tile-major points (the point ring)
    tile_arrangement = property(lambda s: s._i(15))       # 0 no tiles (slabs), 1 tile-major, 2 stream-major
    pass_recomputes_gradients = property(lambda s: bool(s._i(16)))   # point pass: gradients of the ring's points recomputed from the point tiles
    pass_register_sets = property(lambda s: s._i(17))     # point pass: register sets per wave (1 = two waves per SIMD)
    host_write_checks = property(lambda s: s._i(18))      # point ring: steps that first compared the aliased arrays with the ring

    def compute_step_direction(self, sync=True):
        """``compute_lbfgs_step_direction!`` (:430-451).  ``sync=False`` only enqueues the kernels (the
        C entry point is asynchronous, include/dzo.h) and returns None."""
        _check(lib().dzo_lbfgs_direction(self.h))
        return self.step_direction if sync else None

    def set_history(self, S, Y, rho=None, iteration_count=None):
        S, Y = _as_dev(S, self.dtype), _as_dev(Y, self.dtype)
        k = S.shape[0] if len(S.shape) == 2 else 0
        rp = None if rho is None else (C.c_double * k)(*[float(r) for r in rho])
        _check(lib().dzo_lbfgs_set_history(self.h, k, S.ptr, Y.ptr, rp, k if iteration_count is None else iteration_count))

    def begin_search(self):
        _check(lib().dzo_lbfgs_begin_search(self.h))

    def trial(self, t):
        ch = C.c_int32()
        _check(lib().dzo_lbfgs_trial(self.h, t, C.byref(ch)))
        return bool(ch.value)

    def accept(self, f_new):
        _check(lib().dzo_lbfgs_accept(self.h, f_new))

    def reject(self):
        _check(lib().dzo_lbfgs_reject(self.h))

    def pre_gradient(self):
        _check(lib().dzo_lbfgs_pre_gradient(self.h))

    def post_gradient(self):
        _check(lib().dzo_lbfgs_post_gradient(self.h))

    def set_objective_value(self, f):
        _check(lib().dzo_lbfgs_set_s(self.h, 0, f))

    def set_last_step_length(self, v):
        _check(lib().dzo_lbfgs_set_s(self.h, 2, v))


class AdGDOptimizer(_OptBase):
    """``AdGDOptimizer(constraint!, objective, gradient!, x0, initial_step_length)``
    (src/DZOptimization.jl:245-251)."""

    _prefix = "adgd"
    _ptr_idx = False
    _has_read = True

    def __init__(self, constraint_function_, objective_function, gradient_function_, initial_point,
                 initial_step_length):
        _need_init()
        x = self.current_point_array = _as_dev(initial_point)
        self.n, self.dtype = x.size, x.dtype
        h = C.c_void_p()
        self._keep = [constraint_function_, objective_function, gradient_function_]
        if isinstance(objective_function, Problem) and constraint_function_ is None:
            _check(lib().dzo_adgd_create_problem(objective_function.h, x.ptr, initial_step_length, C.byref(h)))
        else:
            if isinstance(objective_function, Problem):
                p = objective_function
                objective_function, gradient_function_ = p, p.gradient_
            if constraint_function_ is not None and not constraint_function_(x):
                raise AssertionFailed(3, "@assert constraint_function!(initial_point) (src/DZOptimization.jl:256-258)")
            f0 = float(objective_function(x))
            g0 = DeviceArray(self.n, self.dtype)
            gradient_function_(g0, x)
            cbs = _wrap_callbacks(constraint_function_, objective_function, gradient_function_, self.n, self.dtype)
            self._keep += [cbs, g0]
            _check(lib().dzo_adgd_create(self.n, _dt(self.dtype), x.ptr, g0.ptr, f0, initial_step_length, C.byref(h)))
            _check(lib().dzo_adgd_set_callbacks(h, cbs[0], cbs[1], cbs[2], None))
        self.h = h

    def step(self):
        _check(lib().dzo_adgd_step(self.h))
        return self

    is_stuck = property(lambda s: bool(s._i(0)))
    delta_objective_value = property(lambda s: s._s(1))
    current_step_size = property(lambda s: s._s(2))
    previous_step_size = property(lambda s: s._s(3))
    fused_steps = property(lambda s: s._i(3))
    fused_rejections = property(lambda s: s._i(4))
    pipelined_passes = property(lambda s: s._i(5))     # passes adopted that were enqueued before the previous decision was seen
    pipeline_discards = property(lambda s: s._i(6))    # passes in flight dropped (a pointer was handed out, an option changed)
    pipeline_corrections = property(lambda s: s._i(7)) # adopted passes whose device-side step size differed from the host's evaluation
    host_gradient_steps = property(lambda s: s._i(8))  # steps on the generic kernels because the host had written current_gradient


class BFGSOptimizer(_OptBase):
    """``BFGSOptimizer(objective_function, gradient_function!, [constraint_function!,]
    initial_point, initial_step_length)`` (README.md:33-36; legacy/DZOptimization.jl:753-766).
    COPIES ``initial_point`` (:769)."""

    _prefix = "bfgs"
    _ptr_idx = False

    def __init__(self, objective_function, gradient_function_, *rest):
        _need_init()
        if len(rest) == 2:
            constraint_function_, (initial_point, initial_step_length) = None, rest
        elif len(rest) == 3:
            constraint_function_, initial_point, initial_step_length = rest
        else:
            raise TypeError("BFGSOptimizer(f, g!, [c!,] x0, step)")
        x0 = _as_dev(initial_point)
        self.n, self.dtype = x0.size, x0.dtype
        self._keep = [objective_function, gradient_function_, constraint_function_, x0]
        h = C.c_void_p()
        if isinstance(objective_function, Problem) and constraint_function_ is None:
            _check(lib().dzo_bfgs_create_problem(objective_function.h, x0.ptr, initial_step_length, C.byref(h)))
        else:
            if isinstance(objective_function, Problem):
                p = objective_function
                objective_function, gradient_function_ = p, p.gradient_
            cbs = _wrap_callbacks(constraint_function_, objective_function, gradient_function_, self.n, self.dtype)
            self._keep.append(cbs)
            _check(lib().dzo_bfgs_create_callbacks(cbs[1], cbs[2], cbs[0], None, self.n, _dt(self.dtype), x0.ptr,
                                                   initial_step_length, C.byref(h)))
        self.h = h

    @classmethod
    def convert(cls, dtype, opt, objective_function=None, gradient_function_=None, constraint_function_=None):
        """``BFGSOptimizer(::Type{T}, opt)`` / ``BFGSOptimizer(T, f, g!, c!, opt)``
        (legacy/DZOptimization.jl:812-862): continue ``opt`` in element type ``dtype``.
        ``objective_function`` must be given as a :class:`Problem` of that dtype or as callables."""
        _need_init()
        new = cls.__new__(cls)
        new.n, new.dtype = opt.n, np.dtype(dtype)
        new._keep = [objective_function, gradient_function_, constraint_function_]
        h = C.c_void_p()
        if isinstance(objective_function, Problem):
            assert objective_function.dtype == new.dtype
            _check(lib().dzo_bfgs_convert_problem(opt.h, objective_function.h, C.byref(h)))
        else:
            cbs = _wrap_callbacks(constraint_function_, objective_function, gradient_function_, new.n, new.dtype)
            new._keep.append(cbs)
            _check(lib().dzo_bfgs_convert_callbacks(opt.h, _dt(new.dtype), cbs[1], cbs[2], cbs[0], None, C.byref(h)))
        new.h = h
        return new

    def step(self):
        """``step!(opt)`` (legacy/DZOptimization.jl:891-994)."""
        _check(lib().dzo_bfgs_step(self.h))
        return self

    has_terminated = property(lambda s: bool(s._i(0)))   # legacy :738
    has_converged = has_terminated                       # README.md:38
    is_stuck = has_terminated
    last_step_type = property(lambda s: s._i(3))
    objective_evaluations = property(lambda s: s._i(4))
    last_step_length = property(lambda s: s._s(1))
    next_step_direction = property(lambda s: s._p(4))

    @property
    def approximate_inverse_hessian(self):
        """n x n; ``to_host()`` gives the row-major view of the column-major (symmetric) H."""
        return self._p(5, shape=(self.n, self.n))

    def line_search(self, use_gradient_direction, t0):
        t, f = C.c_double(), C.c_double()
        _check(lib().dzo_bfgs_line_search(self.h, int(use_gradient_direction), t0, C.byref(t), C.byref(f)))
        return t.value, f.value

    def set_max_increases(self, v):
        _check(lib().dzo_bfgs_set_max_increases(self.h, v))

    def reset_inverse_hessian(self):
        """H <- I, next_step_direction <- gradient (the reset of legacy/DZOptimization.jl:981-986)."""
        _check(lib().dzo_bfgs_reset(self.h))
        return self

    def install_state(self, x, g, H, d, f, last_step_length, iteration_count=0, last_step_type=STEP_NULL,
                      dx=None, dg=None):
        """Overwrite the whole optimizer state (every field of the reference's struct is public,
        legacy/DZOptimization.jl:733-751; README.md:11 "save/load data in the middle of optimization").
        ``H``: (n, n) symmetric or column-major host array."""
        dt = self.dtype
        self.current_point.upload(np.asarray(x, dt))
        self.current_gradient.upload(np.asarray(g, dt))
        self.approximate_inverse_hessian.upload(np.ascontiguousarray(np.asarray(H, dt).T))   # column-major on the device
        self.next_step_direction.upload(np.asarray(d, dt))
        if dx is not None:
            self.delta_point.upload(np.asarray(dx, dt))
        if dg is not None:
            self.delta_gradient.upload(np.asarray(dg, dt))
        L = lib()
        _check(L.dzo_bfgs_set_s(self.h, 0, float(f)))
        _check(L.dzo_bfgs_set_s(self.h, 1, float(last_step_length)))
        _check(L.dzo_bfgs_set_i(self.h, 0, 0))
        _check(L.dzo_bfgs_set_i(self.h, 1, int(iteration_count)))
        _check(L.dzo_bfgs_set_i(self.h, 3, int(last_step_type)))
        return self


class GradientDescentOptimizer(BFGSOptimizer):
    """``GradientDescentOptimizer([constraint_function!,] objective_function, gradient_function!,
    line_search_function!, initial_point, initial_step_length)`` (legacy/DZOptimization.jl:330-390).
    ``line_search_function!`` must be ``QuadraticLineSearch()`` -- pass ``None`` or the string;
    it is the only search the reference defines (:181-216)."""

    def __init__(self, *args):
        _need_init()
        if len(args) == 5:
            constraint_function_, (objective_function, gradient_function_, _ls, initial_point, initial_step_length) = None, args
        elif len(args) == 6:
            constraint_function_, objective_function, gradient_function_, _ls, initial_point, initial_step_length = args
        else:
            raise TypeError("GradientDescentOptimizer([c!,] f, g!, line_search!, x0, step)")
        x0 = _as_dev(initial_point)
        self.n, self.dtype = x0.size, x0.dtype
        self._keep = [objective_function, gradient_function_, constraint_function_, x0]
        h = C.c_void_p()
        if isinstance(objective_function, Problem) and constraint_function_ is None:
            _check(lib().dzo_gd_create_problem(objective_function.h, x0.ptr, initial_step_length, C.byref(h)))
        else:
            if isinstance(objective_function, Problem):
                p = objective_function
                objective_function, gradient_function_ = p, p.gradient_
            cbs = _wrap_callbacks(constraint_function_, objective_function, gradient_function_, self.n, self.dtype)
            self._keep.append(cbs)
            _check(lib().dzo_gd_create_callbacks(cbs[0], cbs[1], cbs[2], None, self.n, _dt(self.dtype), x0.ptr,
                                                 initial_step_length, C.byref(h)))
        self.h = h

    def step(self):
        """``step!(opt)`` (legacy/DZOptimization.jl:393-449)."""
        _check(lib().dzo_gd_step(self.h))
        return self

    delta_objective_value = property(lambda s: s._s(2))
    approximate_inverse_hessian = property(lambda s: None)


def update_inverse_hessian_(H, step_length, d, dg, scratch, g=None, d_next=None):
    """``update_inverse_hessian!`` (legacy/DZOptimization.jl:864-889) on device arrays, with
    the optional fused next direction ``d_next = H_new * g`` (:958-960)."""
    _check(lib().dzo_bfgs_update(d.size, _dt(d.dtype), H.ptr, step_length, d.ptr, dg.ptr, scratch.ptr,
                                 g.ptr if g is not None else None, d_next.ptr if d_next is not None else None))
    return H


def update_inverse_hessian_mfma_(H, step_length, d, dg, scratch):
    """``update_inverse_hessian!`` with the rank-2 term on MFMA (fp64, n % 16 == 0); see dzo.h."""
    _check(lib().dzo_bfgs_update_mfma(d.size, _dt(d.dtype), H.ptr, step_length, d.ptr, dg.ptr, scratch.ptr))
    return H


def symv_(out, H, v):
    _check(lib().dzo_symv(v.size, _dt(v.dtype), H.ptr, v.ptr, out.ptr))
    return out


class BatchedBFGS:
    """B independent ``BFGSOptimizer`` instances on one device (config 5)."""

    def __init__(self, problem_kind, x0, initial_step_length, device=None, matrices=None):
        """``device``: the GPU this shard lives on (default: the device selected with ``init``); a host that
        drives several shards from one process passes each shard's device and a :class:`Comm` built with
        ``Comm.init_all``.  ``matrices`` (B x n x n, each symmetric; with a QUADRATIC ``Problem``): instance b
        minimises 1/2 x'A_b x instead of sharing the problem's A (dzo_bfgs_batch_create_problem_matrices)."""
        _need_init()
        if device is not None:
            init(int(device))                           # x0 is uploaded to that device
        x0 = _as_dev(x0)
        self.batch, self.n = x0.shape
        self.dtype = x0.dtype
        self._x0 = x0
        h = C.c_void_p()
        if matrices is not None:
            assert isinstance(problem_kind, Problem), "per-instance matrices need a QUADRATIC Problem (n, dtype, decorators)"
            self.problem = problem_kind
            assert self.problem.n == self.n and self.problem.dtype == self.dtype
            if not isinstance(matrices, DeviceArray):   # symmetric, so the row-major host layout IS column-major
                matrices = DeviceArray.from_host(np.ascontiguousarray(matrices, dtype=self.dtype))
            assert tuple(matrices.shape) == (self.batch, self.n, self.n), matrices.shape
            self._matrices = matrices                   # the caller's array must outlive the batch
            _check(lib().dzo_bfgs_batch_create_problem_matrices(self.problem.h, self.batch, matrices.ptr, self.n * self.n, x0.ptr,
                                                                initial_step_length, -1 if device is None else int(device), C.byref(h)))
        elif isinstance(problem_kind, Problem):
            # objective, shared matrix and decorators from a problem handle (dzo_bfgs_batch_create_problem)
            self.problem = problem_kind
            assert self.problem.n == self.n and self.problem.dtype == self.dtype
            _check(lib().dzo_bfgs_batch_create_problem(self.problem.h, self.batch, x0.ptr, initial_step_length,
                                                       -1 if device is None else int(device), C.byref(h)))
        elif device is None:
            _check(lib().dzo_bfgs_batch_create(problem_kind, self.batch, self.n, _dt(self.dtype), x0.ptr,
                                               initial_step_length, C.byref(h)))
        else:
            _check(lib().dzo_bfgs_batch_create_on(int(device), problem_kind, self.batch, self.n, _dt(self.dtype), x0.ptr,
                                                  initial_step_length, C.byref(h)))
        self.h = h

    @property
    def device(self):
        v = C.c_int32()
        _check(lib().dzo_bfgs_batch_device(self.h, C.byref(v)))
        return v.value

    def set_max_increases(self, v):
        """``QuadraticLineSearch.max_increases`` (legacy/DZOptimization.jl:181-188) of every instance."""
        _check(lib().dzo_bfgs_batch_set_max_increases(self.h, int(v)))

    def step(self, steps=1, poll=True):
        """Runs ``steps`` step! calls on every live instance; returns all_done if ``poll``."""
        done = C.c_int32()
        _check(lib().dzo_bfgs_batch_step(self.h, steps, C.byref(done) if poll else None))
        return bool(done.value) if poll else None

    def _p(self, what, shape, dtype=None):
        p = C.c_void_p()
        _check(lib().dzo_bfgs_batch_get_ptr(self.h, what, C.byref(p)))
        return DeviceArray(shape, self.dtype if dtype is None else dtype, ptr=p.value, owner=False)

    current_point = property(lambda s: s._p(0, (s.batch, s.n)))
    current_gradient = property(lambda s: s._p(1, (s.batch, s.n)))
    approximate_inverse_hessian = property(lambda s: s._p(2, (s.batch, s.n, s.n)))
    current_objective_value = property(lambda s: s._p(3, (s.batch,), np.float64))
    has_terminated = property(lambda s: s._p(4, (s.batch,), np.int32))
    iteration_count = property(lambda s: s._p(5, (s.batch,), np.int64))
    delta_point = property(lambda s: s._p(6, (s.batch, s.n)))
    delta_gradient = property(lambda s: s._p(7, (s.batch, s.n)))
    next_step_direction = property(lambda s: s._p(8, (s.batch, s.n)))
    last_step_length = property(lambda s: s._p(9, (s.batch,), np.float64))
    last_step_type = property(lambda s: s._p(10, (s.batch,), np.int32))

    def count_active(self):
        v = C.c_int64()
        _check(lib().dzo_bfgs_batch_count_active(self.h, C.byref(v)))
        return v.value

    def install_state(self, x, g, H, d, f, last_step_length, iteration_count=None, last_step_type=None,
                      has_terminated=None, dx=None, dg=None):
        """Overwrite the state of every instance: the arrays behind ``dzo_bfgs_batch_get_ptr`` ARE the
        state (include/dzo.h).  ``H``: (B, n, n), each symmetric or column-major."""
        B, n, dt = self.batch, self.n, self.dtype
        self.current_point.upload(np.asarray(x, dt).reshape(B, n))
        self.current_gradient.upload(np.asarray(g, dt).reshape(B, n))
        self.approximate_inverse_hessian.upload(np.ascontiguousarray(np.transpose(np.asarray(H, dt).reshape(B, n, n), (0, 2, 1))))
        self.next_step_direction.upload(np.asarray(d, dt).reshape(B, n))
        self.current_objective_value.upload(np.asarray(f, np.float64).reshape(B))
        self.last_step_length.upload(np.asarray(last_step_length, np.float64).reshape(B))
        self.iteration_count.upload(np.zeros(B, np.int64) if iteration_count is None else np.asarray(iteration_count, np.int64))
        self.last_step_type.upload(np.zeros(B, np.int32) if last_step_type is None else np.asarray(last_step_type, np.int32))
        self.has_terminated.upload(np.zeros(B, np.int32) if has_terminated is None else np.asarray(has_terminated, np.int32))
        if dx is not None:
            self.delta_point.upload(np.asarray(dx, dt).reshape(B, n))
        if dg is not None:
            self.delta_gradient.upload(np.asarray(dg, dt).reshape(B, n))
        return self

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            lib().dzo_bfgs_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _preload_rccl():
    """Same reasoning as _preload_hip_runtime: libdzo_hip.so dlopen()s librccl.so.1 at first use; when
    PyTorch is installed its bundled copy (same SONAME) must be the one in the process."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "librccl.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


class Comm:
    """The one collective of the design (include/dzo.h): all-reduce(MIN) of the convergence flag of
    sharded independent optimizers over RCCL / xGMI, behind the C ABI.

    ``Comm.init_all(devices)``: one process, several GPUs (ncclCommInitAll).
    ``Comm.init_rank(uid, nranks, rank)``: one process per GPU; ``Comm.unique_id()`` on rank 0, carried to
    the other ranks by the launcher (``Comm.from_torch_distributed()`` does that over an initialised
    ``torch.distributed`` group)."""

    def __init__(self, h):
        self.h = h

    @staticmethod
    def unique_id() -> bytes:
        _need_init()
        _preload_rccl()
        buf = C.create_string_buffer(128)
        _check(lib().dzo_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def init_rank(cls, uid: bytes, nranks: int, rank: int):
        _need_init()
        _preload_rccl()
        assert len(uid) == 128
        h = C.c_void_p()
        _check(lib().dzo_comm_init_rank(C.create_string_buffer(uid, 128), nranks, rank, C.byref(h)))
        return cls(h)

    @classmethod
    def init_all(cls, devices):
        lib()
        _preload_rccl()
        arr = (C.c_int32 * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        _check(lib().dzo_comm_init_all(arr, len(devices), C.byref(h)))
        global _inited
        _inited = True
        return cls(h)

    @classmethod
    def from_torch_distributed(cls):
        """One rank per process of an initialised ``torch.distributed`` group: rank 0's unique id is
        broadcast through the group (any backend), then every rank joins."""
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()
        # Every rank runs the SAME sequence of collectives whatever fails locally (a rank that raised before a
        # collective the others have entered would hang the group): (1) every rank probes that it can load RCCL
        # through the C ABI -- rank 0's probe is the id itself -- and the outcomes are gathered; (2) only when all
        # are fine does anyone enter the RCCL bootstrap; (3) the outcomes of that are gathered as well.
        uid, err = None, None
        try:
            uid = cls.unique_id()                         # (ranks > 0: a probe; their id is dropped)
        except Exception as e:                            # noqa: BLE001 -- reported to every rank below
            err = f"rank {rank}: {e}"
        box = [None] * world
        dist.all_gather_object(box, (uid if rank == 0 else None, err))
        errs = [b[1] for b in box if b[1]]
        if errs:
            raise DzoError(5, "RCCL communicator not created (no rank entered the bootstrap): " + "; ".join(errs))
        comm, err = None, None
        try:
            comm = cls.init_rank(box[0][0], world, rank)
        except Exception as e:                            # noqa: BLE001
            err = f"rank {rank}: {e}"
        box = [None] * world
        dist.all_gather_object(box, err)
        errs = [b for b in box if b]
        if errs:
            if comm is not None:
                comm.close()
            raise DzoError(2, "ncclCommInitRank failed: " + "; ".join(errs))
        return comm

    def _info(self):
        a, b, c, d = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
        _check(lib().dzo_comm_info(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return a.value, b.value, c.value, d.value

    nranks = property(lambda s: s._info()[0])
    nlocal = property(lambda s: s._info()[1])
    first_rank = property(lambda s: s._info()[2])
    collectives = property(lambda s: s._info()[3])

    def allreduce_min(self, local_flags) -> int:
        flags = [int(local_flags)] if np.isscalar(local_flags) or isinstance(local_flags, bool) else [int(f) for f in local_flags]
        arr = (C.c_int32 * len(flags))(*flags)
        out = C.c_int32()
        _check(lib().dzo_flag_allreduce_min_n(self.h, arr, len(flags), C.byref(out)))   # (the count is checked against the local ranks)
        return out.value

    def all_done(self, batches) -> bool:
        """``dzo_bfgs_batch_all_done``: every instance of every shard (local and remote) has terminated."""
        hs = (C.c_void_p * len(batches))(*[b.h for b in batches])
        out = C.c_int32()
        _check(lib().dzo_bfgs_batch_all_done(self.h, hs, len(batches), C.byref(out)))
        return bool(out.value)

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            lib().dzo_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def batches_all_done(batches, comm=None) -> bool:
    """``dzo_bfgs_batch_all_done`` with or without a communicator."""
    hs = (C.c_void_p * len(batches))(*[b.h for b in batches])
    out = C.c_int32()
    _check(lib().dzo_bfgs_batch_all_done(comm.h if comm is not None else None, hs, len(batches), C.byref(out)))
    return bool(out.value)


def step_(opt):
    """``step!(opt)``: the reference's generic function (src/DZOptimization.jl:104)."""
    return opt.step()


__all__ = [
    "LBFGSOptimizer", "BFGSOptimizer", "AdGDOptimizer", "GradientDescentOptimizer", "BatchedBFGS", "Comm", "batches_all_done", "LineSearchEvaluator", "Problem", "NativeCallbacks", "DeviceArray", "step_",
    "axpy_", "axpby_", "rmul_", "copy_", "fill_", "dot", "norm", "isequal", "trial_point_", "box_clamp_",
    "norm2", "inv_norm", "negate_", "scale_",
    "update_inverse_hessian_", "update_inverse_hessian_mfma_", "symv_", "init", "build", "lib", "device_info", "device_count", "synchronize",
    "profile_enable", "profile_reset", "profile_table", "DzoError", "AssertionFailed",
]
