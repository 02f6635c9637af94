"""Pins the CPU oracle (oracle/dzo_oracle.c).

The reference ships no tests or golden vectors for this path and cannot be executed here
(it is Julia), so the oracle is pinned by:
  (1) an independent big-integer restatement of PCG32 (legacy/PCG.jl:7-22);
  (2) an mpmath twin of the two-loop recursion / dense update (oracle/mp_twoloop.py);
  (3) analytic identities: two-loop == -H_k g from the dense recursion, secant equation,
      symmetry, k=0 and k=1 closed forms;
  (4) the invariants of the reference's own (dead) run_and_test! checker
      (legacy/DZOptimization.jl:998-1049), which hold EXACTLY;
  (5) convergence of config 1 (2-D Rosenbrock) to (1, 1).
"""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import mp_twoloop, oracle as orc


# ------------------------------------------------------------------------------- PCG32 (1)

def _pcg_py(n, seed):
    """Independent restatement of legacy/PCG.jl:7-22 with Python big integers."""
    M64 = (1 << 64) - 1
    mult, inc = 0x5851F42D4C957F2D, 0x14057B7EF767814F
    adv = lambda s: (mult * s + inc) & M64
    state = adv((inc + seed) & M64)
    out = []
    for _ in range(n):
        v = (((state >> 18) ^ state) >> 27) & 0xFFFFFFFF
        r = state >> 59
        out.append(((v >> r) | (v << ((32 - r) & 31))) & 0xFFFFFFFF)
        state = adv(state)
    return np.array(out, dtype=np.uint32)


@pytest.mark.parametrize("seed", [0, 1, 5, 42, 2**40 + 7])
def test_pcg_matches_independent_restatement(seed):
    raw = orc.pcg_raw(257, seed)
    assert np.array_equal(raw, _pcg_py(257, seed))
    u = orc.pcg_fill(257, seed)
    assert np.array_equal(u, raw.astype(np.float64) * 2.0**-32)
    assert (u >= 0).all() and (u < 1).all()
    u32 = orc.pcg_fill(257, seed, np.float32)
    assert np.array_equal(u32, (raw.astype(np.float64) * 2.0**-32).astype(np.float32))


def test_pcg_is_roughly_uniform():
    u = orc.pcg_fill(200_000, 7)
    assert abs(u.mean() - 0.5) < 5e-3 and abs(u.var() - 1 / 12) < 5e-3


# --------------------------------------------------------------------------- primitives (a7)

def test_dot_modes_agree_and_sequential_is_literal():
    rng = np.random.default_rng(0)
    a, b = rng.standard_normal(1001), rng.standard_normal(1001)
    seq = 0.0
    for x, y in zip(a, b):
        seq += x * y                               # legacy/Kernels.jl:15-18, literally
    orc.set_dot_mode(orc.DOT_SEQUENTIAL)
    assert orc.dot(a, b) == seq
    ref = float(np.dot(a.astype(np.longdouble), b.astype(np.longdouble)))
    for mode in (orc.DOT_EIGHT_LANE, orc.DOT_WIDE):
        orc.set_dot_mode(mode)
        assert abs(orc.dot(a, b) - ref) <= 1e-13 * np.abs(a * b).sum()
    orc.set_dot_mode(orc.DOT_SEQUENTIAL)


def test_isequal_semantics():
    a = np.array([1.0, np.nan, 0.0])
    assert orc.isequal(a, a.copy())                       # NaN equals NaN
    b = a.copy(); b[2] = -0.0
    assert not orc.isequal(a, b)                          # -0.0 differs from +0.0
    c = a.copy(); c[0] = np.nextafter(1.0, 2.0)
    assert not orc.isequal(a, c)


def test_axpby_delta_is_exact_difference():
    rng = np.random.default_rng(1)
    x, y = rng.standard_normal(100), rng.standard_normal(100)
    want = x - y
    orc.axpby(1.0, x, -1.0, y)
    assert np.array_equal(y, want)


def test_axpy_is_fused():
    x = np.array([1.0 + 2.0**-30]); y = np.array([-1.0])
    a = 1.0 - 2.0**-30
    orc.axpy(a, x, y)
    assert y[0] == -(2.0**-60)                            # a*x = 1 - 2^-60 exactly; fma keeps it


# ------------------------------------------------------------------ two-loop recursion (a3)

def _pairs(n, k, seed, dtype=np.float64):
    rng = np.random.default_rng(seed)
    S = rng.standard_normal((k, n))
    Y = S * rng.uniform(0.5, 2.0, size=(1, n)) + 0.1 * rng.standard_normal((k, n))
    g = rng.standard_normal(n)
    rho = np.array([orc.dot(S[i].copy(), Y[i].copy()) for i in range(k)])
    return g.astype(dtype), S.astype(dtype), Y.astype(dtype), rho.astype(dtype)


def test_two_loop_k0_is_plain_copy():
    g = np.arange(5, dtype=np.float64)
    d, alpha = orc.lbfgs_direction(g, np.zeros((0, 5)), np.zeros((0, 5)), np.zeros(0))
    assert np.array_equal(d, g) and alpha.size == 0       # :438 then the :443 guard


def test_two_loop_k1_closed_form():
    g, S, Y, rho = _pairs(40, 1, 3)
    s, y = S[0], Y[0]
    d, _ = orc.lbfgs_direction(g, S, Y, rho)
    a = s @ g / (s @ y)
    q = g - a * y
    r = -(s @ y) / (y @ y) * q
    b = y @ r / (s @ y)
    want = r - (a + b) * s
    assert np.linalg.norm(d - want) <= 1e-13 * np.linalg.norm(want)


@pytest.mark.parametrize("n,k,seed", [(50, 5, 0), (33, 8, 1), (7, 3, 2)])
def test_two_loop_matches_mpmath(n, k, seed):
    g, S, Y, rho = _pairs(n, k, seed)
    d, alpha = orc.lbfgs_direction(g, S, Y, rho)
    d_mp, alpha_mp = mp_twoloop.two_loop(g, S, Y, rho)
    assert np.linalg.norm(d - d_mp) <= 1e-12 * np.linalg.norm(d_mp)
    assert np.allclose(alpha, alpha_mp, rtol=1e-11, atol=0)


def test_two_loop_equals_dense_inverse_recursion():
    g, S, Y, rho = _pairs(12, 4, 5)
    d, _ = orc.lbfgs_direction(g, S, Y, rho)
    d_dense = mp_twoloop.dense_inverse_from_pairs(g, S, Y)
    assert np.linalg.norm(d - d_dense) <= 1e-11 * np.linalg.norm(d_dense)


@settings(max_examples=25, deadline=None, derandomize=True)
@given(n=st.integers(1, 64), k=st.integers(0, 6), seed=st.integers(0, 2**31))
def test_two_loop_dot_mode_invariance(n, k, seed):
    g, S, Y, rho = _pairs(n, k, seed)
    outs = []
    for mode in (orc.DOT_SEQUENTIAL, orc.DOT_EIGHT_LANE, orc.DOT_WIDE):
        orc.set_dot_mode(mode)
        outs.append(orc.lbfgs_direction(g, S, Y, rho)[0])
    orc.set_dot_mode(orc.DOT_SEQUENTIAL)
    scale = max(np.linalg.norm(outs[2]), 1e-300)
    assert np.linalg.norm(outs[0] - outs[2]) <= 1e-9 * scale
    assert np.linalg.norm(outs[1] - outs[2]) <= 1e-9 * scale


# ----------------------------------------------------------------- L-BFGS step! (a4, a5)

def _run_and_test(opt, recompute_f, recompute_g, max_steps=10_000, flag="is_stuck"):
    """legacy/DZOptimization.jl:998-1049, restated for any optimizer wrapper."""
    hist = []

    def snap():
        hist.append(dict(stuck=getattr(opt, flag), it=opt.iteration_count,
                         x=opt.current_point.copy(), g=opt.current_gradient.copy(),
                         dx=opt.delta_point.copy(), dg=opt.delta_gradient.copy(),
                         f=opt.current_objective_value))
    snap()
    while not getattr(opt, flag) and len(hist) <= max_steps:
        opt.step()
        snap()
    assert hist[-1]["stuck"], "did not terminate"
    assert not any(h["stuck"] for h in hist[:-1])                      # :1007-1010
    for i, h in enumerate(hist[:-1]):
        assert h["it"] == hist[0]["it"] + i                            # :1013-1015
    assert hist[-2]["it"] == hist[-1]["it"]                            # :1016
    for h in hist:
        assert recompute_f(h["x"]) == h["f"]                           # :1019-1022 exact
        assert np.array_equal(recompute_g(h["x"]), h["g"])             # :1025-1032 exact
    for i in range(len(hist) - 2):
        assert np.array_equal(hist[i + 1]["x"] - hist[i]["x"], hist[i + 1]["dx"])  # :1035-1039
        assert np.array_equal(hist[i + 1]["g"] - hist[i]["g"], hist[i + 1]["dg"])  # :1042-1046
    assert np.array_equal(hist[-2]["x"], hist[-1]["x"])                # :1039
    assert np.array_equal(hist[-2]["g"], hist[-1]["g"])                # :1046
    return hist


def test_lbfgs_constructor_matches_reference_init():
    p = orc.Problem(orc.ROSENBROCK_CHAIN, 10)
    x0 = orc.rosenbrock_chain_x0(10)
    opt = orc.LBFGS(p, x0.copy(), 0.5, 4)
    g0 = p.grad(x0)
    assert np.array_equal(opt.current_gradient, g0)
    assert opt.current_objective_value == p.eval(x0)
    assert np.allclose(opt.step_direction, -0.5 * g0 / np.linalg.norm(g0), rtol=1e-15)
    assert not opt.delta_point.any() and not opt.delta_gradient.any()
    assert opt.history_count == 0 and opt.iteration_count == 0 and not opt.is_stuck
    assert opt.current_point.ctypes.data == opt.x.ctypes.data           # aliasing, :393


def test_lbfgs_zero_gradient_is_stuck_immediately():
    p = orc.Problem(orc.ROSENBROCK_CHAIN, 6)
    opt = orc.LBFGS(p, np.ones(6), 1.0, 3)                              # x = 1 is the minimiser
    assert opt.is_stuck and not opt.step_direction.any()                # :382-384
    opt.step()
    assert opt.iteration_count == 0


@pytest.mark.parametrize("n,m", [(2, 3), (10, 5), (50, 20)])
def test_lbfgs_invariants_and_convergence_rosenbrock_chain(n, m):
    p = orc.Problem(orc.ROSENBROCK_CHAIN, n)
    opt = orc.LBFGS(p, orc.rosenbrock_chain_x0(n), 1.0, m)
    hist = _run_and_test(opt, p.eval, p.grad)
    assert hist[-1]["f"] < 1e-12 or np.linalg.norm(hist[-1]["g"]) < 1e-5
    assert opt.history_count == min(m, opt.iteration_count)
    assert len(opt.rho_history) == min(m, opt.iteration_count)
    # rho[0] is the newest s.y and the histories are newest-first (:483,487,505)
    assert np.array_equal(opt.S(0), hist[-2]["dx"]) and np.array_equal(opt.Y(0), hist[-2]["dg"])
    assert opt.rho_history[0] == orc.dot(opt.S(0).copy(), opt.Y(0).copy())


def test_lbfgs_step_direction_is_two_loop_of_history():
    n, m = 30, 6
    p = orc.Problem(orc.ROSENBROCK_CHAIN, n)
    opt = orc.LBFGS(p, orc.rosenbrock_chain_x0(n), 1.0, m)
    for _ in range(10):
        opt.step()
    S, Y = opt.history_arrays()
    rho, g = opt.rho_history, opt.current_gradient.copy()
    opt.step()
    # d was computed from the pre-step history; replay it
    d_replay, _ = orc.lbfgs_direction(g, S, Y, rho)
    assert np.array_equal(d_replay, opt.step_direction)


def test_lbfgs_fp32_instantiation_runs():
    p = orc.Problem(orc.ROSENBROCK_CHAIN, 16, dtype=np.float32)
    opt = orc.LBFGS(p, orc.rosenbrock_chain_x0(16, np.float32), 1.0, 5)
    f0 = opt.current_objective_value
    for _ in range(300):
        opt.step()
    assert opt.is_stuck and opt.current_objective_value < 1e-9 * f0
    assert opt.current_point.dtype == np.float32


def test_nan_direction_escape_is_bounded():
    p = orc.Problem(orc.ROSENBROCK_CHAIN, 4)
    opt = orc.LBFGS(p, orc.rosenbrock_chain_x0(4), 1.0, 2)
    opt.step_direction[:] = np.nan
    opt.set_max_halvings(8)
    x_before = opt.current_point.copy()
    opt.step()
    assert opt.is_stuck and np.array_equal(opt.current_point, x_before)


# ------------------------------------------------------------------------------ AdGD (8f.1)

def test_adgd_invariants_and_progress():
    n = 12
    A = orc.quadratic_matrix(n)
    p = orc.Problem(orc.QUADRATIC, n, A=A)
    x0 = orc.pcg_fill(n, 4) - 0.5
    opt = orc.AdGD(p, x0.copy(), 0.1)
    f0 = opt.current_objective_value
    prev_x, prev_g = opt.current_point.copy(), opt.current_gradient.copy()
    for i in range(200):
        opt.step()
        if opt.is_stuck:
            break
        assert opt.iteration_count == i + 1
        assert np.array_equal(opt.current_point - prev_x, opt.delta_point)
        assert np.array_equal(opt.current_gradient - prev_g, opt.delta_gradient)
        assert p.eval(opt.current_point) == opt.current_objective_value
        prev_x, prev_g = opt.current_point.copy(), opt.current_gradient.copy()
    assert opt.current_objective_value < 1e-6 * f0


# ----------------------------------------------------------------------- dense BFGS (a9-a13)

def test_bfgs_update_secant_symmetry_and_mpmath():
    n = 9
    rng = np.random.default_rng(11)
    M = rng.standard_normal((n, n))
    H = np.asfortranarray(M @ M.T + n * np.eye(n))
    d = rng.standard_normal(n)
    y = rng.standard_normal(n)
    lam = -0.37 if d @ y < 0 else 0.37                  # make s.y = lam * d.y > 0
    H_mp = mp_twoloop.bfgs_update(H.copy(), lam, d, y)
    H_new, d_scaled = H.copy(order="F"), d.copy()
    t = orc.bfgs_update(H_new, lam, d_scaled, y.copy())
    assert np.array_equal(H_new, H_new.T)                # symmetry preserved exactly
    s = lam * d
    assert np.allclose(H_new @ y, s, rtol=1e-10, atol=1e-12)      # secant equation H+ y = s
    assert np.linalg.norm(H_new - H_mp) <= 1e-12 * np.linalg.norm(H_mp)
    assert np.allclose(d_scaled, d / (d @ y), rtol=1e-15)         # :874 side effect
    assert np.allclose(t, H @ y, rtol=1e-13)
    # textbook form (I - r s y')H(I - r y s') + r s s'
    r = 1.0 / (s @ y)
    V = np.eye(n) - r * np.outer(s, y)
    assert np.allclose(H_new, V @ H @ V.T + r * np.outer(s, s), rtol=1e-10, atol=1e-10)


def test_bfgs_line_search_parabola_vertex_on_exact_quadratic():
    # f(x) = 1/2 x'Ax along -g: exact minimiser t* = g.g / g.A.g; the search brackets it and
    # one parabola vertex through (0, x1, 2x1) lands on it (legacy :203-209).
    n = 6
    A = orc.quadratic_matrix(n)
    p = orc.Problem(orc.QUADRATIC, n, A=A)
    x0 = orc.pcg_fill(n, 4) - 0.5
    opt = orc.BFGS(p, x0, 1.0)
    g = opt.current_gradient.copy()
    t_star = (g @ g) / (g @ A @ g)
    t, f = opt.line_search(True, 1.0 / np.linalg.norm(g))
    assert abs(t - t_star) <= 1e-9 * t_star
    assert f < opt.current_objective_value


def test_bfgs_readme_example_converges_config1():
    """Config 1: BFGSOptimizer on 2-D Rosenbrock, rand(2) start (README.md:33-41)."""
    p = orc.Problem(orc.ROSENBROCK2D, 2)
    x0 = orc.pcg_fill(2, 1)
    opt = orc.BFGS(p, x0, 1.0)
    hist = _run_and_test(opt, p.eval, p.grad, flag="has_terminated")
    assert np.allclose(opt.current_point, [1.0, 1.0], atol=1e-6)
    assert opt.current_objective_value < 1e-12
    assert len(hist) < 200
    H = opt.approximate_inverse_hessian
    assert np.array_equal(H, H.T)


def test_bfgs_quadratic_invariants_and_direction_identity():
    n = 16
    A = orc.quadratic_matrix(n)
    p = orc.Problem(orc.QUADRATIC, n, A=A)
    opt = orc.BFGS(p, orc.pcg_fill(n, 4) - 0.5, 1.0)
    for _ in range(5):
        opt.step()
    H = opt.approximate_inverse_hessian.copy()
    assert np.allclose(opt.next_step_direction, H @ opt.current_gradient, rtol=1e-12, atol=1e-14)
    hist = _run_and_test(opt, p.eval, p.grad, flag="has_terminated")
    assert hist[-1]["f"] < 1e-20 * max(hist[0]["f"], 1e-300) or hist[-1]["f"] < 1e-25


# ------------------------------------------------------- legacy GradientDescentOptimizer (8f.4)
def test_legacy_gradient_descent_invariants():
    """legacy/DZOptimization.jl:305-449 with QuadraticLineSearch; run_and_test! equalities."""
    n = 12
    p = orc.Problem(orc.ROSENBROCK_CHAIN, n)
    x0 = orc.rosenbrock_chain_x0(n)
    opt = orc.GradientDescent(p, x0, 0.1)
    g0 = p.grad(x0)
    assert np.allclose(opt.next_step_direction, -0.1 * g0 / np.linalg.norm(g0), rtol=1e-14)   # :354-357
    assert opt.last_step_length == 0.0 and not opt.has_terminated                              # :351,:364
    prev_x, prev_g, prev_f = opt.current_point.copy(), opt.current_gradient.copy(), opt.current_objective_value
    for i in range(60):
        opt.step()
        if opt.has_terminated:
            break
        x, g = opt.current_point.copy(), opt.current_gradient.copy()
        assert opt.iteration_count == i + 1
        assert np.array_equal(x - prev_x, opt.delta_point) and np.array_equal(g - prev_g, opt.delta_gradient)
        assert p.eval(x) == opt.current_objective_value < prev_f
        assert opt.delta_objective_value == opt.current_objective_value - prev_f               # :428-429
        assert opt.last_step_length == pytest.approx(np.linalg.norm(x - prev_x), rel=1e-14)    # :424
        want_d = -opt.last_step_length * g / np.linalg.norm(g)                                 # :445-446
        assert np.allclose(opt.next_step_direction, want_d, rtol=1e-13)
        prev_x, prev_g, prev_f = x, g, opt.current_objective_value
    assert opt.iteration_count > 20


def test_chained_quadratic_follows_its_formula_and_lbfgs_solves_it():
    """The chained quadratic (build-defined synthetic objective, DZO_PROBLEM_QUADRATIC_CHAIN: the large-n member of
    north_star's "synthetic quadratic" problems): f = sum 1/2 (x[i+1]-x[i])^2 + lambda/2 (x[i]-1)^2, its gradient against
    the closed form and central differences (legacy/ExampleFunctions.jl:290-303's idea), and L-BFGS (:454-509) converging
    to its minimiser x = 1."""
    n, lam = 37, 0.25
    x = orc.pcg_fill(n, 3) * 2
    p = orc.Problem(orc.QUADRATIC_CHAIN, n, lam=lam)
    f_ref = 0.5 * np.sum(np.diff(x) ** 2) + 0.5 * lam * np.sum((x - 1) ** 2)
    assert abs(p.eval(x) - f_ref) <= 1e-14 * abs(f_ref)
    g_ref = lam * (x - 1); g_ref[:-1] += x[:-1] - x[1:]; g_ref[1:] += x[1:] - x[:-1]
    assert np.allclose(p.grad(x), g_ref, rtol=1e-14, atol=1e-15)
    h = 1e-6
    fd = np.array([(p.eval(x + h * e) - p.eval(x - h * e)) / (2 * h) for e in np.eye(n)])
    assert np.allclose(p.grad(x), fd, atol=1e-7)
    opt = orc.LBFGS(p, x.copy(), 1.0, 6)
    for _ in range(400):
        opt.step()
        if opt.is_stuck:
            break
    assert np.abs(opt.current_point - 1).max() <= 1e-6 and opt.current_objective_value <= 1e-12
