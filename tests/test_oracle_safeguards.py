"""CPU tests of the optional L-BFGS safeguards in the oracle:
legacy descent check / steepest-descent fallback with history reset
(legacy/DZOptimization.jl:588-610, :682-692) and the strong-Wolfe search on the
LineSearchEvaluator quotients (src/DZOptimization.jl:65-92).  With every option off the
optimizer is the live reference's step! (covered by test_oracle.py)."""
import numpy as np
import pytest

from oracle import oracle as orc


def _lbfgs(n=60, m=5, dtype=np.float64):
    p = orc.Problem(orc.ROSENBROCK_CHAIN, n, dtype)
    return orc.LBFGS(p, orc.rosenbrock_chain_x0(n, dtype), 1.0, m), p


def test_options_off_is_the_reference_trajectory():
    a, _ = _lbfgs()
    b, _ = _lbfgs()
    b.set_safeguards(False, False); b.set_line_search(0)
    for _ in range(40):
        a.step(); b.step()
    assert np.array_equal(a.current_point, b.current_point) and a.last_trials == b.last_trials


def test_safeguards_are_inert_on_a_healthy_run():
    a, _ = _lbfgs()
    b, _ = _lbfgs()
    b.set_safeguards(True, True)
    for _ in range(60):
        a.step(); b.step()
    assert np.array_equal(a.current_point, b.current_point)
    assert b.history_resets == 0 and b.descent_resets == 0
    assert b.last_step_length == pytest.approx(np.linalg.norm(b.delta_point), rel=1e-14)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_wolfe_steps_satisfy_both_conditions_and_converge(dtype):
    n = 40
    o, p = _lbfgs(n, 6, dtype)
    c1, c2 = 1e-4, 0.9
    o.set_line_search(1, c1, c2, 40)
    evals = 0
    for it in range(3000):
        x0, g0, f0 = o.current_point.copy(), o.current_gradient.copy(), o.current_objective_value
        o.step()
        if o.is_stuck:
            break
        evals += o.last_trials
        s, y = o.delta_point.astype(np.float64), o.delta_gradient.astype(np.float64)
        assert float(s @ y) > 0.0, it                                   # curvature: every pair is usable
        # conditions along the ACTUAL step s = t d (t d recovered as s): Armijo and strong curvature
        slope0 = float(g0.astype(np.float64) @ s)
        assert slope0 < 0
        tol = 1e-4 if dtype == np.float32 else 1e-9
        assert o.current_objective_value - f0 <= c1 * slope0 * (1 - tol) + tol * abs(f0), it
        assert abs(float(o.current_gradient.astype(np.float64) @ s)) <= c2 * abs(slope0) * (1 + tol) + tol * abs(slope0), it
        assert np.array_equal(o.current_gradient, p.grad(o.current_point))   # trial gradient reused
    assert o.is_stuck
    assert o.current_objective_value < (1e-9 if dtype == np.float32 else 1e-20)
    assert evals < 6 * it


def test_descent_check_replaces_an_ascent_direction():
    n, m = 50, 4
    o, p = _lbfgs(n, m)
    for _ in range(6):
        o.step()
    S, Y = o.history_arrays()
    o.set_history(S, -Y, iteration_count=o.iteration_count)          # s.y < 0: H_k is negative definite
    plain, _ = _lbfgs(n, m)
    plain.current_point[:] = o.current_point; plain.current_gradient[:] = o.current_gradient
    plain.set_history(S, -Y, iteration_count=o.iteration_count)
    plain.set_max_halvings(64)
    o.set_safeguards(descent_check=True)
    f0, g0 = o.current_objective_value, o.current_gradient.copy()
    lsl = o.last_step_length
    o.step()
    assert o.descent_resets == 1 and o.last_step_kind == 1 and not o.is_stuck
    assert o.current_objective_value < f0
    # the step went along -g, first trial length = last_step_length
    d = o.step_direction
    assert np.allclose(d, -lsl * g0 / np.linalg.norm(g0), rtol=1e-13, atol=0)
    assert o.history_count == m                                       # no history reset on this path (:688-690)


def test_fallback_resets_history_after_a_failed_quasi_newton_search():
    n, m = 50, 4
    o, p = _lbfgs(n, m)
    for _ in range(6):
        o.step()
    S, Y = o.history_arrays()
    o.set_history(S, 0 * Y, iteration_count=o.iteration_count)        # rho = 0 -> NaN direction
    o.set_max_halvings(64)
    o.set_safeguards(steepest_descent_fallback=True)
    f0 = o.current_objective_value
    o.step()
    assert not o.is_stuck and o.history_resets == 1 and o.last_step_kind == 2
    assert o.current_objective_value < f0
    assert o.history_count == 1                                       # legacy :609 then the push of this step
    assert np.array_equal(o.S(0), o.delta_point) and np.array_equal(o.Y(0), o.delta_gradient)
    o.step()                                                          # and the optimizer carries on
    assert not o.is_stuck and o.history_count == 2
    # without the fallback the same state is terminal
    q, _ = _lbfgs(n, m)
    for _ in range(6):
        q.step()
    q.set_history(S, 0 * Y, iteration_count=q.iteration_count)
    q.set_max_halvings(64)
    q.step()
    assert q.is_stuck
