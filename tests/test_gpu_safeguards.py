"""GPU parity of the optional L-BFGS safeguards (include/dzo.h: dzo_lbfgs_set_safeguards,
dzo_lbfgs_set_line_search) against the oracle, per step from identical uploaded state."""
import numpy as np
import pytest

from dzo_loader import dzo
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _pair(n, m, dtype=np.float64, mode=dzo.TWOLOOP_GRAM):
    x0 = orc.rosenbrock_chain_x0(n, dtype)
    ref = orc.LBFGS(orc.Problem(orc.ROSENBROCK_CHAIN, n, dtype), x0.copy(), 1.0, m)
    prob = dzo.Problem(dzo.ROSENBROCK_CHAIN, n, dtype)
    opt = dzo.LBFGSOptimizer(None, prob, None, dzo.DeviceArray.from_host(x0), 1.0, m)
    opt.set_two_loop_mode(mode)
    return opt, ref, prob


def _sync(opt, ref):
    opt.current_point.upload(ref.current_point)
    opt.current_gradient.upload(ref.current_gradient)
    opt.set_objective_value(ref.current_objective_value)
    S, Y = ref.history_arrays()
    opt.set_history(S, Y, ref.rho_history, iteration_count=ref.iteration_count)
    opt.set_last_step_length(ref.last_step_length)


def _same_step(opt, ref, tol=1e-10):
    assert opt.is_stuck == ref.is_stuck and opt.iteration_count == ref.iteration_count
    assert opt.last_trials == ref.last_trials
    assert opt.history_count == ref.history_count
    assert opt.last_step_kind == ref.last_step_kind
    assert rel(opt.current_point.to_host(), ref.current_point) <= 1e-12
    assert abs(opt.current_objective_value - ref.current_objective_value) <= 1e-12 * max(abs(ref.current_objective_value), 1e-300)
    if not ref.is_stuck:
        assert rel(opt.delta_point.to_host(), ref.delta_point) <= tol
        assert rel(opt.delta_gradient.to_host(), ref.delta_gradient) <= 1e-9
        assert np.allclose(opt.rho_history, ref.rho_history, rtol=1e-9)


@pytest.mark.parametrize("mode", [dzo.TWOLOOP_CHAIN, dzo.TWOLOOP_GRAM], ids=["chain", "gram"])
@pytest.mark.parametrize("n,m", [(2, 3), (1000, 5), (4099, 12)])
def test_wolfe_step_matches_oracle_on_identical_state(n, m, mode):
    opt, ref, prob = _pair(n, m, mode=mode)
    opt.set_line_search(dzo.LINE_SEARCH_WOLFE); ref.set_line_search(1)
    ref_p = orc.Problem(orc.ROSENBROCK_CHAIN, n)
    for it in range(60):
        _sync(opt, ref)
        opt.step(); ref.step()
        _same_step(opt, ref)
        if ref.is_stuck:
            break
        # the accepted trial gradient IS the gradient at the new point (bit-exact kernel)
        assert np.array_equal(opt.current_gradient.to_host(), ref_p.grad(opt.current_point.to_host()))
        assert float(opt.delta_point.to_host() @ opt.delta_gradient.to_host()) > 0


def test_wolfe_free_run_converges_with_positive_curvature_pairs():
    n, m = 1000, 8
    opt, _, _ = _pair(n, m)
    opt.set_line_search(dzo.LINE_SEARCH_WOLFE, 1e-4, 0.9, 40)
    steps = 0
    while not opt.is_stuck and steps < 20000:
        opt.step(); steps += 1
        if not opt.is_stuck:
            assert opt.rho_history[0] > 0
    assert opt.is_stuck and opt.current_objective_value < 1e-20
    assert np.allclose(opt.current_point.to_host(), 1.0, atol=1e-9)


def test_wolfe_with_host_callbacks_matches_builtin_problem():
    n, m = 257, 4
    x0 = orc.rosenbrock_chain_x0(n)
    ref_p = orc.Problem(orc.ROSENBROCK_CHAIN, n)
    a = dzo.LBFGSOptimizer(None, dzo.Problem(dzo.ROSENBROCK_CHAIN, n), None, dzo.DeviceArray.from_host(x0), 1.0, m)

    def objective(x):
        return ref_p.eval(x.to_host())

    def gradient(g, x):
        g.upload(ref_p.grad(x.to_host()))

    b = dzo.LBFGSOptimizer(None, objective, gradient, dzo.DeviceArray.from_host(x0), 1.0, m)
    for o in (a, b):
        o.set_line_search(dzo.LINE_SEARCH_WOLFE)
    for it in range(15):
        a.step(); b.step()
        assert a.last_trials == b.last_trials
        assert rel(b.current_point.to_host(), a.current_point.to_host()) <= 1e-11, it


def test_line_search_argument_checks():
    opt, _, _ = _pair(16, 2)
    with pytest.raises(dzo.DzoError):
        opt.set_line_search(7)
    with pytest.raises(dzo.DzoError):
        opt.set_line_search(dzo.LINE_SEARCH_WOLFE, 0.5, 0.1)
    with pytest.raises(dzo.DzoError):
        opt.set_line_search(dzo.LINE_SEARCH_WOLFE, 1e-4, 1.5)


@pytest.mark.parametrize("mode", [dzo.TWOLOOP_CHAIN, dzo.TWOLOOP_GRAM], ids=["chain", "gram"])
def test_safeguards_inert_on_healthy_steps(mode):
    n, m = 1000, 5
    opt, ref, _ = _pair(n, m, mode=mode)
    opt.set_safeguards(True, True); ref.set_safeguards(True, True)
    for it in range(40):
        _sync(opt, ref)
        opt.step(); ref.step()
        _same_step(opt, ref)
        assert opt.last_step_length == pytest.approx(ref.last_step_length, rel=1e-12)
    assert opt.history_resets == 0 and opt.descent_resets == 0


@pytest.mark.parametrize("mode", [dzo.TWOLOOP_CHAIN, dzo.TWOLOOP_GRAM], ids=["chain", "gram"])
def test_descent_check_parity(mode):
    n, m = 1000, 4
    opt, ref, _ = _pair(n, m, mode=mode)
    for _ in range(6):
        ref.step()
    S, Y = ref.history_arrays()
    ref.set_history(S, -Y, iteration_count=ref.iteration_count)      # negative curvature everywhere
    opt.set_safeguards(True, False); ref.set_safeguards(True, False)
    _sync(opt, ref)
    opt.step(); ref.step()
    assert ref.descent_resets == 1 and opt.descent_resets == 1
    _same_step(opt, ref)
    assert rel(opt.step_direction.to_host(), ref.step_direction) <= 1e-13
    assert opt.history_count == m


@pytest.mark.parametrize("mode", [dzo.TWOLOOP_CHAIN, dzo.TWOLOOP_GRAM], ids=["chain", "gram"])
def test_fallback_parity_and_history_reset(mode):
    n, m = 1000, 4
    opt, ref, _ = _pair(n, m, mode=mode)
    for _ in range(6):
        ref.step()
    S, Y = ref.history_arrays()
    ref.set_history(S, 0 * Y, iteration_count=ref.iteration_count)   # rho = 0 -> non-finite direction
    for o in (opt, ref):
        o.set_max_halvings(64)
        o.set_safeguards(False, True)
    _sync(opt, ref)
    opt.step(); ref.step()
    assert ref.history_resets == 1 and opt.history_resets == 1 and opt.last_step_kind == 2
    assert opt.is_stuck == ref.is_stuck is False
    assert opt.history_count == ref.history_count == 1
    assert rel(opt.current_point.to_host(), ref.current_point) <= 1e-12
    assert rel(opt.delta_gradient.to_host(), ref.delta_gradient) <= 1e-9
    assert np.allclose(opt.rho_history, ref.rho_history, rtol=1e-9)
    # the next steps run on the restarted history, still in lock step from synced state
    for it in range(5):
        _sync(opt, ref)
        opt.step(); ref.step()
        _same_step(opt, ref)
    assert opt.history_count == min(m, 6)
    # without the fallback the same state is terminal on both sides
    opt2, ref2, _ = _pair(n, m, mode=mode)
    for _ in range(6):
        ref2.step()
    ref2.set_history(S, 0 * Y, iteration_count=ref2.iteration_count)
    for o in (opt2, ref2):
        o.set_max_halvings(64)
    _sync(opt2, ref2)
    opt2.step(); ref2.step()
    assert opt2.is_stuck and ref2.is_stuck
