"""Objective / constraint decorators of legacy/DZOptimization.jl:219-296:
L2RegularizationWrapper, L2GradientWrapper, UniformBoxConstraint, UniformBoxGradientWrapper.
CPU: the oracle against the formulas as written in the reference.  GPU: the device versions
against the oracle, and whole optimizer steps with them active."""
import numpy as np
import pytest

from oracle import oracle as orc


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def test_oracle_l2_wrappers_follow_the_reference_formulas():
    n, lam = 40, 0.125
    x = orc.pcg_fill(n, 3) - 0.5
    base = orc.Problem(orc.ROSENBROCK_CHAIN, n)
    wrapped = orc.Problem(orc.ROSENBROCK_CHAIN, n, l2=lam)
    assert wrapped.eval(x) == base.eval(x) + lam * orc.dot(x.copy(), x.copy())          # :231-232
    g = base.grad(x)
    orc.axpy(lam + lam, x, g)                                                            # :247
    assert np.array_equal(wrapped.grad(x), g)


def test_oracle_box_wrappers_follow_the_reference_formulas():
    n = 30
    x = (orc.pcg_fill(n, 4) - 0.5) * 4
    lo, hi = -0.5, 0.75
    clamped = orc.box_clamp(x.copy(), lo, hi)
    assert np.array_equal(clamped, np.clip(x, lo, hi))                                   # :269
    base = orc.Problem(orc.ROSENBROCK_CHAIN, n)
    boxed = orc.Problem(orc.ROSENBROCK_CHAIN, n, box_gradient=(lo, hi))
    g = base.grad(clamped)
    mask = ((clamped <= lo) & (g >= 0)) | ((clamped >= hi) & (g <= 0))                   # :290-291
    want = np.where(mask, 0.0, g)
    assert np.array_equal(boxed.grad(clamped), want) and mask.any()


def test_oracle_lbfgs_with_box_constraint_stays_feasible_and_decreases():
    n, lo, hi = 50, -0.25, 0.9
    p = orc.Problem(orc.ROSENBROCK_CHAIN, n, box_gradient=(lo, hi), box_constraint=(lo, hi))
    opt = orc.LBFGS(p, orc.rosenbrock_chain_x0(n), 1.0, 5)
    assert opt.current_point.min() >= lo and opt.current_point.max() <= hi              # :412-414 projects x0
    f_prev = opt.current_objective_value
    for _ in range(200):
        opt.step()
        if opt.is_stuck:
            break
        assert opt.current_point.min() >= lo and opt.current_point.max() <= hi
        assert opt.current_objective_value < f_prev
        f_prev = opt.current_objective_value
    assert opt.iteration_count > 3


# ------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_device_decorators_match_oracle(dtype):
    from dzo_loader import dzo
    n, lam, lo, hi = 10_007, 0.05, -0.5, 0.75
    x = ((orc.pcg_fill(n, 9) - 0.5) * 3).astype(dtype)
    dx = dzo.DeviceArray.from_host(x)
    assert dzo.box_clamp_(dx, lo, hi)
    xc = orc.box_clamp(x.copy(), lo, hi)
    assert np.array_equal(dx.to_host(), xc)
    ref = orc.Problem(orc.ROSENBROCK_CHAIN, n, dtype, l2=lam, box_gradient=(lo, hi))
    dev = dzo.Problem(dzo.ROSENBROCK_CHAIN, n, dtype, l2=lam, box_gradient=(lo, hi))
    g_dev = dev.gradient_(dzo.DeviceArray(n, dtype), dx).to_host()
    assert np.array_equal(g_dev, ref.grad(xc))                       # elementwise: bit-exact
    tol = 1e-6 if dtype == np.float32 else 1e-13
    assert abs(dev(dx) - ref.eval(xc)) <= tol * abs(ref.eval(xc))


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["lbfgs", "adgd", "bfgs"])
def test_optimizers_with_decorators_match_oracle_step_by_step(which):
    from dzo_loader import dzo
    n, lo, hi = (24 if which == "bfgs" else 600), -0.25, 0.9
    kw = dict(l2=0.01, box_gradient=(lo, hi), box_constraint=(lo, hi))
    ref_p = orc.Problem(orc.ROSENBROCK_CHAIN, n, **kw)
    dev_p = dzo.Problem(dzo.ROSENBROCK_CHAIN, n, **kw)
    x0 = orc.rosenbrock_chain_x0(n)
    if which == "lbfgs":
        ref = orc.LBFGS(ref_p, x0.copy(), 1.0, 4)
        opt = dzo.LBFGSOptimizer(None, dev_p, None, dzo.DeviceArray.from_host(x0), 1.0, 4)
    elif which == "adgd":
        ref = orc.AdGD(ref_p, x0.copy(), 0.1)
        opt = dzo.AdGDOptimizer(None, dev_p, None, dzo.DeviceArray.from_host(x0), 0.1)
    else:
        ref = orc.BFGS(ref_p, x0.copy(), 1.0)
        opt = dzo.BFGSOptimizer(dev_p, None, dzo.DeviceArray.from_host(x0), 1.0)
    assert rel(opt.current_point.to_host(), ref.current_point) == 0.0        # projected start
    scale = np.linalg.norm(ref.current_point)
    for it in range(12):
        opt.step(); ref.step()
        x = opt.current_point.to_host()
        assert x.min() >= lo and x.max() <= hi
        assert np.linalg.norm(x - ref.current_point) <= 1e-9 * scale, it
        assert abs(opt.current_objective_value - ref.current_objective_value) <= 1e-9 * abs(ref.current_objective_value)
        assert np.array_equal(opt.current_gradient.to_host(), ref_p.grad(x))   # decorated gradient, bit-exact
