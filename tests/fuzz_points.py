"""GPU fuzz of step!() on the point ring: random n (row boundaries of the 62-vector wave-rows included), history
lengths, initial step lengths (rejected first trials, deep halvings) and both dtypes; the GPU optimizer runs free,
the oracle is given its state before every step.  Test infrastructure (uses oracle/).  tests/test_gpu_fuzz.py runs
`run()` with a fixed seed and 20 cases under pytest; by hand for more:  FUZZ_CASES=200 FUZZ_SEED=7 python tests/fuzz_points.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dzo_loader import dzo  # noqa: E402
from oracle import oracle as orc  # noqa: E402

def rel(a, b):
    return np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / max(np.linalg.norm(b.astype(np.float64)), 1e-300)


def run(cases=60, seed=2468):
    """`cases` random optimizers; every step compared with the oracle.  Returns the worst errors and the step count."""
    rng = np.random.default_rng(seed)
    worst = {np.float64: 0.0, np.float32: 0.0}
    retries = steps_total = 0
    for ex in range(cases):
        dtype = np.float64 if rng.integers(0, 3) else np.float32
        vecn = 16 // np.dtype(dtype).itemsize
        base = int(rng.choice([4, 31, 61, 62, 63, 123, 124, 125, 186, 248, 500, 2047, 4099, 25000]))
        n = base * vecn
        if base > 4 and rng.integers(0, 2):                              # ragged: the last vector of every ring stream is padded with phantom elements
            n += int(rng.integers(1, vecn))
        m = int(rng.integers(1, 25 if dtype == np.float64 else 21))     # (fp64: up to the K = 24 instantiation)
        step0 = float(rng.choice([1e-2, 1.0, 1.0, 30.0, 3000.0]))
        x0 = ((orc.pcg_fill(n, int(rng.integers(0, 10**6))) - 0.5) * 2.0).astype(dtype)
        # a third of the cases with decorators (legacy/DZOptimization.jl:219-296) riding on the pass
        decor = {}
        which = int(rng.integers(0, 9))
        if which in (0, 2):
            decor["l2"] = float(rng.choice([1e-3, 0.05]))
        if which in (1, 2):
            lo, hi = float(rng.choice([-0.8, -0.3])), float(rng.choice([0.5, 0.95]))
            decor["box_gradient"] = (lo, hi)
            if rng.integers(0, 4):
                decor["box_constraint"] = (lo, hi)
        # every sixth case on one of the other objectives of the point ring: the chained quadratic (same pass, ChainObj<T, 1>) or
        # log-sum-exp (trial pass + dots pass); no decorators there
        other = int(rng.integers(0, 6))
        if dtype == np.float32:
            orc.set_dot_mode(orc.DOT_WIDE)
        try:
            if other == 0:
                decor = {}
                lam = float(rng.choice([1e-3, 0.05, 0.5]))
                ref_p, dev_p = orc.Problem(orc.QUADRATIC_CHAIN, n, dtype, lam=lam), dzo.Problem(dzo.QUADRATIC_CHAIN, n, dtype, lam=lam)
            elif other == 1:
                decor = {}
                cc = (orc.pcg_fill(n, int(rng.integers(0, 10**6))) - 0.5).astype(dtype)
                lam = float(rng.choice([1e-2, 1e-4]))
                ref_p, dev_p = orc.Problem(orc.LSE, n, dtype, c=cc, lam=lam), dzo.Problem(dzo.LSE, n, dtype, c=cc, lam=lam)
            else:
                ref_p, dev_p = orc.Problem(orc.ROSENBROCK_CHAIN, n, dtype, **decor), dzo.Problem(dzo.ROSENBROCK_CHAIN, n, dtype, **decor)
            ref = orc.LBFGS(ref_p, x0.copy(), step0, m)
            opt = dzo.LBFGSOptimizer(None, dev_p, None, dzo.DeviceArray.from_host(x0), step0, m)
            assert opt.ring_layout == 2, (n, m)
            for it in range(int(rng.integers(3, 2 * m + 8))):
                k = opt.history_count
                S = np.stack([h.to_host() for h in opt.delta_point_history]) if k else np.zeros((0, n), dtype)
                Y = np.stack([h.to_host() for h in opt.delta_gradient_history]) if k else np.zeros((0, n), dtype)
                ref.install_state(opt.current_point.to_host(), opt.current_gradient.to_host(), opt.current_objective_value,
                                  S, Y, opt.rho_history[:k], opt.iteration_count)
                f_before = opt.current_objective_value
                if other in (0, 1) and it > 0 and abs(opt.delta_objective_value) <= (1e-12 if dtype == np.float64 else 1e-5) * max(abs(f_before), 1e-30):
                    break        # (a convex objective at its minimiser: what is left is rounding noise)
                opt.step(); ref.step()
                assert opt.ring_layout == 2, ("left the point ring", ex, it, n, m, step0, np.dtype(dtype).name, other, decor, opt.last_trials, ref.last_trials, opt.is_stuck)
                if opt.is_stuck != ref.is_stuck:
                    # only after dozens of halvings, where f_new - f is one unit in the last place and the two summation
                    # orders may decide differently (seed 12, case 2: the oracle accepts trial 43 with f lower by 5e-14,
                    # the GPU halves on to x + t d == x)
                    # ... or at the minimiser itself: f has fallen by 20 orders of magnitude, every trial's f_new - f is rounding
                    # noise whatever the step (seed 406, case 72: n = 8, 47 steps in)
                    # ... or at a minimiser itself (the global one, or the local one of the chained function at f = 3.98...): the
                    # side that still found a decrease found one at rounding level, whatever the step (seed 406, case 72: n = 8,
                    # 47 steps in, f = 3.9858877695994894: the oracle accepts trial 14, the GPU halves on until x + t d == x)
                    moved = ref if opt.is_stuck else opt
                    decrease = f_before - moved.current_objective_value
                    assert decrease <= (1e-13 if dtype == np.float64 else 1e-6) * abs(f_before) or min(opt.last_trials, ref.last_trials) > 30, (
                        ex, it, n, m, step0, decor, other, opt.last_trials, ref.last_trials, f_before, decrease, opt.is_stuck, ref.is_stuck)
                    break
                if ref.is_stuck:
                    break
                if ref.last_trials > 30 and opt.last_trials != ref.last_trials:
                    break        # dozens of halvings: f_new - f is at rounding level, the two summation orders may decide differently
                if opt.last_trials != ref.last_trials and min(abs(ref.delta_objective_value), abs(opt.delta_objective_value)) <= (1e-12 if dtype == np.float64 else 1e-5) * abs(f_before):
                    break        # a decrease at rounding level (a run at its minimiser, box constraints active): :139 may go either way for the two summation orders
                assert opt.last_trials == ref.last_trials, (ex, it, n, m, step0, decor, other, opt.last_trials, ref.last_trials, ref.delta_objective_value, opt.delta_objective_value, f_before)
                e = rel(opt.step_direction.to_host(), ref.step_direction)
                worst[dtype] = max(worst[dtype], e)
                assert e <= (1e-9 if dtype == np.float64 else 1e-3), (ex, it, n, m, step0, decor, other, e)
                ex_ = rel(opt.current_point.to_host(), ref.current_point)
                # (x_new = x + t d: the point inherits at most the direction's relative error -- seed 404, case 299, fp32, n = 16:
                # direction 2.2e-5 off, within its tolerance, and the point 8.9e-6)
                # fp64: the direction's error reaches the point scaled by |x_new - x| / |x_new|, nothing more
                moved_by = np.linalg.norm(ref.delta_point.astype(np.float64)) / max(np.linalg.norm(ref.current_point.astype(np.float64)), 1e-300)
                tol_x = max(1e-12, 2 * e * moved_by) if dtype == np.float64 else max(1e-6, 2 * e)
                assert ex_ <= tol_x, (ex, it, n, m, step0, np.dtype(dtype).name, "direction", e, "point", ex_, "trials", opt.last_trials)
                steps_total += 1
            retries += opt.single_pass_retries
            assert opt.ring_layout == 2, (ex, n, m, step0, np.dtype(dtype).name, other, decor, opt.iteration_count, opt.is_stuck)
        finally:
            orc.set_dot_mode(orc.DOT_SEQUENTIAL)
    return {"worst": {k.__name__: v for k, v in worst.items()}, "steps": steps_total, "later_passes": retries}


if __name__ == "__main__":
    print("ok:", run(int(os.environ.get("FUZZ_CASES", 60)), int(os.environ.get("FUZZ_SEED", 2468))))
