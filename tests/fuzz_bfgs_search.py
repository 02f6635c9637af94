"""GPU fuzz of the dense BFGS step's two line searches: device-driven (bfgs_dev_search, the default) against host-driven
rounds on random quadratics -- sizes, dtypes, initial step lengths (long doubling chains, long halving chains), caps on
the doublings, numbers of rounds enqueued ahead.  The two are re-schedulings of one algorithm: every step must agree bit
for bit (point, gradient, objective, step type and length, evaluation count).  tests/test_gpu_fuzz.py runs `run()` with
a fixed seed under pytest; by hand for more:  FUZZ_CASES=300 FUZZ_SEED=5 python tests/fuzz_bfgs_search.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dzo_loader import dzo  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def _trajectory(A, x0, dtype, step0, max_inc, steps):
    opt = dzo.BFGSOptimizer(dzo.Problem(dzo.QUADRATIC, len(x0), A=A, dtype=dtype), None, dzo.DeviceArray.from_host(x0), step0)
    if max_inc:
        opt.set_max_increases(max_inc)
    rows = []
    for _ in range(steps):
        opt.step()
        rows.append((opt.current_point.to_host(), opt.current_gradient.to_host(), opt.current_objective_value, opt.last_step_type,
                     opt.last_step_length, opt.objective_evaluations, opt.iteration_count, opt.has_terminated))
        if opt.has_terminated:
            break
    return rows


def run(cases=40, seed=97531):
    """`cases` random problems, each run host-driven and device-driven; returns (steps compared, cases that terminated)."""
    rng = np.random.default_rng(seed)
    steps_total = terminated = 0
    saved = {k: os.environ.get(k) for k in ("DZO_TUNE_BFGS_DEV_SEARCH", "DZO_TUNE_BFGS_DEV_ROUNDS")}
    try:
        for ex in range(cases):
            dtype = np.float64 if rng.integers(0, 3) else np.float32
            n = int(rng.choice([2, 3, 8, 16, 33, 64, 100, 130, 200, 256, 513, 1024]))
            A = orc.quadratic_matrix(n).astype(dtype)
            scale = float(rng.choice([1e-3, 1.0, 1.0, 1e3]))
            x0 = ((orc.pcg_fill(n, int(rng.integers(0, 10**6))) - 0.5) * scale).astype(dtype)
            step0 = float(rng.choice([1e-9, 1e-4, 1.0, 1.0, 50.0, 1e5]))
            max_inc = int(rng.choice([0, 0, 1, 2, 5]))
            steps = int(rng.integers(3, 40))
            os.environ["DZO_TUNE_BFGS_DEV_SEARCH"] = "0"
            want = _trajectory(A, x0, dtype, step0, max_inc, steps)
            os.environ["DZO_TUNE_BFGS_DEV_SEARCH"] = "1"
            os.environ["DZO_TUNE_BFGS_DEV_ROUNDS"] = str(int(rng.choice([1, 2, 2, 3, 5])))
            got = _trajectory(A, x0, dtype, step0, max_inc, steps)
            assert len(got) == len(want), (ex, n, dtype, step0, max_inc, len(got), len(want))
            for i, (a, b) in enumerate(zip(want, got)):
                assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2:] == b[2:], (ex, n, dtype, step0, max_inc, i, a[2:], b[2:])
            steps_total += len(want)
            terminated += bool(want[-1][7])
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return steps_total, terminated


if __name__ == "__main__":
    dzo.init(0)
    n_steps, n_term = run(int(os.environ.get("FUZZ_CASES", "120")), int(os.environ.get("FUZZ_SEED", "97531")))
    print(f"bfgs search fuzz ok: {n_steps} steps compared bit for bit, {n_term} runs reached termination")
