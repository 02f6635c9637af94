"""GPU fuzz of step!() over ragged n, ring wrap-around and both two-loop modes, with per-step resync from the
oracle.  Test infrastructure (lives under tests/ because it uses oracle/).  tests/test_gpu_fuzz.py runs `run()` with a
fixed seed and 20 cases under pytest; by hand for more:  FUZZ_CASES=120 FUZZ_SEED=12345 python tests/fuzz_lbfgs.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dzo_loader import dzo  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


SIZES = [1, 2, 3, 5, 8, 63, 64, 65, 122, 124, 126, 127, 129, 248, 255, 257, 372, 1000, 4097, 7936, 7938, 65537, 100000, 200002, 200003]


def run(cases=120, seed=12345):
    rng = np.random.default_rng(seed)
    worst, steps_total = 0.0, 0
    for ex in range(cases):
        n = int(rng.choice(SIZES))
        m = int(rng.integers(1, 23)); warm = int(rng.integers(0, 45)); mode = int(rng.integers(0, 2))
        x0 = (orc.pcg_fill(n, int(rng.integers(0, 10**6))) - 0.5) * 2.0
        ref = orc.LBFGS(orc.Problem(orc.ROSENBROCK_CHAIN, n), x0.copy(), 0.5, m)
        opt = dzo.LBFGSOptimizer(None, dzo.Problem(dzo.ROSENBROCK_CHAIN, n), None, dzo.DeviceArray.from_host(x0), 0.5, m)
        opt.set_two_loop_mode(mode)
        # both take `warm` steps; after every step the device's state is replaced by the oracle's (a per-step test:
        # free-running trajectories drift), its ring having been built step by step on the device
        for it in range(warm):
            if ref.is_stuck:
                break
            opt.step(); ref.step()
            if ref.is_stuck or opt.is_stuck:
                break
            if ref.last_trials > 30 and opt.last_trials != ref.last_trials:
                break        # dozens of halvings: f_new - f is at rounding level, the two summation orders may accept one trial apart
            e = rel(opt.step_direction.to_host(), ref.step_direction)
            S, Y = ref.history_arrays()
            k = ref.history_count
            for i in (0, k - 1):                          # ring order, before the resync
                assert rel(opt.delta_point_history[i].to_host(), S[i]) <= 1e-6, (ex, it, i)
            opt.current_point.upload(ref.current_point); opt.current_gradient.upload(ref.current_gradient)
            opt.set_objective_value(ref.current_objective_value)
            opt.set_history(S, Y, ref.rho_history, iteration_count=ref.iteration_count)
            worst = max(worst, e)
            steps_total += 1
            assert e <= 1e-8, (ex, it, n, m, mode, e)
    return {"worst": worst, "steps": steps_total}


if __name__ == "__main__":
    print("ok:", run(int(os.environ.get("FUZZ_CASES", 120)), int(os.environ.get("FUZZ_SEED", 12345))))
