"""Random looks between steps (reads, pointer hand-outs, uploads of the SAME values, synchronize, gradient / delta / direction /
history reads) must not change a run: every optimizer ends bit for bit where an unobserved run does (an L-BFGS that was handed
the same values through an upload may continue on the pair ring: the same steps, to rounding).
FUZZ_CASES / FUZZ_SEED for longer runs by hand: `FUZZ_CASES=400 python3 tests/fuzz_looks.py`."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from dzo_loader import dzo


def run(cases=60, seed=1):
    dzo.init(0)
    rng = np.random.default_rng(seed)
    for case in range(cases):
        kind = ("lbfgs", "adgd")[case % 2]
        dtype = (np.float64, np.float32)[int(rng.integers(2))]
        n = int(rng.choice([64, 1000, 4099, 4100, 20_000, 100_003]))
        if kind == "adgd": n -= n % (2 if dtype == np.float64 else 4)
        m = int(rng.integers(1, 12))
        steps = int(rng.integers(5, 40))
        x0 = bench.rosenbrock_chain_x0(n, seed=int(rng.integers(1000))).astype(dtype)
        def make():
            xd = dzo.DeviceArray.from_host(x0)
            p = dzo.Problem(dzo.ROSENBROCK_CHAIN, n, dtype)
            return (dzo.LBFGSOptimizer(None, p, None, xd, 1.0, m) if kind == "lbfgs" else dzo.AdGDOptimizer(None, p, None, xd, 0.1)), xd
        a, _ = make()
        for _ in range(steps): a.step()
        want = (a.current_point.to_host(), a.current_gradient.to_host(), a.current_objective_value, a.iteration_count, a.is_stuck)
        b, xd = make()
        for s in range(steps):
            b.step()
            for _ in range(int(rng.integers(0, 3))):
                act = int(rng.integers(10 if kind == "lbfgs" else 7))
                if act == 0: b.current_point.to_host()
                elif act == 1: b.current_gradient.to_host()
                elif act == 2: b.delta_point.to_host(); b.delta_gradient.to_host()
                elif act == 3: _ = b.current_point.ptr
                elif act == 4: dzo.synchronize()
                elif act == 5: xd.upload(b.current_point.to_host())                     # the same values through the caller's handle
                elif act == 6: b.current_gradient.upload(b.current_gradient.to_host())
                elif act == 7: b.step_direction.to_host()
                elif act == 8: [h.to_host() for h in b.delta_point_history[:2]]; [h.to_host() for h in b.delta_gradient_history[:2]]
                elif act == 9: _ = (b.rho_history, b.alpha_history)
        got = (b.current_point.to_host(), b.current_gradient.to_host(), b.current_objective_value, b.iteration_count, b.is_stuck)
        ok = np.array_equal(want[0], got[0]) and np.array_equal(want[1], got[1]) and want[2:] == got[2:]
        if kind == "lbfgs" and not ok:
            # an upload makes the L-BFGS continue on the pair ring (same values: same steps, sums in another order)
            ok = want[3:] == got[3:] and np.linalg.norm(want[0].astype(np.float64) - got[0]) <= (1e-9 if dtype == np.float64 else 2e-3) * np.linalg.norm(want[0].astype(np.float64))
        assert ok, (case, kind, dtype, n, m, steps)
    return {"cases": cases}


if __name__ == "__main__":
    out = run(int(os.environ.get("FUZZ_CASES", "60")), int(os.environ.get("FUZZ_SEED", "1")))
    print("look fuzz ok:", out)
