import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dzo_loader import dzo
from oracle import oracle as orc
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import fuzz_lbfgs
rng = np.random.default_rng(12345)
for ex in range(20):
    n = int(rng.choice(fuzz_lbfgs.SIZES))
    m = int(rng.integers(1, 23)); warm = int(rng.integers(0, 45)); mode = int(rng.integers(0, 2))
    x0 = (orc.pcg_fill(n, int(rng.integers(0, 10**6))) - 0.5) * 2.0
    print("case", ex, "n", n, "m", m, "warm", warm, "mode", mode, flush=True)
    ref = orc.LBFGS(orc.Problem(orc.ROSENBROCK_CHAIN, n), x0.copy(), 0.5, m)
    opt = dzo.LBFGSOptimizer(None, dzo.Problem(dzo.ROSENBROCK_CHAIN, n), None, dzo.DeviceArray.from_host(x0), 0.5, m)
    print("  layout", opt.ring_layout, flush=True)
    opt.set_two_loop_mode(mode)
    for it in range(warm):
        if ref.is_stuck: break
        print("  step", it, "layout", opt.ring_layout, "k", opt.history_count, flush=True)
        opt.step(); ref.step()
        if ref.is_stuck or opt.is_stuck: break
        S, Y = ref.history_arrays()
        opt.current_point.upload(ref.current_point); opt.current_gradient.upload(ref.current_gradient)
        opt.set_objective_value(ref.current_objective_value)
        opt.set_history(S, Y, ref.rho_history, iteration_count=ref.iteration_count)
print("done")
