"""GPU fuzz of step!() over ragged n, ring wrap-around and both two-loop modes, with per-step resync
from the oracle.  Test infrastructure (lives under tests/ because it uses oracle/); not collected by
pytest -- run it by hand:  python tests/fuzz_lbfgs.py"""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dzo_loader import dzo
from oracle import oracle as orc
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", 12345)))
def rel(a,b): return np.linalg.norm(a-b)/max(np.linalg.norm(b),1e-300)
worst = 0
for ex in range(120):
    n = int(rng.choice([1,2,3,5,8,63,64,65,122,124,126,127,129,248,255,257,372,1000,4097,7936,7938,65537,100000,200002,200003]))
    m = int(rng.integers(1, 23)); warm = int(rng.integers(0, 45)); mode = int(rng.integers(0, 2))
    dtype = np.float64
    x0 = (orc.pcg_fill(n, int(rng.integers(0, 10**6))) - 0.5) * 2.0
    ref = orc.LBFGS(orc.Problem(orc.ROSENBROCK_CHAIN, n), x0.copy(), 0.5, m)
    opt = dzo.LBFGSOptimizer(None, dzo.Problem(dzo.ROSENBROCK_CHAIN, n), None, dzo.DeviceArray.from_host(x0), 0.5, m)
    opt.set_two_loop_mode(mode)
    # free-run both `warm` steps with per-step resync (exercises ring wrap on the device, not via set_history)
    for it in range(warm):
        if ref.is_stuck: break
        # sync only x, g, f; keep the DEVICE's own ring (built step by step), but overwrite its contents to match
        opt.step(); ref.step()
        if ref.is_stuck or opt.is_stuck: break
        if ref.last_trials > 30 and opt.last_trials != ref.last_trials:
            break        # dozens of halvings: f_new - f is at rounding level, the two summation orders may accept one trial apart
        e = rel(opt.step_direction.to_host(), ref.step_direction)
        # drift is allowed to grow in free-run; resync everything every step to keep it a per-step test
        S, Y = ref.history_arrays()
        k = ref.history_count
        # compare device ring content against oracle's BEFORE resync (ring order check)
        for i in (0, k - 1):
            assert rel(opt.delta_point_history[i].to_host(), S[i]) <= 1e-6, (ex, it, i)
        opt.current_point.upload(ref.current_point); opt.current_gradient.upload(ref.current_gradient)
        opt.set_objective_value(ref.current_objective_value)
        opt.set_history(S, Y, ref.rho_history, iteration_count=ref.iteration_count)
        worst = max(worst, e)
        assert e <= 1e-8, (ex, it, n, m, mode, e)
print("ok, worst free-run single-step direction error", worst)
