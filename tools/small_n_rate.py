"""Dev tool: L-BFGS step!() rate at small and large n without event records (the chain pass -> decide -> finish)."""
import sys, time; sys.path.insert(0, '.')
import bench
from dzo_loader import dzo
dzo.init(0)
for n, steps in ((100_000, 400), (1_000_000, 300), (10_000_000, 100)):
    x0 = bench.rosenbrock_chain_x0(n, seed=5)
    opt = dzo.LBFGSOptimizer(None, dzo.Problem(dzo.ROSENBROCK_CHAIN, n), None, dzo.DeviceArray.from_host(x0), 1.0, 20)
    for _ in range(30): opt.step()
    dzo.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): opt.step()
    dzo.synchronize(); dt = time.perf_counter() - t0
    print(n, 'steps/s %.0f' % (steps / dt), 'us/step %.1f' % (dt / steps * 1e6), 'retries', opt.single_pass_retries, 'f', opt.current_objective_value, flush=True)
