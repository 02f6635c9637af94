"""Dev tool: long runs of each optimizer on one handle (no error, finite state, the device's free memory where it started)."""
import sys, time; sys.path.insert(0, '.')
import numpy as np
import bench
from dzo_loader import dzo
dzo.init(0)
def free_mb():
    # free device memory as a large probe allocation sees it (MiB, to the nearest GiB step that still succeeds)
    lo, hi = 0, 300 * 1024
    while hi - lo > 256:
        mid = (lo + hi) // 2
        try:
            a = dzo.DeviceArray(mid * 2**20 // 8)
            a.free(); lo = mid
        except Exception:
            hi = mid
    return lo
m0 = free_mb()
n = 1_000_000
x0 = bench.rosenbrock_chain_x0(n, seed=7)
t0 = time.perf_counter()
opt = dzo.LBFGSOptimizer(None, dzo.Problem(dzo.ROSENBROCK_CHAIN, n), None, dzo.DeviceArray.from_host(x0), 1.0, 20)
for i in range(20000):
    opt.step()
    if i % 5000 == 4999: print("lbfgs", i + 1, opt.current_objective_value, opt.is_stuck, opt.ring_layout, flush=True)
    if opt.is_stuck: break
assert np.isfinite(opt.current_objective_value)
x = opt.current_point.to_host(); assert np.all(np.isfinite(x))
del opt
opt = dzo.AdGDOptimizer(None, dzo.Problem(dzo.ROSENBROCK_CHAIN, n), None, dzo.DeviceArray.from_host(x0), 0.1)
for i in range(60000):
    opt.step()
    if i % 20000 == 19999: print("adgd", i + 1, opt.current_objective_value, opt.is_stuck, opt.pipelined_passes, opt.pipeline_discards, flush=True)
    if opt.is_stuck: break
assert np.isfinite(opt.current_objective_value)
del opt
nb = 1024
A = bench.quadratic_matrix(nb)
opt = dzo.BFGSOptimizer(dzo.Problem(dzo.QUADRATIC, nb, A=A), None, dzo.DeviceArray.from_host(bench.pcg32_uniform(nb, 4) - 0.5), 1.0)
for i in range(5000):
    opt.step()
    if opt.has_terminated: break
print("bfgs steps", i + 1, opt.current_objective_value, opt.has_terminated, flush=True)
del opt
dzo.synchronize()
print("free MiB before / after", round(m0), round(free_mb()), "wall s", round(time.perf_counter() - t0, 1))
