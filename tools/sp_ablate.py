"""Dev tool: time lbfgs_single_pass_kernel alone on a frozen random state (n = 1e7, k = m = 20),
optionally with parts switched off (DZO_TUNE_SP_DEBUG bit mask: 1 no pair dots, 2 no stores, 256 plain
history loads ...).  AB_MASKS="0,256" interleaves several masks in one process (the mask is read at every
step), one launch per mask and round -- also the order of the dispatches in a rocprofv3 --pmc trace."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dzo_loader import dzo  # noqa: E402

n, k = int(os.environ.get("AB_N", 10_000_000)), int(os.environ.get("AB_K", 20))
dt = np.dtype(os.environ.get("AB_DTYPE", "float64"))
dzo.init(0)
S = np.empty((k, n), dt); Y = np.empty((k, n), dt)
for i in range(k):
    S[i] = (bench.pcg32_uniform(n, 100 + i) - 0.5) * 1e-3
    Y[i] = (bench.pcg32_uniform(n, 200 + i) - 0.5) * 1e-3 + S[i]
Sd, Yd = dzo.DeviceArray.from_host(S), dzo.DeviceArray.from_host(Y)
del S, Y
prob = dzo.Problem(dzo.ROSENBROCK_CHAIN, n, dtype=dt)
x = dzo.DeviceArray.from_host(bench.rosenbrock_chain_x0(n, seed=5).astype(dt))
opt = dzo.LBFGSOptimizer(None, prob, None, x, 1.0, k)
masks = [m.strip() for m in os.environ.get("AB_MASKS", os.environ.get("DZO_TUNE_SP_DEBUG", "0")).split(",")]
times = {m: [] for m in masks}
for it in range(int(os.environ.get("AB_ROUNDS", 6))):
    for mask in masks:
        os.environ["DZO_TUNE_SP_DEBUG"] = mask
        opt.set_history(Sd, Yd, iteration_count=50)
        lib = dzo.lib()
        lib.dzo_lbfgs_set_stuck(opt.h, 0)
        opt.set_objective_value(1e300)
        dzo.profile_reset(); dzo.profile_enable(2)
        opt.step()
        dzo.synchronize()
        dzo.profile_enable(0)
        tab = dzo.profile_table()
        if "lbfgs_single_pass" in tab:
            times[mask].append(tab["lbfgs_single_pass"][1] / tab["lbfgs_single_pass"][0] * 1e3)
for mask in masks:
    print("DZO_TUNE_SP_DEBUG =", mask, str(dt), "k", k, " single pass us:", [round(t, 1) for t in times[mask]], flush=True)
