"""dev: which free runs of the point ring survive m + 30 steps (the reference's search has no curvature condition)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dzo_loader import dzo
from oracle import oracle as orc
dzo.init(0)
for (dtype, n, m) in [(np.float64, 250_000, 8), (np.float64, 250_000, 12), (np.float64, 250_000, 16), (np.float64, 400_000, 8), (np.float64, 400_000, 12), (np.float64, 400_000, 16),
                      (np.float32, 500_000, 10), (np.float32, 500_000, 7), (np.float32, 1_000_000, 14), (np.float32, 5_000_000, 20), (np.float64, 12_000_000, 20)]:
    for step0 in (1.0, 0.5, 2.0, 0.25, 4.0):
        x0 = orc.rosenbrock_chain_x0(n, dtype)
        opt = dzo.LBFGSOptimizer(None, dzo.Problem(dzo.ROSENBROCK_CHAIN, n, dtype), None, dzo.DeviceArray.from_host(x0), step0, m)
        tr = []
        for it in range(m + 30):
            opt.step()
            if opt.is_stuck:
                break
            tr.append(opt.last_trials)
        print(np.dtype(dtype).name, n, m, step0, "stuck at" if opt.is_stuck else "ok", len(tr), "rejections", sum(1 for t in tr if t > 1), "max trials", max(tr), flush=True)
        opt.close()
