"""Dev tool: long free runs of step!() (single-pass vs two-pass path) -- sanity of the shipped default
on many consecutive steps; prints f after fixed step counts."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dzo_loader import dzo  # noqa: E402

dzo.init(0)
n, m = int(os.environ.get("AB_N", 100_000)), int(os.environ.get("AB_K", 20))
dt = np.dtype(os.environ.get("AB_DTYPE", "float64"))
for single in ("1", "0"):
    os.environ["DZO_TUNE_SINGLE_PASS"] = single
    x0 = bench.rosenbrock_chain_x0(n, seed=5).astype(dt)
    opt = dzo.LBFGSOptimizer(None, dzo.Problem(dzo.ROSENBROCK_CHAIN, n, dtype=dt), None, dzo.DeviceArray.from_host(x0), 1.0, m)
    t0 = time.time()
    marks = {}
    for it in range(1, int(os.environ.get("AB_STEPS", 20000)) + 1):
        opt.step()
        if opt.is_stuck:
            break
        if it in (100, 1000, 5000, 20000):
            marks[it] = opt.current_objective_value
    print("single_pass", single, "n", n, "steps", it, "stuck", opt.is_stuck, "f", opt.current_objective_value, marks,
          "single", opt.single_pass_steps, "rej", opt.single_pass_rejections, "sec %.2f" % (time.time() - t0), str(dt), flush=True)
