"""Dev tool: time lbfgs_point_pass_kernel on a naturally filled point ring (n = 1e7, m = 20) with parts switched
off (DZO_TUNE_SP_DEBUG: 1 no pair dots, 2 no stores).  Each mask gets a fresh optimizer: 26 normal steps, then a
few steps under the mask (the results are garbage from there on; only the first pass of each step is timed)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dzo_loader import dzo  # noqa: E402

n, m = int(os.environ.get("AB_N", 10_000_000)), int(os.environ.get("AB_K", 20))
dzo.init(0)
x0 = bench.rosenbrock_chain_x0(n, seed=5)
for mask in os.environ.get("AB_MASKS", "0,1,2,3").split(","):
    os.environ["DZO_TUNE_SP_DEBUG"] = "0"
    opt = dzo.LBFGSOptimizer(None, dzo.Problem(dzo.ROSENBROCK_CHAIN, n), None, dzo.DeviceArray.from_host(x0), 1.0, m)
    for _ in range(m + 6):
        opt.step()
    assert opt.ring_layout == int(os.environ.get("AB_LAYOUT", 2))
    os.environ["DZO_TUNE_SP_DEBUG"] = mask
    times = []
    for _ in range(int(os.environ.get("AB_ROUNDS", 5))):
        dzo.lib().dzo_lbfgs_set_stuck(opt.h, 0)
        dzo.profile_reset(); dzo.profile_enable(2)
        opt.step()
        dzo.synchronize()
        dzo.profile_enable(0)
        tab = dzo.profile_table()
        if "lbfgs_single_pass" in tab:
            times.append(tab["lbfgs_single_pass"][1] / tab["lbfgs_single_pass"][0] * 1e3)
    print("layout", opt.ring_layout, "mask", mask, "pass us:", [round(t, 1) for t in times], flush=True)
    del opt
