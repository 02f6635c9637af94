"""Dev tool: per-kernel HIP-event times of the two-loop on a frozen random state, direction
calls back to back (no line search / gradient kernels in between).  Compare with the same
kernels inside step! (bench.py) and with tools/grambench.hip."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dzo_loader import dzo  # noqa: E402

n, k = int(os.environ.get("AB_N", 10_000_000)), int(os.environ.get("AB_K", 20))
dzo.init(0)
g = bench.pcg32_uniform(n, 10) - 0.5
S = np.empty((k, n)); Y = np.empty((k, n))
zero = os.environ.get("AB_ZERO") == "1"
for i in range(k):
    S[i] = 0 if zero else bench.pcg32_uniform(n, 100 + i) - 0.5
    Y[i] = 0 if zero else bench.pcg32_uniform(n, 200 + i) - 0.5 + S[i]
Sd, Yd = dzo.DeviceArray.from_host(S), dzo.DeviceArray.from_host(Y)
del S, Y
x, gd = dzo.DeviceArray.zeros(n), dzo.DeviceArray.from_host(g)
o = dzo.LBFGSOptimizer(None, lambda x_: 0.0, lambda g_, x_: None, x, 0.0, gd, 1.0, k)
o.set_history(Sd, Yd)
dirty = int(os.environ.get("AB_DIRTY", 0))      # 80-MB device-to-device writes before every direction
scratch = [dzo.DeviceArray.zeros(n) for _ in range(dirty)]
for _ in range(3):
    o.compute_step_direction()
dzo.synchronize()
dzo.profile_reset(); dzo.profile_enable(1)
for _ in range(int(os.environ.get("AB_ROUNDS", 20))):
    for w in scratch:
        dzo.copy_(w, gd)
    o.compute_step_direction()
dzo.synchronize()
dzo.profile_enable(0)
for name, (launches, ms) in dzo.profile_table().items():
    print(f"{name:28s} {launches:4d} launches  avg {ms / launches * 1e3:8.1f} us")
