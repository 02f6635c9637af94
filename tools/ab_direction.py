"""Dev tool: interleaved A/B of two-loop kernel variants in ONE process on ONE device
(cdna_hip_programming.md rule 24).  Each variant is an optimizer created under its own
DZO_TUNE_* environment; rounds are interleaved; reports median / min wall time of
compute_lbfgs_step_direction! (blocking call) on the frozen n = 10^7, k = 20 state."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dzo_loader import dzo  # noqa: E402

n, k = int(os.environ.get("AB_N", 10_000_000)), int(os.environ.get("AB_K", 20))
variants = [json.loads(a) for a in sys.argv[1:]] or [{}]
rounds = int(os.environ.get("AB_ROUNDS", 12))

dzo.init(0)
g = bench.pcg32_uniform(n, 10) - 0.5
S = np.empty((k, n)); Y = np.empty((k, n))
for i in range(k):
    S[i] = bench.pcg32_uniform(n, 100 + i) - 0.5
    Y[i] = bench.pcg32_uniform(n, 200 + i) - 0.5 + S[i]
Sd, Yd = dzo.DeviceArray.from_host(S), dzo.DeviceArray.from_host(Y)
del S, Y
opts = []
for v in variants:
    for key in list(os.environ):
        if key.startswith("DZO_TUNE_"):
            del os.environ[key]
    for key, val in v.items():
        os.environ[key] = str(val)
    x, gd = dzo.DeviceArray.zeros(n), dzo.DeviceArray.from_host(g)
    o = dzo.LBFGSOptimizer(None, lambda x_: 0.0, lambda g_, x_: None, x, 0.0, gd, 1.0, k)
    o.set_history(Sd, Yd)
    o.compute_step_direction()
    opts.append((v, o, x, gd, []))
ref = opts[0][1].step_direction.to_host()
for v, o, *_ in opts[1:]:
    err = np.linalg.norm(o.step_direction.to_host() - ref) / np.linalg.norm(ref)
    if err >= 1e-12: print('WARNING: variant differs', v, err)
for r in range(rounds):
    for v, o, x, gd, ts in opts:
        dzo.synchronize()
        t0 = time.perf_counter()
        o.compute_step_direction()
        ts.append(time.perf_counter() - t0)
for v, o, x, gd, ts in opts:
    ts = np.array(ts[2:]) * 1e6
    print(json.dumps(v), f"median {np.median(ts):.1f} us  min {ts.min():.1f} us  ->  "
          f"{(4 * k + 2) * n * 8 / np.median(ts) / 1e3:.0f} GB/s algorithmic", flush=True)
