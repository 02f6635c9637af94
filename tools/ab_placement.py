"""Dev tool: how much does the physical placement of the history slab matter?  Creates several
identical optimizers (same knobs), prints the slab base address and the direction time."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from dzo_loader import dzo
n, k = 10_000_000, 20
copies = int(os.environ.get("AB_COPIES", 6))
dzo.init(0)
g = bench.pcg32_uniform(n, 10) - 0.5
S = np.empty((k, n)); Y = np.empty((k, n))
for i in range(k):
    S[i] = bench.pcg32_uniform(n, 100 + i) - 0.5
    Y[i] = bench.pcg32_uniform(n, 200 + i) - 0.5 + S[i]
Sd, Yd = dzo.DeviceArray.from_host(S), dzo.DeviceArray.from_host(Y)
del S, Y
for key, val in (json.loads(sys.argv[1]) if len(sys.argv) > 1 else {}).items():
    os.environ[key] = str(val)
opts = []
pad = []
for c in range(copies):
    if os.environ.get("AB_PAD"):
        pad.append(dzo.DeviceArray(int(os.environ["AB_PAD"]) * (c + 1) // 8))   # shift the next allocation
    x, gd = dzo.DeviceArray.zeros(n), dzo.DeviceArray.from_host(g)
    o = dzo.LBFGSOptimizer(None, lambda x_: 0.0, lambda g_, x_: None, x, 0.0, gd, 1.0, k)
    o.set_history(Sd, Yd)
    o.compute_step_direction()
    opts.append((o, x, gd, []))
for r in range(10):
    for o, x, gd, ts in opts:
        dzo.synchronize(); t0 = time.perf_counter(); o.compute_step_direction(); ts.append(time.perf_counter() - t0)
for o, x, gd, ts in opts:
    s0 = o.delta_point_history[0].ptr; y0 = o.delta_gradient_history[0].ptr; d = o.step_direction.ptr
    print(f"S0={s0:#x} Y0-S0={y0 - s0:#x} g={gd.ptr:#x} d={d:#x}  median {np.median(ts[2:]) * 1e6:.1f} us", flush=True)
