"""Dev tool (GPU box): whole step!() calls on the log-sum-exp objective, point ring (trial pass + dots pass) against the general
two-pass path (DZO_TUNE_LSE_POINTS=0).  The problem of config 4 is dominated by its quadratic term, so L-BFGS is done within a
handful of steps: the first `STEPS` steps from a far start are timed, per step, with kernel tables.
    python3 tools/lse_steps.py [n] [m] [f32|f64]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dzo_loader import dzo
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dt = np.float32 if (len(sys.argv) > 3 and sys.argv[3] == "f32") else np.float64
dzo.init(0)
c = (bench.pcg32_uniform(n, 6) - 0.5).astype(dt)
x0 = (5.0 * (bench.pcg32_uniform(n, 8) - 0.5)).astype(dt)
for points, events in ((1, 0), (0, 0), (1, 0), (0, 0), (1, 2), (0, 2)):
    os.environ["DZO_TUNE_LSE_POINTS"] = str(points)
    prob = dzo.Problem(dzo.LSE, n, dt, c=c, lam=1e-2)
    opt = dzo.LBFGSOptimizer(None, prob, None, dzo.DeviceArray.from_host(x0), 1.0, m)
    opt.step()
    dzo.synchronize()
    dzo.profile_reset(); dzo.profile_enable(events)
    done = 0; trials = 0; el = 0.0
    for _ in range(12):                                  # each step timed by itself; the step that ends stuck (dozens of halvings) is left out
        t0 = time.perf_counter()
        opt.step()
        dzo.synchronize()
        dt_step = time.perf_counter() - t0
        if opt.is_stuck:
            break
        done += 1; trials += opt.last_trials; el += dt_step
    print(f"points={points} events={events} layout {opt.ring_layout} n={n} m={m} {np.dtype(dt).name}: {done} steps, {done / el:.1f} step!()/s, {1e3 * el / max(done, 1):.3f} ms/step, "
          f"{trials / max(done, 1):.2f} evals/step, f = {opt.current_objective_value:.10g}", flush=True)
    dzo.profile_enable(0)
    if events:
        print("   ", {k: (v[0], round(1e3 * v[1] / v[0], 1)) for k, v in sorted(dzo.profile_table().items(), key=lambda kv: -kv[1][1]) if v[0]})
    opt.close()
