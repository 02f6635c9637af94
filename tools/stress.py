"""Dev tool: handle-lifetime and concurrency stress (no error, results as a lone run gives, free memory where it started)."""
import sys, time, threading; sys.path.insert(0, '.')
import numpy as np
import bench
from dzo_loader import dzo
dzo.init(0)
def probe_free():
    lo, hi = 0, 300 * 1024
    while hi - lo > 256:
        mid = (lo + hi) // 2
        try:
            a = dzo.DeviceArray(mid * 2**20 // 8); a.free(); lo = mid
        except Exception:
            hi = mid
    return lo
t0 = time.perf_counter()
m0 = probe_free()
n = 20_000
x0 = bench.rosenbrock_chain_x0(n, seed=3)
def lone(kind, steps):
    o = make(kind)
    for _ in range(steps): o.step()
    return o.current_point.to_host(), o.current_objective_value
def make(kind):
    if kind == "lbfgs": return dzo.LBFGSOptimizer(None, dzo.Problem(dzo.ROSENBROCK_CHAIN, n), None, dzo.DeviceArray.from_host(x0), 1.0, 7)
    if kind == "adgd": return dzo.AdGDOptimizer(None, dzo.Problem(dzo.ROSENBROCK_CHAIN, n), None, dzo.DeviceArray.from_host(x0), 0.1)
    raise ValueError(kind)
ref = {k: lone(k, 40) for k in ("lbfgs", "adgd")}
# 1. many handles alive, stepped round-robin, with looks in between
opts = [(k, make(k)) for k in ("lbfgs", "adgd") * 60]
for s in range(40):
    for i, (k, o) in enumerate(opts):
        o.step()
        if (s + i) % 17 == 0: o.current_point.to_host()
    if s % 9 == 0: dzo.synchronize()
for k, o in opts:
    x, f = o.current_point.to_host(), o.current_objective_value
    assert f == ref[k][1] and np.array_equal(x, ref[k][0]), k
print("1. 120 handles round-robin: every one ends where a lone run does", flush=True)
del opts, o
# 2. create / destroy
for i in range(600):
    k = ("lbfgs", "adgd")[i % 2]
    o = make(k)
    for _ in range(3): o.step()
    if i % 5 == 0: o.current_gradient.to_host()
    del o
print("2. 600 create / step / destroy cycles", flush=True)
# 3. threads, one optimizer each
errs = []
def worker(kind):
    try:
        dzo.init(0)
        x, f = lone(kind, 40)
        assert f == ref[kind][1] and np.array_equal(x, ref[kind][0])
    except Exception as e:
        errs.append(repr(e))
ths = [threading.Thread(target=worker, args=(("lbfgs", "adgd")[i % 2],)) for i in range(6)]
for t in ths: t.start()
for t in ths: t.join()
assert not errs, errs
print("3. six threads, one optimizer each: same results", flush=True)
dzo.synchronize()
m1 = probe_free()
print("free MiB before / after", m0, m1, "wall s", round(time.perf_counter() - t0, 1))
assert abs(m1 - m0) <= 512
