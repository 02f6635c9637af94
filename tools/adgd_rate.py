import sys, time; sys.path.insert(0,'/root/repo')
import bench
from dzo_loader import dzo
dzo.init(0)
n=10_000_000
x0=bench.rosenbrock_chain_x0(n, seed=5)
opt=dzo.AdGDOptimizer(None, dzo.Problem(dzo.ROSENBROCK_CHAIN, n), None, dzo.DeviceArray.from_host(x0), 1.0)
for _ in range(10): opt.step()
dzo.synchronize(); t0=time.perf_counter()
for _ in range(100): opt.step()
dzo.synchronize(); dt=time.perf_counter()-t0
print('AdGD n=1e7: %.1f step!()/s, %.3f ms/step, f=%.6e' % (100/dt, dt*10, opt.current_objective_value))
dzo.profile_reset(); dzo.profile_enable(2)
for _ in range(50): opt.step()
dzo.synchronize(); dzo.profile_enable(0)
for name, (launches, ms) in sorted(dzo.profile_table().items(), key=lambda kv: -kv[1][1]):
    print(f"  {name:32s} {launches:4d} launches  avg {ms / launches * 1e3:8.1f} us")
print("  fused steps", opt.fused_steps, "rejected first trials", opt.fused_rejections)
