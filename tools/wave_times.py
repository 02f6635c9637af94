"""Dev tool (GPU box): start / end clock of every wave of the last point pass (DZO_TUNE_SP_DEBUG=1024 build hook)."""
import ctypes, os, sys
import numpy as np
which = os.environ.get("WT_KERNEL", "pass")     # pass (the point pass), gram, combine (the two-pass step's kernels)
if which == "pass":
    os.environ["DZO_TUNE_SP_DEBUG"] = "1024"    # (DZO_TUNE_POINT_PRIO=0 in the environment: the pass without its issue priorities)
else:
    os.environ["DZO_TUNE_SINGLE_PASS"] = "0"
    os.environ["DZO_TUNE_TP_DEBUG"] = "1" if which == "gram" else "2"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dzo_loader import dzo
from bench import rosenbrock_chain_x0
dzo.init(0)
n, m = 10_000_000, 20
x = dzo.DeviceArray.from_host(rosenbrock_chain_x0(n, seed=5))
opt = dzo.LBFGSOptimizer(None, dzo.Problem(dzo.ROSENBROCK_CHAIN, n), None, x, 1.0, m)
for _ in range(30):
    opt.step()
dzo.synchronize()
lib = dzo.lib()
cnt = 2 * 4096
buf = (ctypes.c_ulonglong * cnt)()
lib.dzo_debug_wave_times.argtypes = [ctypes.c_void_p, ctypes.c_int32]
assert lib.dzo_debug_wave_times(buf, cnt) == 0
t = np.array(buf, dtype=np.float64).reshape(-1, 2) / 100.0     # us (100 MHz clock)
t = t[t[:, 1] > 0]                                             # (waves of blocks the launch did not have)
print("kernel:", which)
t0 = t[:, 0].min()
st, en = t[:, 0] - t0, t[:, 1] - t0
print("waves", len(st), "start: min %.1f max %.1f mean %.1f" % (st.min(), st.max(), st.mean()))
print("end:   min %.1f max %.1f mean %.1f  p10 %.1f p50 %.1f p90 %.1f" % (en.min(), en.max(), en.mean(), *np.percentile(en, [10, 50, 90])))
print("lifetime mean %.1f  /  span %.1f = %.3f" % ((en - st).mean(), en.max(), (en - st).mean() / en.max()))
blk = np.arange(len(st)) // 4
for x8 in range(8):
    sel = (blk % 8) == x8
    print("xcd %d: start mean %.1f end mean %.1f max %.1f" % (x8, st[sel].mean(), en[sel].mean(), en[sel].max()))
nb = int(blk.max()) + 1
for q4 in range(4):
    sel = (blk * 4 // nb) == q4
    print("blocks quarter %d of %d: start mean %.1f end mean %.1f" % (q4, nb, st[sel].mean(), en[sel].mean()))
