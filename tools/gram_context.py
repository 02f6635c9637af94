"""Dev tool: the same two-loop kernels (a) inside step! and (b) as back-to-back direction calls
on the history those steps left behind.  Separates "context" effects (neighbouring kernels,
dirty lines, clocks) from data / layout effects."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dzo_loader import dzo  # noqa: E402

n, m = int(os.environ.get("AB_N", 10_000_000)), int(os.environ.get("AB_K", 20))
dzo.init(0)
x0 = bench.rosenbrock_chain_x0(n, seed=5)
prob = dzo.Problem(dzo.ROSENBROCK_CHAIN, n)
x_dev = dzo.DeviceArray.from_host(x0)
opt = dzo.LBFGSOptimizer(None, prob, None, x_dev, 1.0, m)
for _ in range(m + 5):
    opt.step()


def show(tag):
    dzo.synchronize()
    dzo.profile_enable(0)
    for name, (launches, ms) in dzo.profile_table().items():
        if name in ("lbfgs_gram_pass", "lbfgs_combine"):
            print(f"{tag:26s} {name:18s} {launches:4d} launches  avg {ms / launches * 1e3:8.1f} us", flush=True)


level = int(os.environ.get("AB_LEVEL", 1))
dzo.profile_reset(); dzo.profile_enable(level)
for _ in range(30):
    opt.step()
show("inside step!")
dzo.profile_reset(); dzo.profile_enable(level)
for _ in range(30):
    opt.compute_step_direction()
show("direction back-to-back")
dzo.profile_reset(); dzo.profile_enable(level)
for _ in range(30):
    opt.step()
show("inside step! again")

import time
import numpy as np
scratch = [dzo.DeviceArray.zeros(n) for _ in range(5)]


def variant(tag, between):
    dzo.profile_reset(); dzo.profile_enable(level)
    for _ in range(30):
        between()
        opt.compute_step_direction()
    show(tag)



import time
import numpy as np
scratch = [dzo.DeviceArray.zeros(n) for _ in range(5)]


def variant(tag, between):
    dzo.profile_reset(); dzo.profile_enable(level)
    for _ in range(30):
        between()
        opt.compute_step_direction()
    show(tag)


S_hist, Y_hist = opt.delta_point_history, opt.delta_gradient_history
variant("write newest pair", lambda: (dzo.fill_(S_hist[0], 0.25), dzo.fill_(Y_hist[0], 0.5)))
variant("write pair, then read 320MB", lambda: (dzo.fill_(S_hist[0], 0.25), dzo.fill_(Y_hist[0], 0.5),
                                                 dzo.dot(scratch[0], scratch[1]), dzo.dot(scratch[2], scratch[3])))
variant("write pair, then read 800MB", lambda: (dzo.fill_(S_hist[0], 0.25), dzo.fill_(Y_hist[0], 0.5),
                                                 [dzo.dot(scratch[i], scratch[(i + 1) % 5]) for i in range(5)]))
variant("write pair, then write 400MB", lambda: (dzo.fill_(S_hist[0], 0.25), dzo.fill_(Y_hist[0], 0.5),
                                                  [dzo.fill_(w, 1.0) for w in scratch]))
