"""Dev tool: how often do the L-BFGS / AdGD decision waits find their (single-line) result unsealed?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from dzo_loader import dzo
dzo.init(0)
import fuzz_points, fuzz_lbfgs
a = fuzz_points.run(cases=150, seed=31); print("points", a["steps"], "unsealed so far", dzo.unsealed_first_reads(), flush=True)
b = fuzz_lbfgs.run(cases=150, seed=32); print("two-pass", b["steps"], "unsealed so far", dzo.unsealed_first_reads(), flush=True)
