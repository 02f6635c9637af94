"""Dev tool: host-driven and device-driven dense BFGS runs alternating (the sequence in which the stale summary showed)."""
import os, sys, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from dzo_loader import dzo
import fuzz_bfgs_search as fz
dzo.init(0)
tot = 0
for seed in (12, 13, 14):
    steps, term = fz.run(cases=300, seed=seed)
    tot += steps
    print("seed", seed, "ok", steps, term, flush=True)
print("unsealed first reads:", dzo.unsealed_first_reads(), "of about", 2 * tot, "waits")
