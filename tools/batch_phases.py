"""Dev tool: where a batched BFGS step spends its cycles (DZO_TUNE_BATCH_DEBUG=2: per-phase clock64 sums)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DZO_TUNE_BATCH_DEBUG"] = "2"
import bench
from dzo_loader import dzo
n, B = int(os.environ.get("AB_N", 256)), int(os.environ.get("AB_B", 1024))
dzo.init(0)
X0 = np.stack([bench.pcg32_uniform(n, 1000 + b) for b in range(B)])
batch = dzo.BatchedBFGS(dzo.ROSENBROCK_CHAIN, X0, 1.0)
batch.step(10, poll=False)
st = batch._p(99, (8,), np.uint64)
st.upload(np.zeros(8, np.uint64))
import time
dzo.synchronize(); t0 = time.perf_counter()
batch.step(50, poll=False)
dzo.synchronize(); el = time.perf_counter() - t0
v = st.to_host().astype(np.float64)
its = batch.iteration_count.to_host().sum()
print(f"wall per step {el / 50 * 1e6:.1f} us; per block-step cycles (100 MHz clock64?): search {v[0] / (B * 50):.0f}, decide+move {v[1] / (B * 50):.0f}, "
      f"symv {v[2] / max(v[4], 1):.0f}, update {v[3] / max(v[4], 1):.0f}; bfgs steps {v[4]:.0f} of {B * 50}")
