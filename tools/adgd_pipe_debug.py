"""Dev tool: AdGD pipelined vs one-round-trip-per-pass run, scalars per step and pipeline counters."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dzo_loader import dzo
from oracle import oracle as orc
dzo.init(0)
n, K = int(os.environ.get("N", 4100)), int(os.environ.get("K", 30))
step0 = float(os.environ.get("STEP0", 0.1))
x0 = orc.rosenbrock_chain_x0(n)
for rep in range(int(os.environ.get("REPS", 4))):
    out = {}
    for mode in ("1", "0", "peek"):
        os.environ["DZO_TUNE_ADGD_PIPELINE"] = "0" if mode == "0" else "1"
        opt = dzo.AdGDOptimizer(None, dzo.Problem(dzo.ROSENBROCK_CHAIN, n), None, dzo.DeviceArray.from_host(x0), step0)
        rows = []
        for it in range(K):
            opt.step()
            rows.append((opt.current_objective_value.hex(), float(opt.current_step_size).hex(), float(opt.previous_step_size).hex(),
                         opt.iteration_count, opt.is_stuck, opt.pipelined_passes, opt.pipeline_discards))
            if mode == "peek" and it % 3 == 1:
                opt.delta_point.to_host()
        out[mode] = rows
    for it in range(K):
        a, b, c = out["1"][it], out["0"][it], out["peek"][it]
        if a[:5] != b[:5] or a[:5] != c[:5]:
            print("rep", rep, "step", it, "DIFF", a, b, c)
    print("rep", rep, "final counters", out["1"][-1][5:], out["0"][-1][5:], out["peek"][-1][5:], flush=True)
