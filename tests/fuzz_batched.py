"""GPU fuzz of the batched dense BFGS kernel (config 5's path, the one the ranks shard): random even n across the
row-pair instantiations (n <= 256, <= 512, <= 1024) and their boundaries, random batch sizes and start steps, the chained Rosenbrock objective or a dense quadratic (shared or one matrix per instance); before
every step each instance gets its oracle's state, after it every field is compared (tests/test_gpu_bfgs_steps.py's
per-step check).  Test infrastructure (uses oracle/).  By hand:  FUZZ_CASES=60 FUZZ_SEED=3 python tests/fuzz_batched.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dzo_loader import dzo  # noqa: E402
from oracle import oracle as orc  # noqa: E402
import test_gpu_bfgs_steps as T  # noqa: E402


def run(cases=30, seed=24680):
    rng = np.random.default_rng(seed)
    steps_total = 0
    quads = 0
    types = set()
    for ex in range(cases):
        n = int(rng.choice([2, 4, 6, 30, 62, 64, 66, 126, 128, 130, 200, 254, 256, 258, 300, 510, 512, 514, 700, 1022, 1024]))
        B = int(rng.integers(1, 7 if n <= 256 else 4))
        step0 = float(rng.choice([1e-3, 1.0, 1.0, 20.0]))
        steps = int(rng.integers(3, 14 if n <= 256 else 7))
        X0 = np.stack([orc.pcg_fill(n, int(rng.integers(0, 10**6))) * float(rng.choice([1.0, 1.0, 2.0])) for _ in range(B)])
        quad = n >= 4 and rng.random() < 0.3                    # dense quadratic: one A per instance, or one shared A
        if quad:
            X0 = X0 - 0.5
            per_instance = bool(rng.random() < 0.7)
            mats = [T._instance_matrix(n, int(rng.integers(0, 50))) for _ in range(B if per_instance else 1)]
            prob = dzo.Problem(dzo.QUADRATIC, n, A=mats[0])
            batch = dzo.BatchedBFGS(prob, X0, step0, matrices=np.stack(mats)) if per_instance else dzo.BatchedBFGS(prob, X0, step0)
            refs = [orc.BFGS(orc.Problem(orc.QUADRATIC, n, A=mats[b if per_instance else 0]), X0[b].copy(), step0) for b in range(B)]
            quads += 1
        else:
            batch = dzo.BatchedBFGS(dzo.ROSENBROCK_CHAIN, X0, step0)
            refs = [orc.BFGS(orc.Problem(orc.ROSENBROCK_CHAIN, n), X0[b].copy(), step0) for b in range(B)]
        for it in range(steps):
            st = [T._oracle_state(r) for r in refs]
            batch.install_state(x=np.stack([s["x"] for s in st]), g=np.stack([s["g"] for s in st]),
                                H=np.stack([s["H"] for s in st]), d=np.stack([s["d"] for s in st]),
                                f=[s["f"] for s in st], last_step_length=[s["last_step_length"] for s in st],
                                iteration_count=[s["iteration_count"] for s in st],
                                last_step_type=[s["last_step_type"] for s in st],
                                has_terminated=[int(r.has_terminated) for r in refs],
                                dx=np.stack([s["dx"] for s in st]), dg=np.stack([s["dg"] for s in st]))
            f_before = [r.current_objective_value for r in refs]
            batch.step(1, poll=False)
            for r in refs:
                r.step()
            got = T._batch_read(batch)
            for b in range(B):
                one = {k: (v[b] if isinstance(v, np.ndarray) else v) for k, v in got.items()}
                one["has_terminated"] = bool(one["has_terminated"])
                if quad and np.linalg.norm(refs[b].current_gradient) <= 1e-13 * max(np.linalg.norm(st[b]["g"]), 1e-300):
                    continue                                    # a quadratic converged to rounding level
                T._check_step(one, refs[b], f_before[b], (ex, n, B, step0, it, b))
                types.add(refs[b].last_step_type)
            steps_total += B
    return {"instance_steps": steps_total, "quadratic_cases": quads, "step_types": sorted(int(t) for t in types)}


if __name__ == "__main__":
    dzo.init(0)
    print("ok:", run(int(os.environ.get("FUZZ_CASES", 40)), int(os.environ.get("FUZZ_SEED", 24680))))
