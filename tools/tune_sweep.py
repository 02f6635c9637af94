"""Dev tool: sweep the DZO_TUNE_* knobs of the two-loop kernels with bench.py (one process per
configuration) and print the per-kernel HIP-event times."""
import itertools
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
configs = []
for gu, gb in itertools.product((1, 2, 4), (4, 8)):
    configs.append({"DZO_TUNE_GRAM_U": gu, "DZO_TUNE_GRAM_BPC": gb})
for cu, cb in itertools.product((1, 2, 4), (4, 8, 16)):
    configs.append({"DZO_TUNE_COMBINE_U": cu, "DZO_TUNE_COMBINE_BPC": cb})
if len(sys.argv) > 1:
    configs = [json.loads(a) for a in sys.argv[1:]]
for cfg in configs:
    env = dict(os.environ)
    env.update({k: str(v) for k, v in cfg.items()})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "2",
                          "--no-cpu-baseline"], env=env, capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        k = d["kernels"]
        print(json.dumps(cfg), "ms/step", d["ms_per_step"], "gram", k["lbfgs_gram_pass"]["avg_us"], "combine",
              k["lbfgs_combine"]["avg_us"], "two_loop_frac", d["roofline"]["two_loop"]["frac"], flush=True)
    except Exception as e:
        print(json.dumps(cfg), "FAILED", e, out.stderr[-400:], flush=True)
