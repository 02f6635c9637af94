"""Dev tool: does the streaming-read ceiling depend on what the buffer holds and on how large it is?"""
import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '.')
from dzo_loader import dzo
dzo.init(0)
rng = np.random.default_rng(1)
for mb in (64, 128, 192, 384, 1024):
    n = mb * (1 << 20) // 8
    row = []
    for what in ("zeros", "ones", "random"):
        if what == "zeros": a = dzo.DeviceArray.zeros(n)
        elif what == "ones": a = dzo.DeviceArray.from_host(np.ones(n))
        else: a = dzo.DeviceArray.from_host(rng.standard_normal(n))
        reps = 40 if mb <= 384 else 10
        bw = [dzo.calibrate_read_bandwidth_of(a, reps) for _ in range(3)]
        row.append(f"{what} {max(bw):7.0f}")
        del a
    print(f"{mb:5d} MiB  GB/s: " + "   ".join(row), flush=True)
