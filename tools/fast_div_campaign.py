"""Dev tool: the long comparison of fd_div against a / b on the device (profiles/r04_fast_div.log)."""
import sys, time; sys.path.insert(0, '.')
from dzo_loader import dzo
dzo.init(0)
total = 0
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    for mode in range(5):
        t0 = time.perf_counter()
        chk, bad, first = dzo.selftest_fast_div(10_000 * rnd + 77 * mode + 5, 1 << 31, mode)
        total += chk
        print(f"round {rnd} mode {mode}: checked {chk} mismatches {bad} ({time.perf_counter() - t0:.1f} s)" + (f" first {first}" if bad else ""), flush=True)
        assert bad == 0
print("total pairs checked", total, "mismatches 0")
