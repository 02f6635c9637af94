"""The headline kernel at the sizes where it runs its real loop.

lbfgs_point_pass_kernel launches at most CUs x 4 wave-rows of 62 16-byte vectors: a wave takes a second row only
beyond n ~ 1.3e5 (fp64), fills its 16-row LDS staging burst beyond n ~ 2e6, and swaps its two register sets every
row.  The parity tests of tests/test_gpu_lbfgs.py stop at n = 100 004, i.e. before any of that; these cases compare
the point path with the oracle (src/DZOptimization.jl:430-451 two-loop, :454-509 step!, :107-154 backtracking) at
n = 2.5e5 ... 1.2e7, for every instantiation K in {6, 8, 10, 12, 14, 16, 18, 20, 24} x {fp32, fp64}, both arrangements of the tiles
(tile-major, stream-major) and above the 32-bit-offset switch (n = 1.2e7, m = 20: tile-major fallback).

How: installing pairs would turn the point ring into a pair ring, so the GPU optimizer runs FREE and the oracle is
given the GPU's complete state (public getters; they do not leave the point layout) right before each checked step;
both then take the step and everything the step produces is compared: step_direction (the on-demand materialize
pass) <= 1e-10 relative (fp32: 2e-4 against the wide-accumulator oracle), the new point, gradient, objective value,
delta_point / delta_gradient (exact run_and_test! equalities), the number of trials.  Checked steps: the first two
with a full history (k = m, ring wrapped) and the first step after them whose first trial is REJECTED (found by a
scout run of the same deterministic trajectory), which exercises the retry pass at scale; fp32 runs that end before
the history is full (fp32 cannot resolve the decrease of f ~ 1e8 for long) are checked on their last safe steps.  A second run with
initial_step_length = 300 checks steps 0..2 (FIRST instantiation, several halvings on the pass, k = 0, 1, 2).
"""
import numpy as np
import pytest

from dzo_loader import dzo
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

TOL_DIRECTION = 1e-10      # north_star: per-step output within 1e-10 relative of the CPU reference


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


DECOR = {}                 # decorators of the problem in the cases that set them (legacy/DZOptimization.jl:219-296)
QCHAIN = {}                # {"lam": ...}: the cases on the chained quadratic (the point pass's second objective)


def _make(n, m, dtype, step0=1.0):
    x0 = orc.rosenbrock_chain_x0(n, dtype)
    prob = dzo.Problem(dzo.QUADRATIC_CHAIN, n, dtype, **QCHAIN) if QCHAIN else dzo.Problem(dzo.ROSENBROCK_CHAIN, n, dtype, **DECOR)
    opt = dzo.LBFGSOptimizer(None, prob, None, dzo.DeviceArray.from_host(x0), step0, m)
    return x0, opt


def _state(opt, n, dtype):
    k = opt.history_count
    S = np.empty((k, n), dtype); Y = np.empty((k, n), dtype)
    hs, hy = opt.delta_point_history, opt.delta_gradient_history
    for i in range(k):
        S[i] = hs[i].to_host(); Y[i] = hy[i].to_host()
    return (opt.current_point.to_host(), opt.current_gradient.to_host(), opt.current_objective_value, S, Y,
            opt.rho_history[:k].copy(), opt.iteration_count)


def _checked_step(opt, ref, n, dtype, where):
    """One step of both from the GPU's state; returns the number of trials."""
    x, g, f, S, Y, rho, its = _state(opt, n, dtype)
    ref.install_state(x, g, f, S, Y, rho, its)
    del S, Y
    opt.step(); ref.step()
    assert opt.ring_layout == 2, where
    assert not opt.is_stuck and not ref.is_stuck, where
    assert opt.iteration_count == ref.iteration_count and opt.last_trials == ref.last_trials, (where, opt.last_trials, ref.last_trials)
    f64 = dtype == np.float64
    e = rel(opt.step_direction.to_host(), ref.step_direction)
    assert e <= (TOL_DIRECTION if f64 else 2e-4), (where, "step_direction", e)
    x1, g1 = opt.current_point.to_host(), opt.current_gradient.to_host()
    assert rel(x1, ref.current_point) <= (1e-12 if f64 else 1e-6), where
    assert rel(g1, ref.current_gradient) <= (1e-9 if f64 else 1e-3), where
    assert opt.current_objective_value == pytest.approx(ref.current_objective_value, rel=1e-12 if f64 else 1e-5), where
    # run_and_test! (legacy/DZOptimization.jl:1035-1046), exact; and the new pair is what the ring hands out
    assert np.array_equal(opt.delta_point.to_host(), x1 - x), where
    assert np.array_equal(opt.delta_gradient.to_host(), g1 - g), where
    assert np.array_equal(opt.delta_point_history[0].to_host(), x1 - x), where
    return opt.last_trials


CASES = [
    # dtype, n, m, tile arrangement forced (None = the library's choice)
    (np.float64, 400_000, 6, None),            # K = 6 (m <= 6), tile-major
    (np.float64, 400_000, 5, 1),               # K = 6, stream-major forced
    (np.float64, 400_000, 8, None),            # K = 8, tile-major (m < 9), 3 rows per wave
    (np.float64, 400_000, 8, 1),               # K = 8, stream-major forced
    (np.float64, 400_000, 10, None),           # K = 10 (the instantiation of m = 9, 10), stream-major
    (np.float64, 400_000, 9, 0),               # K = 10, tile-major forced
    (np.float64, 400_000, 12, None),           # K = 12, stream-major
    (np.float64, 400_000, 12, 0),              # K = 12, tile-major forced
    (np.float64, 400_000, 14, None),           # K = 14 (m = 13, 14)
    (np.float64, 400_000, 16, None),           # K = 16
    (np.float64, 400_000, 17, None),           # K = 18 (m = 17, 18)
    (np.float64, 400_000, 16, 0),
    (np.float64, 2_500_000, 20, None),         # K = 20, 20 rows per wave: full staging bursts
    (np.float64, 2_500_000, 20, 0),
    (np.float64, 10_000_000, 20, None),        # config 3 itself
    (np.float64, 400_000, 24, None),           # K = 24: two register sets, one wave per SIMD (m = 23, 24, fp64 only)
    (np.float64, 2_500_000, 22, 0),            # K = 22 (m = 21, 22; round 4), tile-major forced, 20 rows per wave
    (np.float64, 400_000, 21, None),           # K = 22
    (np.float32, 500_000, 6, None),            # fp32, K = 6
    (np.float32, 500_000, 7, None),            # fp32, K = 8, tile-major
    (np.float32, 500_000, 10, None),           # fp32, K = 10
    (np.float32, 500_000, 12, None),           # fp32, K = 12
    (np.float32, 1_000_000, 14, 0),            # fp32, K = 16, tile-major forced
    (np.float32, 1_000_000, 14, None),         # fp32, K = 16, stream-major
    (np.float32, 5_000_000, 20, None),         # fp32, K = 20, 20 rows per wave
    (np.float64, 12_000_000, 20, None),        # ring > 4 GiB: 32-bit stream offsets do not fit -> tile-major fallback
    # ragged n (not a multiple of the 16-byte vector): the last vector of every ring stream is padded with phantom
    # elements whose stencil coefficients are zero
    (np.float64, 400_001, 12, None),
    (np.float64, 400_001, 7, None),            # tile-major
    (np.float64, 10_000_001, 20, None),        # config 3 + 1
    (np.float32, 500_001, 10, None),
    (np.float32, 500_002, 12, 0),
    (np.float32, 1_000_003, 16, None),
]


def _scout(n, m, dtype, arrangement):
    """The longest free run among a few initial step lengths (the first that survives the window): returns
    (step0, trials per step, f at the end, steps taken).  The reference's backtracking search has no curvature condition
    and fp32 cannot resolve the decrease of an objective of size 1e8 for long, so runs may end stuck after 10-20 steps
    (e.g. n = 250 000 fp64 at step 13 on every implementation here and on the oracle; fp32 at n = 5e6 at step 11)."""
    window = m + 30
    best = None
    for step0 in (1.0, 0.5, 4.0, 0.25):
        _, scout = _make(n, m, dtype, step0=step0)
        want_layout = {None: None, 0: 1, 1: 2}[arrangement]
        if want_layout is not None:
            assert scout.tile_arrangement == want_layout
        if n == 12_000_000:
            assert scout.tile_arrangement == 1            # above the 32-bit-offset switch
        assert scout.ring_layout == 2
        trials = []
        for _ in range(window):
            scout.step()
            if scout.is_stuck:
                break
            trials.append(scout.last_trials)
        f_end = scout.current_objective_value
        assert scout.ring_layout == 2
        scout.close()
        if best is None or len(trials) > len(best[1]):
            best = (step0, trials, f_end)
        if len(trials) == window:
            break
    return best


@pytest.mark.parametrize("dtype,n,m,arrangement", CASES,
                         ids=[f"{np.dtype(d).name}-n{n}-m{m}-{'auto' if a is None else ('stream' if a else 'tile')}" for d, n, m, a in CASES])
def test_point_pass_at_scale_matches_the_oracle_step_by_step(dtype, n, m, arrangement, monkeypatch):
    if arrangement is not None:
        monkeypatch.setenv("DZO_TUNE_STREAM_MAJOR", str(arrangement))
    orc.set_threads(8)
    if dtype == np.float32:
        orc.set_dot_mode(orc.DOT_WIDE)                    # the device sums in fp64
    try:
        # ---- scout: the same (deterministic) trajectory without looking
        step0, trials, f_scout = _scout(n, m, dtype, arrangement)
        L = len(trials)
        if L >= m + 3:                                    # full history: the first two steps with k = m, then a rejected first trial
            rejected = next((i for i in range(m + 2, L - 2) if 1 < trials[i] <= 20), None)
            checked = [m, m + 1] + ([rejected] if rejected is not None else [])
        else:                                             # the run ends early: the last steps that are safely before its end
            assert L >= 9, (L, "run too short to check anything")
            rejected = None
            checked = [L - 6, L - 5, L - 4]
        if dtype == np.float64:
            assert L >= m + 3                             # every fp64 case reaches its steady state
        # ---- the checked run
        x0, opt = _make(n, m, dtype, step0=step0)
        ref = orc.LBFGS(orc.Problem(orc.ROSENBROCK_CHAIN, n, dtype), x0.copy(), step0, m)
        it = 0
        for target in checked:
            while it < target:
                opt.step(); it += 1
                assert opt.last_trials == trials[it - 1], (it, "the trajectory is not reproducible")
            t = _checked_step(opt, ref, n, dtype, (n, m, "step", target))
            assert t == trials[target]
            it += 1
        if rejected is not None:
            assert trials[rejected] >= 2 and opt.single_pass_retries >= 1
        # looking at the state must not have changed the trajectory: finish the run and compare with the scout
        if n <= 2_500_000:
            while it < L:
                opt.step(); it += 1
            assert opt.current_objective_value == f_scout
        opt.close(); ref.close()
    finally:
        orc.set_dot_mode(orc.DOT_SEQUENTIAL)
        orc.set_threads(1)


_DECOR_AT_SCALE = [
    (np.float64, 400_000, 12, dict(l2=0.01)),
    (np.float64, 400_001, 8, dict(box_gradient=(-1.1, 0.9), box_constraint=(-1.1, 0.9))),
    (np.float64, 2_500_000, 20, dict(l2=0.01, box_gradient=(-1.1, 0.9), box_constraint=(-1.1, 0.9))),
    (np.float64, 10_000_000, 20, dict(l2=0.001, box_gradient=(-1.15, 0.95), box_constraint=(-1.15, 0.95))),
    (np.float64, 400_000, 22, dict(l2=0.01, box_gradient=(-1.1, 0.9), box_constraint=(-1.1, 0.9))),
    (np.float32, 500_000, 10, dict(l2=0.01, box_gradient=(-1.1, 0.9), box_constraint=(-1.1, 0.9))),
    (np.float32, 1_000_002, 16, dict(l2=0.01, box_gradient=(-1.1, 0.9))),
]


@pytest.mark.parametrize("dtype,n,m,decor", _DECOR_AT_SCALE, ids=[f"{np.dtype(d).name}-n{n}-m{m}-{'+'.join(sorted(k))}" for d, n, m, k in _DECOR_AT_SCALE])
def test_decorated_point_pass_at_scale_matches_the_oracle_step_by_step(dtype, n, m, decor, monkeypatch):
    """The DEC instantiations of the pass (L2 term, box-gradient mask, box projection riding along) at the
    sizes where a wave takes many rows: same protocol as the undecorated cases -- scout run, then the oracle follows the
    GPU's state through the first two steps with a full history and the first rejected first trial."""
    DECOR.clear(); DECOR.update(decor)
    orc.set_threads(8)
    if dtype == np.float32:
        orc.set_dot_mode(orc.DOT_WIDE)
    try:
        step0, trials, f_scout = _scout(n, m, dtype, None)
        L = len(trials)
        assert L >= 9, (L, "run too short to check anything")
        if L >= m + 3:
            rejected = next((i for i in range(m + 2, L - 2) if 1 < trials[i] <= 20), None)
            checked = [m, m + 1] + ([rejected] if rejected is not None else [])
        else:
            checked = [L - 6, L - 5, L - 4]
        x0, opt = _make(n, m, dtype, step0=step0)
        ref_p = orc.Problem(orc.ROSENBROCK_CHAIN, n, dtype, **decor)
        ref = orc.LBFGS(ref_p, x0.copy(), step0, m)
        it = 0
        for target in checked:
            while it < target:
                opt.step(); it += 1
                assert opt.last_trials == trials[it - 1], (it, "the trajectory is not reproducible")
            t = _checked_step(opt, ref, n, dtype, (n, m, "step", target))
            assert t == trials[target]
            x = opt.current_point.to_host()
            assert np.array_equal(opt.current_gradient.to_host(), ref_p.grad(x))      # the decorated gradient, bit-exact
            if "box_constraint" in decor:
                assert x.min() >= dtype(decor["box_constraint"][0]) and x.max() <= dtype(decor["box_constraint"][1])
            it += 1
        opt.close(); ref.close()
    finally:
        DECOR.clear()
        orc.set_dot_mode(orc.DOT_SEQUENTIAL)
        orc.set_threads(1)


_QCHAIN_AT_SCALE = [(np.float64, 400_000, 12), (np.float64, 2_500_001, 20), (np.float64, 10_000_000, 20), (np.float64, 400_000, 23),
                    (np.float32, 500_000, 10), (np.float32, 1_000_003, 16)]


@pytest.mark.parametrize("dtype,n,m", _QCHAIN_AT_SCALE, ids=[f"{np.dtype(d).name}-n{n}-m{m}" for d, n, m in _QCHAIN_AT_SCALE])
def test_chained_quadratic_point_pass_at_scale_matches_the_oracle_step_by_step(dtype, n, m):
    """The point pass's second objective (ChainObj<T, 1>: the chained quadratic) at the sizes where a
    wave takes many rows: the oracle follows the GPU's state through the first two steps with a full history and two
    later ones (lambda = 1e-4: a condition number of 4e4, so the run is still far from its minimiser there)."""
    QCHAIN.clear(); QCHAIN.update(lam=1e-4)
    orc.set_threads(8)
    if dtype == np.float32:
        orc.set_dot_mode(orc.DOT_WIDE)
    try:
        x0, opt = _make(n, m, dtype)
        assert opt.ring_layout == 2
        ref_p = orc.Problem(orc.QUADRATIC_CHAIN, n, dtype, lam=1e-4)
        ref = orc.LBFGS(ref_p, x0.copy(), 1.0, m)
        it = 0
        for target in (m, m + 1, m + 7, m + 8):
            while it < target:
                opt.step(); it += 1
                assert not opt.is_stuck
            _checked_step(opt, ref, n, dtype, (n, m, "step", target))
            x = opt.current_point.to_host()
            assert np.array_equal(opt.current_gradient.to_host(), ref_p.grad(x))      # the stencil, bit-exact
            it += 1
        assert opt.single_pass_steps == it and opt.ring_layout == 2
        opt.close(); ref.close()
    finally:
        QCHAIN.clear()
        orc.set_dot_mode(orc.DOT_SEQUENTIAL)
        orc.set_threads(1)


@pytest.mark.parametrize("dtype,n,m", [(np.float64, 2_500_000, 20), (np.float32, 5_000_000, 20), (np.float64, 10_000_000, 20)],
                         ids=["float64-n2500000", "float32-n5000000", "float64-n10000000"])
def test_first_steps_with_deep_halvings_at_scale(dtype, n, m):
    """A large initial_step_length (300 at n = 4100, scaled with sqrt(n) so that the first trial's per-element move stays the
    same): the first step!() needs several halvings, all of them passes of the FIRST
    instantiation (d read from the constructor's step_direction, :463); steps 1 and 2 run the K = 20 kernel with k = 1, 2
    (the points beyond k alias point k)."""
    orc.set_threads(8)
    if dtype == np.float32:
        orc.set_dot_mode(orc.DOT_WIDE)
    try:
        step0 = float(np.float32(300.0 * np.sqrt(n / 4100.0)))
        x0, opt = _make(n, m, dtype, step0=step0)
        ref = orc.LBFGS(orc.Problem(orc.ROSENBROCK_CHAIN, n, dtype), x0.copy(), step0, m)
        # :386-387 the constructor's direction
        assert rel(opt.step_direction.to_host(), ref.step_direction) <= (1e-13 if dtype == np.float64 else 1e-6)
        seen = []
        for it in range(3):
            x, g, f, S, Y, rho, its = _state(opt, n, dtype)
            ref.install_state(x, g, f, S, Y, rho, its)
            opt.step(); ref.step()
            assert not opt.is_stuck and not ref.is_stuck
            assert opt.last_trials == ref.last_trials, (it, opt.last_trials, ref.last_trials)
            seen.append(opt.last_trials)
            if it > 0:                                    # (step 0 walks along the constructor's direction: nothing is formed)
                assert rel(opt.step_direction.to_host(), ref.step_direction) <= (TOL_DIRECTION if dtype == np.float64 else 2e-4), it
            x1 = opt.current_point.to_host()
            assert rel(x1, ref.current_point) <= (1e-12 if dtype == np.float64 else 1e-6), it
            assert np.array_equal(opt.delta_point.to_host(), x1 - x), it
            assert np.array_equal(opt.current_gradient.to_host() - g, opt.delta_gradient.to_host()), it
            assert opt.current_objective_value == pytest.approx(ref.current_objective_value, rel=1e-12 if dtype == np.float64 else 1e-5)
        assert seen[0] >= 3, seen                         # deep halvings happened, on passes
        assert opt.ring_layout == 2 and opt.single_pass_steps == 3
        opt.close(); ref.close()
    finally:
        orc.set_dot_mode(orc.DOT_SEQUENTIAL)
        orc.set_threads(1)


@pytest.mark.parametrize("dtype,n,m", [(np.float64, 400_000, 16), (np.float32, 500_000, 7), (np.float64, 400_000, 12)],
                         ids=["float64-m16", "float32-m7", "float64-m12"])
def test_one_register_set_variant_of_the_point_pass_runs_the_same_trajectory(dtype, n, m, monkeypatch):
    """DZO_TUNE_POINT_SETS=1 selects the point pass with ONE register set per wave and two waves per SIMD (K <= 16; measured
    slower than the two-set form at config-3-like sizes, kept as a knob).  Same arithmetic per element and the same
    reduction tree per wave-row, but a different grid, i.e. a different order of the per-block partial sums: the
    trajectory must agree with the default variant to reduction-order accuracy, step by step, trial counts included."""
    def run(sets):
        monkeypatch.setenv("DZO_TUNE_POINT_SETS", str(sets))
        _, opt = _make(n, m, dtype, step0=0.5)
        fs, tr = [], []
        for _ in range(m + 6):
            opt.step()
            if opt.is_stuck:
                break
            fs.append(opt.current_objective_value); tr.append(opt.last_trials)
        x = opt.current_point.to_host()
        assert opt.ring_layout == 2 and opt.single_pass_steps >= len(fs)
        opt.close()
        return fs, tr, x
    f2, t2, x2 = run(2)
    f1, t1, x1 = run(1)
    assert len(f1) == len(f2) >= m + 2 or len(f1) == len(f2)
    assert t1 == t2
    tol = 1e-9 if dtype == np.float64 else 1e-3
    assert np.allclose(f1, f2, rtol=tol, atol=0)
    assert rel(x1, x2) <= (1e-7 if dtype == np.float64 else 1e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,n,m", [(np.float64, 1_000_000, 20), (np.float64, 400_000, 8), (np.float32, 600_000, 12)])
def test_issue_priorities_and_wave_clocks_leave_the_results_alone(dtype, n, m, monkeypatch):
    """DZO_TUNE_POINT_PRIO (s_setprio by phase in the point pass) and bit 1024 of DZO_TUNE_SP_DEBUG (every wave stamps its
    start and end clock) change WHEN instructions issue, never what they compute: bit-identical trajectories; and the
    stamps are there to be read (tools/wave_times.py)."""
    import ctypes

    def run(prio, debug):
        monkeypatch.setenv("DZO_TUNE_POINT_PRIO", str(prio))
        monkeypatch.setenv("DZO_TUNE_SP_DEBUG", str(debug))
        _, opt = _make(n, m, dtype, step0=0.5)
        fs = []
        for _ in range(m + 5):
            opt.step()
            if opt.is_stuck:
                break
            fs.append((opt.current_objective_value, opt.last_trials))
        x, g = opt.current_point.to_host(), opt.current_gradient.to_host()
        assert opt.ring_layout == 2 and opt.single_pass_steps >= len(fs) - 1
        opt.close()
        return fs, x, g
    base = run(1, 0)
    assert len(base[0]) >= 3
    for prio, debug in ((0, 0), (1, 1024)):
        other = run(prio, debug)
        assert other[0] == base[0] and np.array_equal(other[1], base[1]) and np.array_equal(other[2], base[2])
    lib = dzo.lib()
    lib.dzo_debug_wave_times.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    buf = (ctypes.c_ulonglong * 64)()
    assert lib.dzo_debug_wave_times(buf, 64) == 0
    t = np.array(buf, dtype=np.uint64).reshape(-1, 2)
    assert (t[:, 1] > t[:, 0]).all()                                   # every wave of the first eight blocks: end after start
    assert lib.dzo_debug_wave_times(buf, 0) != 0 and lib.dzo_debug_wave_times(buf, 1 << 20) != 0   # count checked


@pytest.mark.parametrize("n,m,constraint", [(1_000_000, 10, False), (2_500_001, 20, False), (1_000_000, 7, True)],
                         ids=["n1000000-m10", "n2500001-m20", "n1000000-m7-box"])
def test_callback_path_at_scale_matches_the_oracle_step_by_step(n, m, constraint):
    """The reference's real API at scale: objective, gradient (and constraint) handed over as C function
    pointers -- the library's dzo_problem_*_cb, i.e. dzo_problem_eval / dzo_problem_grad behind the callback signature,
    what the Julia host's closures do -- so step!() runs its general path: Gram pass + reduce + finish + combine, trial
    kernel, callbacks (src/DZOptimization.jl:134-138, :479), accept and delta kernels on the slab ring.  Per step from the
    oracle's installed state: direction <= 1e-10, point, gradient bit-exact, objective value, trial counts."""
    orc.set_threads(8)
    try:
        decor = dict(box_gradient=(-1.1, 0.9), box_constraint=(-1.1, 0.9)) if constraint else {}
        x0 = orc.rosenbrock_chain_x0(n)
        prob = dzo.Problem(dzo.ROSENBROCK_CHAIN, n, **decor)
        ref_p = orc.Problem(orc.ROSENBROCK_CHAIN, n, **decor)
        opt = dzo.LBFGSOptimizer(None, prob.native_callbacks(with_constraint=constraint), None, dzo.DeviceArray.from_host(x0), 1.0, m)
        ref = orc.LBFGS(ref_p, x0.copy(), 1.0, m)
        assert opt.ring_layout == 0                       # callbacks: the slab ring and the two-pass kernels
        assert np.array_equal(opt.current_point.to_host(), ref.current_point)      # (:412-414 projected start when there is a constraint)
        for it in range(m + 4):
            S, Y = ref.history_arrays()
            opt.current_point.upload(ref.current_point); opt.current_gradient.upload(ref.current_gradient)
            opt.set_objective_value(ref.current_objective_value)
            opt.set_history(S, Y, ref.rho_history, iteration_count=ref.iteration_count)
            del S, Y
            x_before, g_before = ref.current_point.copy(), ref.current_gradient.copy()
            opt.step(); ref.step()
            assert not opt.is_stuck and not ref.is_stuck, it
            assert opt.last_trials == ref.last_trials and opt.iteration_count == ref.iteration_count, it
            e_d = 0.0
            if it > 0:
                e_d = rel(opt.step_direction.to_host(), ref.step_direction)
                assert e_d <= TOL_DIRECTION, it
            x = opt.current_point.to_host()
            # (x_new = x + t d: the direction's error reaches the point scaled by |x_new - x| / |x_new|)
            moved_by = np.linalg.norm(ref.delta_point) / np.linalg.norm(ref.current_point)
            assert rel(x, ref.current_point) <= max(1e-12, 2 * e_d * moved_by), it
            assert np.array_equal(opt.current_gradient.to_host(), ref_p.grad(x)), it
            # (a sum of n terms in two different orders; with half of the coordinates on the box the terms are large and alike)
            assert opt.current_objective_value == pytest.approx(ref.current_objective_value, rel=1e-11)
            # run_and_test! (legacy/DZOptimization.jl:1035-1046), exact
            assert np.array_equal(opt.delta_point.to_host(), x - x_before), it
            assert np.array_equal(opt.delta_gradient.to_host(), opt.current_gradient.to_host() - g_before), it
        assert opt.single_pass_steps == 0 and opt.ring_layout == 0
        opt.close(); ref.close()
    finally:
        orc.set_threads(1)

