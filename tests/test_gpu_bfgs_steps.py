"""Per-step parity of the dense BFGS ``step!`` (legacy/DZOptimization.jl:891-994) and of the batched
mode from IDENTICAL state: before every step the oracle's complete state (x, g, H, d, f,
last_step_length, counters) is installed in the GPU optimizer through the C ABI
(dzo_bfgs_set_s / set_i + the get_ptr arrays), both sides take one step, and the results must agree
to the north-star tolerance (1e-10 relative; H 1e-12).  Free-running trajectories cannot be held to
that bar by any implementation (the searches branch on comparisons of sums whose order differs), which
is why tests/test_gpu_bfgs.py uses looser bounds there.
"""
import numpy as np
import pytest

from dzo_loader import dzo
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

TOL = 1e-10        # BASELINE.json north_star: per-step output within 1e-10 relative of the CPU reference
TOL_H = 1e-12


def rel(a, b, scale=None):
    return np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / max(
        np.linalg.norm(np.asarray(b, np.float64)) if scale is None else scale, 1e-300)


def _oracle_state(ref):
    return dict(x=ref.current_point.copy(), g=ref.current_gradient.copy(),
                H=np.ascontiguousarray(ref.approximate_inverse_hessian), d=ref.next_step_direction.copy(),
                f=ref.current_objective_value, last_step_length=ref.last_step_length,
                iteration_count=ref.iteration_count, last_step_type=ref.last_step_type,
                dx=ref.delta_point.copy(), dg=ref.delta_gradient.copy())


def _check_step(opt_state, ref, f_before, where):
    """opt_state: dict of host arrays / scalars read back from the device after the step.

    Tolerances are relative to the magnitudes that ENTER the step, as for any backward-stable
    computation: x_new = x_old - t d cancels when the step lands near the minimiser (a quadratic
    converges superlinearly to exactly 0), so its error is bounded relative to |x_old| ~ |delta_point|,
    not |x_new|; likewise g_new against |delta_gradient|, f_new against f_old, and d = H g_new against
    |H| (|g_new| + |delta_gradient|) (tolerances relative to sum |a_i b_i|)."""
    assert opt_state["iteration_count"] == ref.iteration_count, where
    assert opt_state["last_step_type"] == ref.last_step_type, where
    assert opt_state["has_terminated"] == ref.has_terminated, where
    x_scale = max(np.linalg.norm(ref.current_point), np.linalg.norm(ref.delta_point))
    assert rel(opt_state["x"], ref.current_point, x_scale) <= TOL, where
    assert abs(opt_state["f"] - ref.current_objective_value) <= TOL * max(abs(f_before), abs(ref.current_objective_value)), where
    if ref.has_terminated:
        return
    assert abs(opt_state["last_step_length"] - ref.last_step_length) <= TOL * max(abs(ref.last_step_length), 1e-300), where
    assert rel(opt_state["dx"], ref.delta_point) <= TOL, where
    g_scale = max(np.linalg.norm(ref.current_gradient), np.linalg.norm(ref.delta_gradient))
    assert rel(opt_state["g"], ref.current_gradient, g_scale) <= TOL, where
    assert rel(opt_state["dg"], ref.delta_gradient, g_scale) <= TOL, where
    H = opt_state["H"]
    assert np.array_equal(H, H.T), where                                   # exact symmetry survives the step
    Href = np.ascontiguousarray(ref.approximate_inverse_hessian)
    assert rel(H, Href) <= TOL_H, where
    d_scale = max(np.linalg.norm(ref.next_step_direction),
                  np.linalg.norm(np.abs(Href) @ (np.abs(ref.current_gradient) + np.abs(ref.delta_gradient))))
    assert rel(opt_state["d"], ref.next_step_direction, d_scale) <= TOL, where


def _read(opt):
    n = opt.n
    return dict(x=opt.current_point.to_host(), g=opt.current_gradient.to_host(), dx=opt.delta_point.to_host(),
                dg=opt.delta_gradient.to_host(), d=opt.next_step_direction.to_host(),
                H=opt.approximate_inverse_hessian.to_host().reshape(n, n), f=opt.current_objective_value,
                last_step_length=opt.last_step_length, iteration_count=opt.iteration_count,
                last_step_type=opt.last_step_type, has_terminated=opt.has_terminated)


@pytest.mark.parametrize("n,steps", [(2, 40), (16, 40), (200, 30), (4096, 10)])
def test_each_bfgs_step_matches_oracle_on_identical_state_quadratic(n, steps):
    """BASELINE config 2 (dense quadratic, up to the full n = 4096: H = 128 MiB)."""
    A = orc.quadratic_matrix(n)
    x0 = orc.pcg_fill(n, 4) - 0.5
    ref_p = orc.Problem(orc.QUADRATIC, n, A=A)
    ref = orc.BFGS(ref_p, x0, 1.0)
    opt = dzo.BFGSOptimizer(dzo.Problem(dzo.QUADRATIC, n, A=A), None, dzo.DeviceArray.from_host(x0), 1.0)
    orc.set_threads(8 if n >= 1024 else 1)
    g0 = np.linalg.norm(ref.current_gradient)
    try:
        types = set()
        for it in range(steps):
            if np.linalg.norm(ref.current_gradient) <= 1e-13 * g0:
                break        # converged to rounding level (a quadratic is solved exactly): the next steps divide 0 by 0
            opt.install_state(**_oracle_state(ref))
            f_before = ref.current_objective_value
            opt.step(); ref.step()
            _check_step(_read(opt), ref, f_before, (n, it))
            types.add(ref.last_step_type)
            if ref.has_terminated:
                break
        assert dzo.STEP_BFGS in types and it >= min(steps - 1, n, 8)   # the rank-2 update + fused direction ran (BFGS solves an n-dim quadratic in about n steps)
    finally:
        orc.set_threads(1)


@pytest.mark.parametrize("n,steps", [(2, 20), (16, 30), (200, 30), (514, 20), (1026, 8)])
def test_lower_triangle_update_path_matches_oracle_per_step(n, steps, monkeypatch):
    """The step!'s lower-triangle form of update_inverse_hessian! + mul! (default for even n >= 1024, forced
    here at small and ragged sizes: n = 514 is two panels and a 17th window with two columns) from the
    oracle's state each step; the matrix handed out is whole and exactly symmetric."""
    monkeypatch.setenv("DZO_TUNE_BFGS_TRI_MIN_N", "2")
    A = orc.quadratic_matrix(n)
    x0 = orc.pcg_fill(n, 4) - 0.5
    ref = orc.BFGS(orc.Problem(orc.QUADRATIC, n, A=A), x0, 1.0)
    opt = dzo.BFGSOptimizer(dzo.Problem(dzo.QUADRATIC, n, A=A), None, dzo.DeviceArray.from_host(x0), 1.0)
    g0 = np.linalg.norm(ref.current_gradient)
    done = 0
    for it in range(steps):
        if np.linalg.norm(ref.current_gradient) <= 1e-13 * g0 or ref.has_terminated:
            break
        opt.install_state(**_oracle_state(ref))
        f_before = ref.current_objective_value
        opt.step(); ref.step()
        _check_step(_read(opt), ref, f_before, (n, it))
        done += 1
    assert done >= min(steps, n, 8)
    # free run: two updates in a row without the host looking at H in between (the upper triangle stays stale)
    opt2 = dzo.BFGSOptimizer(dzo.Problem(dzo.QUADRATIC, n, A=A), None, dzo.DeviceArray.from_host(x0), 1.0)
    ref2 = orc.BFGS(orc.Problem(orc.QUADRATIC, n, A=A), x0, 1.0)
    for _ in range(min(6, n)):
        opt2.step(); ref2.step()
    H = opt2.approximate_inverse_hessian.to_host().reshape(n, n)
    assert np.array_equal(H, H.T)
    assert rel(H, np.ascontiguousarray(ref2.approximate_inverse_hessian)) <= 1e-7


@pytest.mark.parametrize("n,steps", [(2, 60), (16, 60), (200, 40)])
def test_each_bfgs_step_matches_oracle_on_identical_state_rosenbrock(n, steps):
    """Non-quadratic objective: the sequential (not side-by-side) line searches, gradient-descent
    steps with the H <- I reset (:962-986) and BFGS steps all occur."""
    x0 = orc.pcg_fill(n, 1000 + n)
    ref_p = orc.Problem(orc.ROSENBROCK_CHAIN, n)
    ref = orc.BFGS(ref_p, x0, 1.0)
    opt = dzo.BFGSOptimizer(dzo.Problem(dzo.ROSENBROCK_CHAIN, n), None, dzo.DeviceArray.from_host(x0), 1.0)
    types = set()
    for it in range(steps):
        opt.install_state(**_oracle_state(ref))
        f_before = ref.current_objective_value
        opt.step(); ref.step()
        _check_step(_read(opt), ref, f_before, (n, it))
        types.add(ref.last_step_type)
        if ref.has_terminated:
            break
    assert dzo.STEP_BFGS in types


def test_bfgs_state_setters_round_trip_and_checkpoint_resume():
    """README.md:11 "save/load data in the middle of optimization": a run checkpointed after 7 steps and
    resumed in a NEW optimizer continues bit for bit."""
    n = 96
    A = orc.quadratic_matrix(n)
    x0 = orc.pcg_fill(n, 4) - 0.5
    a = dzo.BFGSOptimizer(dzo.Problem(dzo.QUADRATIC, n, A=A), None, dzo.DeviceArray.from_host(x0), 1.0)
    for _ in range(7):
        a.step()
    snap = _read(a)
    b = dzo.BFGSOptimizer(dzo.Problem(dzo.QUADRATIC, n, A=A), None, dzo.DeviceArray.from_host(x0), 1.0)
    b.install_state(snap["x"], snap["g"], snap["H"], snap["d"], snap["f"], snap["last_step_length"],
                    snap["iteration_count"], snap["last_step_type"], snap["dx"], snap["dg"])
    assert b.iteration_count == 7 and b.last_step_type == snap["last_step_type"]
    assert b.current_objective_value == snap["f"] and b.last_step_length == snap["last_step_length"]
    for _ in range(6):
        a.step(); b.step()
        ra, rb = _read(a), _read(b)
        for key in ("x", "g", "dx", "dg", "d", "H"):
            assert np.array_equal(ra[key], rb[key]), key
        assert ra["f"] == rb["f"] and ra["last_step_length"] == rb["last_step_length"]
    with pytest.raises(AssertionError):                                  # :773 @assert !isnan(f)
        dzo._check(dzo.lib().dzo_bfgs_set_s(b.h, 0, float("nan")))
    with pytest.raises(dzo.DzoError):
        dzo._check(dzo.lib().dzo_bfgs_set_i(b.h, 3, 7))                  # not a StepType (:727-731)
    dzo._check(dzo.lib().dzo_bfgs_set_i(b.h, 0, 1))
    its = b.iteration_count
    b.step()
    assert b.has_terminated and b.iteration_count == its                 # :893 terminated optimizers never move


# ------------------------------------------------------------------------------ batched (K11)
def _batch_read(batch):
    B, n = batch.batch, batch.n
    H = batch.approximate_inverse_hessian.to_host().reshape(B, n, n)
    return dict(x=batch.current_point.to_host(), g=batch.current_gradient.to_host(), H=H,
                d=batch.next_step_direction.to_host(), dx=batch.delta_point.to_host(), dg=batch.delta_gradient.to_host(),
                f=batch.current_objective_value.to_host(), last_step_length=batch.last_step_length.to_host(),
                iteration_count=batch.iteration_count.to_host(), last_step_type=batch.last_step_type.to_host(),
                has_terminated=batch.has_terminated.to_host())


@pytest.mark.parametrize("n,B,steps", [(2, 6, 30), (16, 5, 30), (256, 4, 12), (512, 3, 8), (1024, 2, 6)])
def test_each_batched_step_matches_per_instance_oracle_on_identical_state(n, B, steps):
    """Config 5's kernel, one oracle per instance; n = 512 / 1024 take the two- and four-row-pair
    instantiations with more than 48 KiB of dynamic LDS."""
    X0 = np.stack([orc.pcg_fill(n, 1000 + b) for b in range(B)])
    batch = dzo.BatchedBFGS(dzo.ROSENBROCK_CHAIN, X0, 1.0)
    refs = [orc.BFGS(orc.Problem(orc.ROSENBROCK_CHAIN, n), X0[b].copy(), 1.0) for b in range(B)]
    types = set()
    for it in range(steps):
        st = [_oracle_state(r) for r in refs]
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
        got = _batch_read(batch)
        for b in range(B):
            one = {k: (v[b] if isinstance(v, np.ndarray) else v) for k, v in got.items()}
            one["has_terminated"] = bool(one["has_terminated"])
            _check_step(one, refs[b], f_before[b], (n, it, b))
            types.add(refs[b].last_step_type)
    assert dzo.STEP_BFGS in types


def _instance_matrix(n, b, r=8):
    """A_b = D_b + U_b U_b'/r (SPD, symmetric bit for bit), a different one per instance."""
    dvec = 1.0 + (9.0 + 30.0 * b) * orc.pcg_fill(n, 20 + b)
    U = (orc.pcg_fill(n * r, 40 + b) - 0.5).reshape(n, r, order="F")
    A = (U @ U.T) / r
    A[np.diag_indices(n)] += dvec
    return np.asfortranarray(0.5 * (A + A.T))


@pytest.mark.parametrize("case", ["quadratic", "rosen_l2", "rosen_box", "rosen_max_increases", "quadratic_all",
                                  "quadratic_per_instance", "quadratic_per_instance_all"])
def test_batched_objectives_and_decorators_match_per_instance_oracle(case):
    """Batched breadth: the dense quadratic with a shared A or with one A per instance ("run multiple optimizers
    in parallel", README.md:12, each on its own objective), the L2 / box-gradient / box-constraint
    decorators (legacy/DZOptimization.jl:219-296) and QuadraticLineSearch.max_increases (:181-188),
    each step from the oracle's state."""
    n, B, steps = 64, 4, 14
    kw, max_inc, mats = {}, 0, None
    if case.startswith("quadratic"):
        A = orc.quadratic_matrix(n)
        kind_o, kind_d, kw = orc.QUADRATIC, dzo.QUADRATIC, dict(A=A)
        X0 = np.stack([orc.pcg_fill(n, 4 + b) - 0.5 for b in range(B)])
        if "per_instance" in case:
            mats = [_instance_matrix(n, b) for b in range(B)]
    else:
        kind_o, kind_d = orc.ROSENBROCK_CHAIN, dzo.ROSENBROCK_CHAIN
        X0 = np.stack([orc.pcg_fill(n, 1000 + b) for b in range(B)])
    if case == "rosen_l2":
        kw.update(l2=0.05)
    if case == "rosen_box":
        kw.update(box_constraint=(0.05, 0.9), box_gradient=(0.05, 0.9))
    if case in ("quadratic_all", "quadratic_per_instance_all"):
        kw.update(l2=0.01, box_constraint=(-0.4, 0.3), box_gradient=(-0.4, 0.3))
    if case == "rosen_max_increases":
        max_inc = 1
    prob_d = dzo.Problem(kind_d, n, **kw)
    if mats is None:
        batch = dzo.BatchedBFGS(prob_d, X0, 1.0)
        refs = [orc.BFGS(orc.Problem(kind_o, n, **kw), X0[b].copy(), 1.0) for b in range(B)]
    else:                                                     # the handle's own A is NOT what the instances minimise
        batch = dzo.BatchedBFGS(prob_d, X0, 1.0, matrices=np.stack(mats))
        refs = [orc.BFGS(orc.Problem(kind_o, n, **dict(kw, A=mats[b])), X0[b].copy(), 1.0) for b in range(B)]
        f0 = batch.current_objective_value.to_host()
        assert len(set(np.round(f0 / np.abs(f0).max(), 6))) == B      # really B different objectives
    if max_inc:
        batch.set_max_increases(max_inc)
        for r in refs:
            r.set_max_increases(max_inc)
    for b in range(B):                                        # constructor: constraint on x0, f0, g0 with the decorators
        assert np.array_equal(batch.current_point.to_host()[b], refs[b].current_point)
        assert batch.current_objective_value.to_host()[b] == pytest.approx(refs[b].current_objective_value, rel=1e-12)
        assert rel(batch.current_gradient.to_host()[b], refs[b].current_gradient) <= 1e-13
    moved = 0
    for it in range(steps):
        st = [_oracle_state(r) for r in refs]
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
        got = _batch_read(batch)
        for b in range(B):
            one = {k: (v[b] if isinstance(v, np.ndarray) else v) for k, v in got.items()}
            one["has_terminated"] = bool(one["has_terminated"])
            if np.linalg.norm(refs[b].current_gradient) <= 1e-13 * max(np.linalg.norm(st[b]["g"]), 1e-300):
                continue                                      # converged to rounding level
            _check_step(one, refs[b], f_before[b], (case, it, b))
            moved += int(not refs[b].has_terminated)
    assert moved >= steps                                     # the comparison really covered moving instances
    if "box" in case or case.endswith("_all"):
        lo, hi = kw["box_constraint"]
        X = batch.current_point.to_host()
        assert X.min() >= lo and X.max() <= hi


def test_per_instance_matrices_constructor_rejects_what_it_cannot_run():
    """dzo_bfgs_batch_create_problem_matrices: quadratic handles only, a stride that holds a matrix, device memory
    (the backend assert of src/DZOptimization.jl:363-364 applies to the matrices as to every other array)."""
    import ctypes as C
    n, B = 8, 3
    X0 = dzo.DeviceArray.from_host(np.stack([orc.pcg_fill(n, b) for b in range(B)]))
    mats = np.stack([_instance_matrix(n, b) for b in range(B)])
    mats_dev = dzo.DeviceArray.from_host(mats)
    quad = dzo.Problem(dzo.QUADRATIC, n, A=orc.quadratic_matrix(n))
    h = C.c_void_p()
    with pytest.raises(dzo.DzoError):                                       # not a quadratic
        dzo._check(dzo.lib().dzo_bfgs_batch_create_problem_matrices(dzo.Problem(dzo.ROSENBROCK_CHAIN, n).h, B, mats_dev.ptr, n * n,
                                                                    X0.ptr, 1.0, -1, C.byref(h)))
    with pytest.raises(dzo.DzoError):                                       # stride shorter than a matrix
        dzo._check(dzo.lib().dzo_bfgs_batch_create_problem_matrices(quad.h, B, mats_dev.ptr, n * n - 1, X0.ptr, 1.0, -1, C.byref(h)))
    with pytest.raises(AssertionError):                                     # host memory
        dzo._check(dzo.lib().dzo_bfgs_batch_create_problem_matrices(quad.h, B, mats.ctypes.data, n * n, X0.ptr, 1.0, -1, C.byref(h)))
    # a padded stride: the same run as the dense array
    pad = np.zeros((B, n * n + 5))
    pad[:, :n * n] = mats.reshape(B, -1)
    pad_dev = dzo.DeviceArray.from_host(pad)
    dzo._check(dzo.lib().dzo_bfgs_batch_create_problem_matrices(quad.h, B, pad_dev.ptr, n * n + 5, X0.ptr, 1.0, -1, C.byref(h)))
    a = dzo.BatchedBFGS(quad, X0.to_host(), 1.0, matrices=mats)
    a.step(5, poll=False)
    dzo._check(dzo.lib().dzo_bfgs_batch_step(h, 5, None))
    p = C.c_void_p()
    dzo._check(dzo.lib().dzo_bfgs_batch_get_ptr(h, 0, C.byref(p)))
    xb = dzo.DeviceArray((B, n), np.float64, ptr=p.value, owner=False).to_host()
    assert np.array_equal(xb, a.current_point.to_host())
    dzo._check(dzo.lib().dzo_bfgs_batch_destroy(h))


def test_dense_step_on_the_lower_triangle_of_A_matches_the_oracle(monkeypatch):
    """DZO_TUNE_QUAD_TRI=1 (off by default: measured slower, csrc/dzo_problems.hip): every evaluation of the dense quadratic
    -- objective, gradient and the line searches' rounds of up to six trial points -- reads the LOWER triangle of the
    symmetric A only (quadratic_tri6_kernel).  Per step from the oracle's installed state, as the default path is tested."""
    monkeypatch.setenv("DZO_TUNE_QUAD_TRI", "1")
    n = 1024
    rng = np.random.default_rng(5)
    U = rng.standard_normal((n, 8))
    A = np.diag(1.0 + 99.0 * rng.random(n)) + U @ U.T / 8
    A = 0.5 * (A + A.T)
    x0 = rng.random(n) - 0.5
    ref = orc.BFGS(orc.Problem(orc.QUADRATIC, n, A=A), x0.copy(), 1.0)
    prob = dzo.Problem(dzo.QUADRATIC, n, A=A)
    opt = dzo.BFGSOptimizer(prob, None, dzo.DeviceArray.from_host(x0), 1.0)
    dx = dzo.DeviceArray.from_host(x0)
    assert abs(prob(dx) - ref.current_objective_value) <= 1e-12 * abs(ref.current_objective_value)
    g = prob.gradient_(dzo.DeviceArray(n, np.float64), dx).to_host()
    assert np.linalg.norm(g - ref.current_gradient) <= 1e-12 * np.linalg.norm(ref.current_gradient)
    for it in range(8):
        opt.step(); ref.step()
        assert opt.last_step_type == ref.last_step_type and opt.iteration_count == ref.iteration_count, it
        x = opt.current_point.to_host()
        assert np.linalg.norm(x - ref.current_point) <= 1e-9 * np.linalg.norm(ref.current_point), it
        assert opt.current_objective_value == pytest.approx(ref.current_objective_value, rel=1e-9)
        assert np.array_equal(opt.current_gradient.to_host(), prob.gradient_(dzo.DeviceArray(n, np.float64), opt.current_point).to_host())   # run_and_test! :1025-1032


@pytest.mark.parametrize("n,dtype", [(1024, np.float64), (1030, np.float64), (772, np.float32), (96, np.float64)])
def test_search_rounds_over_several_columns_at_once_are_the_same_bits(n, dtype, monkeypatch):
    """quadratic_phi6_cols_kernel (default: two columns of A per block, three register sets of requests in flight, every
    request unconditional) keeps the one-column kernel's assignment of elements to threads and its order of sums, so a run
    with DZO_TUNE_PHI6_COLS = 1 / 2 / 22 / 4 is the same run bit for bit -- points, gradients (the search's by-product against a
    separate evaluation, run_and_test! legacy :1025-1032), step types and evaluation counts.  n = 1030 / 772: a last iteration
    that only some threads have, a last group with a column past the end."""
    rng = np.random.default_rng(7)
    U = rng.standard_normal((n, 6))
    A = np.diag(1.0 + 49.0 * rng.random(n)) + U @ U.T / 6
    A = (0.5 * (A + A.T)).astype(dtype)
    x0 = (rng.random(n) - 0.5).astype(dtype)
    runs = []
    for cols in ("1", "2", "22", "4"):
        monkeypatch.setenv("DZO_TUNE_PHI6_COLS", cols)
        prob = dzo.Problem(dzo.QUADRATIC, n, dtype=dtype, A=A)
        opt = dzo.BFGSOptimizer(prob, None, dzo.DeviceArray.from_host(x0), 1.0)
        trace = []
        for it in range(10):
            opt.step()
            x = opt.current_point.to_host()
            g = opt.current_gradient.to_host()
            assert np.array_equal(g, prob.gradient_(dzo.DeviceArray(n, dtype), opt.current_point).to_host())
            trace.append((x, g, opt.current_objective_value, opt.last_step_type, opt.objective_evaluations))
            if opt.has_terminated:
                break
        runs.append(trace)
    for other in runs[1:]:
        assert len(other) == len(runs[0])
        for a, b in zip(runs[0], other):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2:] == b[2:]
