"""GPU parity tests for dense BFGS (config 1, config 2), AdGD and the batched mode (config 5),
through the C ABI.  The rank-2 update is evaluated in the reference's rounding order, so H is
compared tightly; H stays EXACTLY symmetric (checked bitwise)."""
import numpy as np
import pytest

from dzo_loader import dzo
from oracle import mp_twoloop, oracle as orc

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _spd(n, seed):
    rng = np.random.default_rng(seed)
    M = rng.standard_normal((n, n))
    H = M @ M.T / n + np.eye(n)
    return 0.5 * (H + H.T)


# ------------------------------------------------------------------------------ K8 / K9
@pytest.mark.parametrize("n", [2, 3, 8, 63, 64, 257, 1024])
def test_update_inverse_hessian_matches_oracle(n):
    rng = np.random.default_rng(n)
    H0 = _spd(n, n)
    d, y, g = rng.standard_normal(n), rng.standard_normal(n), rng.standard_normal(n)
    lam = -0.37 if d @ y < 0 else 0.37
    H_ref, d_ref = np.asfortranarray(H0.copy()), d.copy()
    t_ref = orc.bfgs_update(H_ref, lam, d_ref, y.copy())
    Hd, dd, yd, gd = (dzo.DeviceArray.from_host(a) for a in (H0, d, y, g))   # H0 symmetric: layouts coincide
    scratch, dnext = dzo.DeviceArray(n), dzo.DeviceArray(n)
    dzo.update_inverse_hessian_(Hd, lam, dd, yd, scratch, gd, dnext)
    H_gpu = Hd.to_host()
    assert np.array_equal(H_gpu, H_gpu.T)                           # exact symmetry
    assert rel(H_gpu, np.ascontiguousarray(H_ref)) <= 1e-13
    assert rel(dd.to_host(), d_ref) <= 1e-14                        # :874 side effect
    assert rel(scratch.to_host(), t_ref) <= 1e-13                   # t = H*dg
    assert rel(dnext.to_host(), H_gpu @ g) <= 1e-13                 # fused next direction (:958-960)
    s = lam * d
    assert np.allclose(H_gpu @ y, s, rtol=1e-9, atol=1e-11)         # secant equation
    # symv on its own
    out = dzo.symv_(dzo.DeviceArray(n), Hd, gd).to_host()
    assert rel(out, H_gpu @ g) <= 1e-13


def test_update_against_mpmath_arbiter():
    n = 12
    rng = np.random.default_rng(5)
    H0 = _spd(n, 5)
    d, y = rng.standard_normal(n), rng.standard_normal(n)
    lam = 0.5 if d @ y > 0 else -0.5
    H_mp = mp_twoloop.bfgs_update(H0.copy(), lam, d, y)
    Hd = dzo.DeviceArray.from_host(H0)
    dzo.update_inverse_hessian_(Hd, lam, dzo.DeviceArray.from_host(d), dzo.DeviceArray.from_host(y), dzo.DeviceArray(n))
    assert rel(Hd.to_host(), H_mp) <= 1e-13


def test_update_fp32():
    n = 128
    rng = np.random.default_rng(9)
    H0 = _spd(n, 9).astype(np.float32)
    d, y = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    lam = 0.25 if float(d @ y) > 0 else -0.25
    H_ref, d_ref = np.asfortranarray(H0.copy()), d.copy()
    orc.bfgs_update(H_ref, lam, d_ref, y.copy())
    Hd = dzo.DeviceArray.from_host(H0)
    dzo.update_inverse_hessian_(Hd, lam, dzo.DeviceArray.from_host(d), dzo.DeviceArray.from_host(y),
                                dzo.DeviceArray(n, np.float32))
    assert rel(Hd.to_host().astype(np.float64), np.ascontiguousarray(H_ref).astype(np.float64)) <= 5e-6


# ------------------------------------------------------------------------------ step!
def test_config1_readme_example_bfgs_rosenbrock2d():
    """BASELINE config 1 / README.md:33-41: BFGSOptimizer on 2-D Rosenbrock, rand(2) start."""
    x0 = orc.pcg_fill(2, 1)
    ref = orc.BFGS(orc.Problem(orc.ROSENBROCK2D, 2), x0, 1.0)
    opt = dzo.BFGSOptimizer(dzo.Problem(dzo.ROSENBROCK2D, 2), None, dzo.DeviceArray.from_host(x0), 1.0)
    steps = 0
    f_start = ref.current_objective_value
    # lock-step while far from convergence; the last steps are decided by last-bit differences
    # of f (f_b < f, f_b > f_g), so the two may terminate a step or two apart
    while not opt.has_converged and not ref.has_converged and steps < 12:
        opt.step(); ref.step(); steps += 1
        assert opt.iteration_count == ref.iteration_count and opt.last_step_type == ref.last_step_type
        assert np.allclose(opt.current_point.to_host(), ref.current_point, rtol=1e-9, atol=1e-12)
        assert abs(opt.current_objective_value - ref.current_objective_value) <= 1e-9 * f_start + 1e-6 * ref.current_objective_value
    while not opt.has_converged and steps < 500:
        opt.step(); steps += 1
    while not ref.has_converged:
        ref.step()
    assert abs(opt.iteration_count - ref.iteration_count) <= 3
    assert opt.has_converged and steps < 200
    assert np.allclose(opt.current_point.to_host(), [1.0, 1.0], atol=1e-6)
    H = opt.approximate_inverse_hessian.to_host()
    assert np.array_equal(H, H.T)


@pytest.mark.parametrize("n", [16, 200])
def test_bfgs_quadratic_trajectory_and_invariants(n):
    A = orc.quadratic_matrix(n)
    x0 = orc.pcg_fill(n, 4) - 0.5
    ref_p = orc.Problem(orc.QUADRATIC, n, A=A)
    ref = orc.BFGS(ref_p, x0, 1.0)
    opt = dzo.BFGSOptimizer(dzo.Problem(dzo.QUADRATIC, n, A=A), None, dzo.DeviceArray.from_host(x0), 1.0)
    assert np.array_equal(opt.approximate_inverse_hessian.to_host(), np.eye(n))      # :783
    assert rel(opt.next_step_direction.to_host(), ref.next_step_direction) <= 1e-14  # :784
    prev_x, prev_g = opt.current_point.to_host(), opt.current_gradient.to_host()
    f_start = ref_p.eval(x0)
    for it in range(25):
        opt.step(); ref.step()
        if ref.has_terminated or opt.has_terminated:
            break
        x, g = opt.current_point.to_host(), opt.current_gradient.to_host()
        assert opt.last_step_type == ref.last_step_type and opt.iteration_count == ref.iteration_count
        # the minimiser is x = 0: measure errors against the problem's scale, not |x| -> 0
        assert np.linalg.norm(x - ref.current_point) <= 1e-9 * np.linalg.norm(x0)
        assert abs(opt.current_objective_value - ref.current_objective_value) <= 1e-9 * f_start
        assert np.array_equal(x - prev_x, opt.delta_point.to_host())          # run_and_test! :1035-1039
        assert np.array_equal(g - prev_g, opt.delta_gradient.to_host())       # :1042-1046
        H = opt.approximate_inverse_hessian.to_host()
        assert np.array_equal(H, H.T)
        assert rel(opt.next_step_direction.to_host(), H @ g) <= 1e-12         # d = H*g (:958-960)
        assert rel(H, np.ascontiguousarray(ref.approximate_inverse_hessian)) <= 1e-7
        prev_x, prev_g = x, g
    assert opt.current_objective_value < 1e-3 * ref_p.eval(x0)


def test_bfgs_line_search_vertex_and_callbacks():
    n = 6
    A = orc.quadratic_matrix(n)
    x0 = orc.pcg_fill(n, 4) - 0.5
    ref_p = orc.Problem(orc.QUADRATIC, n, A=A)
    opt = dzo.BFGSOptimizer(lambda x: ref_p.eval(x.to_host()), lambda g, x: g.upload(ref_p.grad(x.to_host())),
                            dzo.DeviceArray.from_host(x0), 1.0)
    g = opt.current_gradient.to_host()
    t_star = (g @ g) / (g @ A @ g)
    t, f = opt.line_search(True, 1.0 / np.linalg.norm(g))
    ref = orc.BFGS(ref_p, x0, 1.0)
    t_ref, f_ref = ref.line_search(True, 1.0 / np.linalg.norm(g))
    assert t == pytest.approx(t_ref, rel=1e-12) and f == pytest.approx(f_ref, rel=1e-12)
    assert abs(t - t_star) <= 1e-9 * t_star
    for _ in range(5):
        opt.step(); ref.step()
    assert rel(opt.current_point.to_host(), ref.current_point) <= 1e-9


def test_full_size_config2_update_properties_n4096():
    """BASELINE config 2 size (n = 4096, H = 128 MiB): size-independent properties."""
    n = 4096
    rng = np.random.default_rng(0)
    Hd = dzo.DeviceArray.from_host(np.eye(n))
    g = rng.standard_normal(n)
    for it in range(3):
        d, y = rng.standard_normal(n), rng.standard_normal(n)
        lam = 0.1 if d @ y > 0 else -0.1
        dd, yd, gd = (dzo.DeviceArray.from_host(a) for a in (d, y, g))
        scratch, dnext = dzo.DeviceArray(n), dzo.DeviceArray(n)
        dzo.update_inverse_hessian_(Hd, lam, dd, yd, scratch, gd, dnext)
        H = Hd.to_host()
        assert np.array_equal(H, H.T)
        assert np.allclose(H @ y, lam * d, rtol=1e-9, atol=1e-10)
        assert rel(dnext.to_host(), H @ g) <= 1e-12
        assert rel(dzo.symv_(dzo.DeviceArray(n), Hd, gd).to_host(), H @ g) <= 1e-12


# ------------------------------------------------------------------------------ AdGD
def test_adgd_matches_oracle():
    n = 300
    x0 = orc.rosenbrock_chain_x0(n)
    ref = orc.AdGD(orc.Problem(orc.ROSENBROCK_CHAIN, n), x0.copy(), 0.1)
    opt = dzo.AdGDOptimizer(None, dzo.Problem(dzo.ROSENBROCK_CHAIN, n), None, dzo.DeviceArray.from_host(x0), 0.1)
    for it in range(40):
        opt.step(); ref.step()
        assert opt.is_stuck == ref.is_stuck and opt.iteration_count == ref.iteration_count
        assert rel(opt.current_point.to_host(), ref.current_point) <= 1e-10
        assert opt.current_step_size == pytest.approx(ref.current_step_size, rel=1e-9)
        assert opt.previous_step_size == pytest.approx(ref.previous_step_size, rel=1e-9)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_adgd_walks_along_what_the_host_wrote_into_current_gradient(dtype):
    """step! walks along opt.current_gradient (src/DZOptimization.jl:301), which is the caller's aliased array (:216-239).
    The fused pass recomputes the gradient from the point, so a host write into the array would be ignored: the step
    that follows a look at the arrays checks the array against the point and takes the generic kernels when it was
    changed (the AdGD counterpart of test_point_ring_adopts_what_the_host_wrote_into_the_aliased_arrays)."""
    n = 4100 if dtype == np.float64 else 8200
    x0 = orc.rosenbrock_chain_x0(n, dtype)
    ref = orc.AdGD(orc.Problem(orc.ROSENBROCK_CHAIN, n, dtype), x0.copy(), 0.1)
    opt = dzo.AdGDOptimizer(None, dzo.Problem(dzo.ROSENBROCK_CHAIN, n, dtype), None, dzo.DeviceArray.from_host(x0), 0.1)
    tol = 1e-10 if dtype == np.float64 else 1e-5
    for _ in range(6):
        opt.step(); ref.step()
    assert opt.fused_steps == 6 and opt.host_gradient_steps == 0
    # looking is free of consequences ...
    g = opt.current_gradient.to_host()
    assert np.array_equal(g, ref.problem.grad(opt.current_point.to_host()))
    opt.step(); ref.step()
    assert opt.fused_steps == 7 and opt.host_gradient_steps == 0
    assert rel(opt.current_point.to_host(), ref.current_point) <= tol
    # ... writing is not: both sides get the same (wrong on purpose) gradient array
    g2 = (ref.current_gradient * dtype(0.5) + dtype(1e-3) * np.sin(np.arange(n))).astype(dtype)
    x_before = opt.current_point.to_host()
    opt.current_gradient.upload(g2); ref.current_gradient[:] = g2
    opt.step(); ref.step()
    assert opt.host_gradient_steps == 1 and opt.fused_steps == 7        # this step ran on the generic kernels, along g2
    assert opt.is_stuck == ref.is_stuck and opt.iteration_count == ref.iteration_count
    x_after = opt.current_point.to_host()
    assert rel(x_after, ref.current_point) <= tol
    t = ref.current_step_size
    assert rel(x_after - x_before, -dtype(t) * g2) <= (1e-9 if dtype == np.float64 else 1e-3)     # x_new = x - t g2 (:301), not x - t grad f(x)
    # the step after is a fused one again, from the gradient the generic step computed
    opt.step(); ref.step()
    assert opt.fused_steps == 8 and opt.host_gradient_steps == 1
    assert rel(opt.current_point.to_host(), ref.current_point) <= tol
    assert opt.current_step_size == pytest.approx(ref.current_step_size, rel=1e-9 if dtype == np.float64 else 1e-4)


# ------------------------------------------------------------------------------ batched (K11)
@pytest.mark.parametrize("n,B", [(2, 5), (16, 7), (256, 3)])
def test_batched_bfgs_matches_single_instance_oracle(n, B):
    X0 = np.stack([orc.pcg_fill(n, 1000 + b) for b in range(B)])
    batch = dzo.BatchedBFGS(dzo.ROSENBROCK_CHAIN, X0, 1.0)
    refs = [orc.BFGS(orc.Problem(orc.ROSENBROCK_CHAIN, n), X0[b].copy(), 1.0) for b in range(B)]
    f0 = batch.current_objective_value.to_host()
    for b in range(B):
        assert f0[b] == pytest.approx(refs[b].current_objective_value, rel=1e-13)
    for it in range(6):
        batch.step(1, poll=False)
        for r in refs:
            r.step()
        X, F = batch.current_point.to_host(), batch.current_objective_value.to_host()
        its, types = batch.iteration_count.to_host(), batch.last_step_type.to_host()
        H = batch.approximate_inverse_hessian.to_host()
        for b in range(B):
            assert its[b] == refs[b].iteration_count and types[b] == refs[b].last_step_type, (it, b)
            assert rel(X[b], refs[b].current_point) <= 1e-8, (it, b)
            assert F[b] == pytest.approx(refs[b].current_objective_value, rel=1e-8)
            assert np.array_equal(H[b], H[b].T)
            assert rel(H[b], np.ascontiguousarray(refs[b].approximate_inverse_hessian)) <= 1e-6


def test_batched_bfgs_runs_to_termination_and_counts():
    n, B = 8, 64
    X0 = np.stack([orc.pcg_fill(n, 1000 + b) for b in range(B)])
    batch = dzo.BatchedBFGS(dzo.ROSENBROCK_CHAIN, X0, 1.0)
    assert batch.count_active() == B
    done, rounds = False, 0
    while not done and rounds < 200:
        done = batch.step(8)
        rounds += 1
    assert done and batch.count_active() == 0
    F = batch.current_objective_value.to_host()
    assert (F < 1e-12).mean() > 0.9            # Rosenbrock-8 has a second local minimum near x1 = -1
    assert batch.has_terminated.to_host().all()
    it_before = batch.iteration_count.to_host().copy()
    batch.step(3)
    assert np.array_equal(batch.iteration_count.to_host(), it_before)   # terminated instances never move


def test_full_size_config5_shard_properties():
    """One GPU's shard of config 5: 1024 instances of n = 256 (512 MiB of H)."""
    n, B = 256, 1024
    X0 = np.stack([orc.pcg_fill(n, 1000 + b) for b in range(B)])
    batch = dzo.BatchedBFGS(dzo.ROSENBROCK_CHAIN, X0, 1.0)
    f0 = batch.current_objective_value.to_host().copy()
    batch.step(5, poll=False)
    f1 = batch.current_objective_value.to_host()
    assert (f1 < f0).all()
    H = batch.approximate_inverse_hessian.to_host()
    for b in (0, 17, 1023):
        assert np.array_equal(H[b], H[b].T)
    g, d = batch.current_gradient.to_host(), batch.next_step_direction.to_host()
    for b in (0, 511):
        assert rel(d[b], H[b] @ g[b]) <= 1e-11
    # one instance against the oracle
    ref = orc.BFGS(orc.Problem(orc.ROSENBROCK_CHAIN, n), X0[3].copy(), 1.0)
    for _ in range(5):
        ref.step()
    assert rel(batch.current_point.to_host()[3], ref.current_point) <= 1e-7


# ------------------------------------------------------------------------------ re-precision (a10)
def test_reprecision_constructor_continues_the_run():
    """BFGSOptimizer(::Type{T}, opt) (legacy/DZOptimization.jl:812-862): fp32 warm start, fp64 finish."""
    n = 32
    x0 = orc.pcg_fill(n, 7)
    p32 = dzo.Problem(dzo.ROSENBROCK_CHAIN, n, np.float32)
    p64 = dzo.Problem(dzo.ROSENBROCK_CHAIN, n, np.float64)
    lo = dzo.BFGSOptimizer(p32, None, dzo.DeviceArray.from_host(x0.astype(np.float32)), 1.0)
    for _ in range(15):
        lo.step()
    H32, x32 = lo.approximate_inverse_hessian.to_host(), lo.current_point.to_host()
    hi = dzo.BFGSOptimizer.convert(np.float64, lo, p64)
    assert hi.dtype == np.float64 and hi.iteration_count == lo.iteration_count            # :848
    assert not hi.has_terminated and hi.last_step_type == lo.last_step_type               # :849,:856
    assert np.array_equal(hi.current_point.to_host(), x32.astype(np.float64))              # :825 T.(x)
    H64 = hi.approximate_inverse_hessian.to_host()
    assert np.array_equal(H64, H32.astype(np.float64))                                     # :832
    ref_p = orc.Problem(orc.ROSENBROCK_CHAIN, n)
    x = hi.current_point.to_host()
    assert hi.current_objective_value == pytest.approx(ref_p.eval(x), rel=1e-13)           # :828 in fp64
    assert np.array_equal(hi.current_gradient.to_host(), ref_p.grad(x))                    # :830-831
    assert rel(hi.next_step_direction.to_host(), H64 @ ref_p.grad(x)) <= 1e-13            # :833-836
    assert np.array_equal(hi.delta_point.to_host(), lo.delta_point.to_host().astype(np.float64))
    f_before = hi.current_objective_value
    steps = 0
    while not hi.has_converged and steps < 400:
        hi.step(); steps += 1
    assert hi.has_converged and hi.current_objective_value < 1e-18 < f_before
    assert np.allclose(hi.current_point.to_host(), 1.0, atol=1e-8)


# ------------------------------------------------------------------------------ legacy GD (8f.4)
def test_legacy_gradient_descent_matches_oracle():
    n = 300
    x0 = orc.rosenbrock_chain_x0(n)
    ref_p = orc.Problem(orc.ROSENBROCK_CHAIN, n)
    ref = orc.GradientDescent(ref_p, x0, 0.1)
    opt = dzo.GradientDescentOptimizer(dzo.Problem(dzo.ROSENBROCK_CHAIN, n), None, "QuadraticLineSearch",
                                       dzo.DeviceArray.from_host(x0), 0.1)
    assert rel(opt.next_step_direction.to_host(), ref.next_step_direction) <= 1e-14
    scale = np.linalg.norm(x0)
    for it in range(25):
        opt.step(); ref.step()
        assert opt.has_terminated == ref.has_terminated and opt.iteration_count == ref.iteration_count
        x = opt.current_point.to_host()
        assert np.linalg.norm(x - ref.current_point) <= 1e-10 * scale, it
        assert opt.current_objective_value == pytest.approx(ref.current_objective_value, rel=1e-10)
        assert opt.last_step_length == pytest.approx(ref.last_step_length, rel=1e-9)
        assert opt.delta_objective_value == pytest.approx(ref.delta_objective_value, rel=1e-6)
        assert np.array_equal(opt.current_gradient.to_host(), ref_p.grad(x))
        assert rel(opt.next_step_direction.to_host(), ref.next_step_direction) <= 1e-9
    with pytest.raises(dzo.DzoError):
        dzo._check(dzo.lib().dzo_bfgs_step(opt.h))          # a GD handle is not a BFGS handle


# ------------------------------------------------------------------------------ MFMA variant of K9
@pytest.mark.parametrize("n", [16, 64, 256, 1024])
def test_mfma_update_matches_oracle_within_rounding(n):
    """v_mfma_f64_16x16x4_f64 form of the rank-2 update: same mathematics, fma-chain rounding;
    asymmetric tiles make a deliberately NON-symmetric V to catch a row/column swap."""
    rng = np.random.default_rng(n)
    H0 = _spd(n, n)
    d, y = rng.standard_normal(n), rng.standard_normal(n)
    lam = -0.37 if d @ y < 0 else 0.37
    H_ref, d_ref = np.asfortranarray(H0.copy()), d.copy()
    t_ref = orc.bfgs_update(H_ref, lam, d_ref, y.copy())
    Hd, dd, yd = (dzo.DeviceArray.from_host(a) for a in (H0, d, y))
    scratch = dzo.DeviceArray(n)
    dzo.update_inverse_hessian_mfma_(Hd, lam, dd, yd, scratch)
    H_gpu = Hd.to_host()
    assert rel(H_gpu, np.ascontiguousarray(H_ref)) <= 1e-13
    assert rel(H_gpu, H_gpu.T) <= 1e-15                       # symmetric to rounding, not bit-exact
    assert rel(dd.to_host(), d_ref) <= 1e-14 and rel(scratch.to_host(), t_ref) <= 1e-13
    assert np.allclose(H_gpu @ y, lam * d, rtol=1e-9, atol=1e-11)      # secant equation
    with pytest.raises(dzo.DzoError):
        dzo.update_inverse_hessian_mfma_(dzo.DeviceArray.from_host(np.eye(24)), 1.0, dzo.DeviceArray(24),
                                         dzo.DeviceArray(24), dzo.DeviceArray(24))         # n % 16 != 0


@pytest.mark.parametrize("n", [16, 200, 513])
def test_side_by_side_line_searches_equal_the_sequential_ones(n, monkeypatch):
    """The dense BFGS step advances its two line searches (gradient and quasi-Newton direction)
    side by side, two evaluations per pass over A, and forms trial points inside the objective
    kernel.  Both are re-schedulings: same values, same decisions, same evaluation counts."""
    A = orc.quadratic_matrix(n)
    x0 = orc.pcg_fill(n, 4) - 0.5
    runs = []
    for dual, fused in (("1", "1"), ("0", "1"), ("0", "0")):
        monkeypatch.setenv("DZO_TUNE_BFGS_DUAL_SEARCH", dual)
        monkeypatch.setenv("DZO_TUNE_BFGS_PHI_FUSED", fused)
        opt = dzo.BFGSOptimizer(dzo.Problem(dzo.QUADRATIC, n, A=A), None, dzo.DeviceArray.from_host(x0), 1.0)
        rows = []
        for _ in range(20):
            opt.step()
            rows.append((opt.current_point.to_host(), opt.current_objective_value, opt.last_step_type,
                         opt.last_step_length, opt.objective_evaluations))
            if opt.has_terminated:
                break
        runs.append(rows)
    for other in runs[1:]:
        assert len(other) == len(runs[0])
        for a, b in zip(runs[0], other):
            assert np.array_equal(a[0], b[0]) and a[1:] == b[1:]


@pytest.mark.parametrize("n,dtype,step0,max_inc", [(16, np.float64, 1.0, 0), (200, np.float64, 1.0, 0), (513, np.float64, 1.0, 0),
                                                   (200, np.float32, 1.0, 0), (64, np.float64, 1e4, 0), (64, np.float64, 1e-7, 0),
                                                   (200, np.float64, 1e-7, 3), (130, np.float32, 50.0, 2), (4096, np.float64, 1.0, 0)])
def test_device_driven_line_searches_equal_the_host_driven_ones(n, dtype, step0, max_inc, monkeypatch):
    """bfgs_dev_search: the two searches' state machines run on the device (norm kernel begins them, every round's
    finish kernel feeds them and posts the next requests), the host waits once per step.  A re-scheduling again:
    same points, values, step types and lengths, evaluation counts -- whatever number of rounds is enqueued ahead."""
    A = orc.quadratic_matrix(n).astype(dtype)
    x0 = (orc.pcg_fill(n, 4) - 0.5).astype(dtype)
    runs = []
    for dev, rounds in (("0", "2"), ("1", "2"), ("1", "1"), ("1", "4")):
        monkeypatch.setenv("DZO_TUNE_BFGS_DEV_SEARCH", dev)
        monkeypatch.setenv("DZO_TUNE_BFGS_DEV_ROUNDS", rounds)
        opt = dzo.BFGSOptimizer(dzo.Problem(dzo.QUADRATIC, n, A=A, dtype=dtype), None, dzo.DeviceArray.from_host(x0), step0)
        if max_inc:
            opt.set_max_increases(max_inc)
        rows = []
        for _ in range(12 if n == 4096 else 40):
            opt.step()
            rows.append((opt.current_point.to_host(), opt.current_gradient.to_host(), opt.current_objective_value, opt.last_step_type,
                         opt.last_step_length, opt.objective_evaluations, opt.iteration_count))
            if opt.has_terminated:
                break
        runs.append(rows)
    assert len(runs[0]) >= 3
    for other in runs[1:]:
        assert len(other) == len(runs[0])
        for a, b in zip(runs[0], other):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2:] == b[2:]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n", [8, 124, 126, 248, 4100, 100_000])
def test_adgd_fused_step_equals_separate_kernels(n, dtype, monkeypatch):
    """The one-pass AdGD step (built-in chained Rosenbrock) against the generic kernel sequence
    (DZO_TUNE_ADGD_FUSED=0): same x, g, deltas bit for bit per step from the same state; objective
    and the two norms are summed in another order, so step sizes agree to rounding."""
    x0 = orc.rosenbrock_chain_x0(n).astype(dtype)
    monkeypatch.setenv("DZO_TUNE_ADGD_FUSED", "1")
    a = dzo.AdGDOptimizer(None, dzo.Problem(dzo.ROSENBROCK_CHAIN, n, dtype=dtype), None, dzo.DeviceArray.from_host(x0), 0.1)
    monkeypatch.setenv("DZO_TUNE_ADGD_FUSED", "0")
    b = dzo.AdGDOptimizer(None, dzo.Problem(dzo.ROSENBROCK_CHAIN, n, dtype=dtype), None, dzo.DeviceArray.from_host(x0), 0.1)
    tol = 1e-12 if dtype == np.float64 else 2e-5
    for it in range(25):
        a.step(); b.step()
        assert a.is_stuck == b.is_stuck and a.iteration_count == b.iteration_count
        if a.is_stuck:
            break
        assert a.current_step_size == pytest.approx(b.current_step_size, rel=tol)
        assert a.current_objective_value == pytest.approx(b.current_objective_value, rel=tol)
        xa, xb = a.current_point.to_host(), b.current_point.to_host()
        assert rel(xa, xb) <= tol
        assert rel(a.current_gradient.to_host(), b.current_gradient.to_host()) <= 10 * tol
        assert rel(a.delta_point.to_host(), b.delta_point.to_host()) <= 1e3 * tol
    vecn = 16 // np.dtype(dtype).itemsize
    assert b.fused_steps == 0 and (a.fused_steps > 0) == (n % vecn == 0 and n >= 4 * vecn)


def test_adgd_fused_step_rejected_first_trial_falls_back():
    """A first step that is far too long: the fused pass is rejected, the point restored and the
    halving loop of take_backtracking_step! (:121-152) runs on the separate kernels; the oracle
    takes the same steps."""
    n = 1240
    x0 = orc.rosenbrock_chain_x0(n)
    ref = orc.AdGD(orc.Problem(orc.ROSENBROCK_CHAIN, n), x0.copy(), 200.0)
    opt = dzo.AdGDOptimizer(None, dzo.Problem(dzo.ROSENBROCK_CHAIN, n), None, dzo.DeviceArray.from_host(x0), 200.0)
    for it in range(60):
        opt.step(); ref.step()
        assert opt.is_stuck == ref.is_stuck and opt.iteration_count == ref.iteration_count
        assert rel(opt.current_point.to_host(), ref.current_point) <= 1e-10
        assert opt.current_objective_value == pytest.approx(ref.current_objective_value, rel=1e-10)
        assert rel(opt.delta_point.to_host(), ref.delta_point) <= 1e-9
        assert rel(opt.delta_gradient.to_host(), ref.delta_gradient) <= 1e-9
    assert opt.fused_rejections >= 1 and opt.fused_steps >= 1


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_adgd_fused_run_to_stuck_leaves_a_consistent_state(dtype):
    """Run the one-pass AdGD step until is_stuck (:128-130): the point, gradient and objective it
    leaves must belong together (the pass had overwritten them and restores from its backups), and
    further step! calls are no-ops (:276-278)."""
    n = 64
    x0 = orc.rosenbrock_chain_x0(n).astype(dtype)
    p = dzo.Problem(dzo.ROSENBROCK_CHAIN, n, dtype=dtype)
    opt = dzo.AdGDOptimizer(None, p, None, dzo.DeviceArray.from_host(x0), 0.1)
    for it in range(200_000):
        opt.step()
        if opt.is_stuck:
            break
    assert opt.is_stuck and opt.fused_steps > 0
    x, g, f, its = opt.current_point.to_host(), opt.current_gradient.to_host(), opt.current_objective_value, opt.iteration_count
    ref_p = orc.Problem(orc.ROSENBROCK_CHAIN, n, dtype=dtype)
    assert np.array_equal(ref_p.grad(x), g)
    assert f == pytest.approx(ref_p.eval(x), rel=1e-5 if dtype == np.float32 else 1e-12, abs=1e-30)
    opt.step()
    assert opt.iteration_count == its and np.array_equal(opt.current_point.to_host(), x)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,step0", [(4100, 0.1), (100_000, 0.1), (4100, 200.0)])
def test_adgd_fused_halo_handover_without_touching_pointers(n, step0, dtype, monkeypatch):
    """The default path of the one-pass AdGD step: the row-boundary vectors of x_new / g_new that a pass
    leaves behind are consumed by the NEXT pass (no snapshot kernel).  Reading opt.current_point goes
    through dzo_adgd_get_ptr, which invalidates that hand-over -- so here K steps run with NO pointer
    access in between (many wave-rows, so inter-row halos matter), and only then the state is compared
    with the separate-kernel path and the oracle.  step0 = 200 makes the first trials fail, so halos
    written by a RETRY pass are consumed as well."""
    K = 24
    x0 = orc.rosenbrock_chain_x0(n).astype(dtype)
    monkeypatch.setenv("DZO_TUNE_ADGD_FUSED", "1")
    a = dzo.AdGDOptimizer(None, dzo.Problem(dzo.ROSENBROCK_CHAIN, n, dtype=dtype), None, dzo.DeviceArray.from_host(x0), step0)
    monkeypatch.setenv("DZO_TUNE_ADGD_FUSED", "0")
    b = dzo.AdGDOptimizer(None, dzo.Problem(dzo.ROSENBROCK_CHAIN, n, dtype=dtype), None, dzo.DeviceArray.from_host(x0), step0)
    if dtype == np.float32:
        orc.set_dot_mode(orc.DOT_WIDE)                    # the device sums in fp64
    try:
        ref = orc.AdGD(orc.Problem(orc.ROSENBROCK_CHAIN, n, dtype), x0.copy(), step0)
        for _ in range(K):
            a.step(); b.step(); ref.step()               # scalars only: no get_ptr between the steps
        assert a.fused_steps == K and b.fused_steps == 0
        if step0 > 1.0:
            assert a.fused_rejections >= 1
        assert a.iteration_count == b.iteration_count == ref.iteration_count == K
        tol = 1e-12 if dtype == np.float64 else 2e-5
        assert a.current_step_size == pytest.approx(b.current_step_size, rel=tol)
        assert a.current_objective_value == pytest.approx(b.current_objective_value, rel=tol)
        xa, ga = a.current_point.to_host(), a.current_gradient.to_host()
        assert rel(xa, b.current_point.to_host()) <= tol
        assert rel(ga, b.current_gradient.to_host()) <= 10 * tol
        assert rel(a.delta_point.to_host(), b.delta_point.to_host()) <= 1e3 * tol
        assert rel(a.delta_gradient.to_host(), b.delta_gradient.to_host()) <= 1e3 * tol
        # a wrong boundary vector would put a wrong gradient next to a row end: the gradient must be
        # the gradient of the point, exactly
        assert np.array_equal(orc.Problem(orc.ROSENBROCK_CHAIN, n, dtype).grad(xa), ga)
        otol = 1e-10 if dtype == np.float64 else 1e-4
        assert rel(xa, ref.current_point) <= otol
        assert a.current_step_size == pytest.approx(ref.current_step_size, rel=1e-9 if dtype == np.float64 else 1e-4)
    finally:
        orc.set_dot_mode(orc.DOT_SEQUENTIAL)


@pytest.mark.parametrize("fused", ["1", "0"])
def test_adgd_stuck_step_leaves_the_reference_deltas(fused, monkeypatch):
    """take_backtracking_step! returns at :128-130 with delta_point == x_old (the :118 copy) and
    delta_gradient untouched since the previous step; the one-pass step must leave the same."""
    n = 64
    monkeypatch.setenv("DZO_TUNE_ADGD_FUSED", fused)
    x0 = orc.rosenbrock_chain_x0(n)
    opt = dzo.AdGDOptimizer(None, dzo.Problem(dzo.ROSENBROCK_CHAIN, n), None, dzo.DeviceArray.from_host(x0), 0.1)
    ref = orc.AdGD(orc.Problem(orc.ROSENBROCK_CHAIN, n), x0.copy(), 0.1)
    prev_dg = None
    for it in range(200_000):
        dg_before = opt.delta_gradient.to_host()
        opt.step()
        if opt.is_stuck:
            prev_dg = dg_before
            break
    assert opt.is_stuck
    assert np.array_equal(opt.delta_point.to_host(), opt.current_point.to_host())     # :118
    assert np.array_equal(opt.delta_gradient.to_host(), prev_dg)                       # untouched
    while not ref.is_stuck:
        ref.step()
    assert np.array_equal(ref.delta_point, ref.current_point)


def test_bfgs_reset_restores_identity_and_gradient_direction():
    """dzo_bfgs_reset = the reset step! performs after a gradient-descent step (legacy :981-986)."""
    n = 48
    x0 = orc.pcg_fill(n, 4) - 0.5
    opt = dzo.BFGSOptimizer(dzo.Problem(dzo.ROSENBROCK_CHAIN, n), None, dzo.DeviceArray.from_host(x0), 1.0)
    for _ in range(40):
        opt.step()
        if opt.last_step_type == 2:                               # a BFGS step: H has been updated
            break
    H = opt.approximate_inverse_hessian.to_host().reshape(n, n)
    assert opt.last_step_type == 2 and not np.array_equal(H, np.eye(n))
    opt.reset_inverse_hessian()
    assert np.array_equal(opt.approximate_inverse_hessian.to_host().reshape(n, n), np.eye(n))
    assert np.array_equal(opt.next_step_direction.to_host(), opt.current_gradient.to_host())
    f = opt.current_objective_value
    opt.step()
    assert opt.current_objective_value < f


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,step0", [(4100, 0.1), (100_000, 0.1), (4100, 200.0)])
def test_adgd_pipelined_passes_are_the_same_steps(n, step0, dtype, monkeypatch):
    """Behind every AdGD pass the next one is enqueued before the host has seen the decision; its step size comes
    from the device-side twin of the recurrence (src/DZOptimization.jl:285-299) or of the halving (:152).  The
    host adopts such a pass only when the device used exactly the host's own value, so a pipelined run must be
    the run with one host round trip per pass (DZO_TUNE_ADGD_PIPELINE=0), bit for bit -- scalars included --
    and the passes must really have been adopted; handing out a pointer in between drops the pass in flight
    and changes nothing either.  Since round 4 the decision on a pass is taken in the prologue of the pass behind it;
    DZO_TUNE_ADGD_PROLOGUE=0 keeps it a kernel of its own ("kernel" below), the same steps again."""
    K = 30
    x0 = orc.rosenbrock_chain_x0(n).astype(dtype)
    runs = {}
    for mode in ("1", "0", "peek", "kernel", "kernel0"):
        monkeypatch.setenv("DZO_TUNE_ADGD_PIPELINE", "0" if mode in ("0", "kernel0") else "1")
        monkeypatch.setenv("DZO_TUNE_ADGD_PROLOGUE", "0" if mode.startswith("kernel") else "1")
        opt = dzo.AdGDOptimizer(None, dzo.Problem(dzo.ROSENBROCK_CHAIN, n, dtype=dtype), None, dzo.DeviceArray.from_host(x0), step0)
        scal = []
        for it in range(K):
            opt.step()
            scal.append((opt.current_objective_value, opt.current_step_size, opt.previous_step_size, opt.iteration_count, opt.is_stuck))
            if mode == "peek" and it % 3 == 1:
                opt.current_point.to_host()                               # get_ptr: the pass in flight is dropped
        runs[mode] = (scal, opt.current_point.to_host(), opt.current_gradient.to_host(), opt.delta_point.to_host(),
                      opt.delta_gradient.to_host(), opt.fused_steps, opt.fused_rejections, opt.pipelined_passes, opt.pipeline_discards)
        assert opt.pipeline_corrections == 0
    a, b, c = runs["1"], runs["0"], runs["peek"]
    for other in (b, c, runs["kernel"], runs["kernel0"]):
        assert a[0] == other[0]
        for i in (1, 2, 3, 4):
            assert np.array_equal(a[i], other[i])
        assert a[5:7] == other[5:7]
    its = a[0][-1][3]                                       # (fp32 may end stuck before K steps)
    assert a[5] == its and its >= 10
    assert b[7] == 0 and b[8] == 0
    assert a[7] >= its - 2 and a[8] <= 2                    # every pass but the very first was already in flight
    assert runs["kernel"][7] >= its - 2 and runs["kernel0"][7] == 0
    assert c[8] >= min(its, K) // 3 - 1 and c[7] >= 1


def test_adgd_watched_through_read_is_the_same_run():
    """dzo_adgd_read (behind `opt.current_point.to_host()`): no pointer hand-out, so the next step neither checks the gradient
    array against the point nor leaves the one-pass kernel; the run is the unwatched run bit for bit.  A write through the
    caller's own handle behind such a read is still seen (the handle stays registered for dzo_memcpy_*)."""
    n = 100_000
    x0 = orc.rosenbrock_chain_x0(n)
    runs = []
    for watch in (False, True):
        xd = dzo.DeviceArray.from_host(x0)
        opt = dzo.AdGDOptimizer(None, dzo.Problem(dzo.ROSENBROCK_CHAIN, n), None, xd, 0.1)
        for it in range(12):
            opt.step()
            if watch and it % 2 == 0:
                x = opt.current_point.to_host()
                assert np.array_equal(opt.current_gradient.to_host(), orc.Problem(orc.ROSENBROCK_CHAIN, n).grad(x))
        assert opt.host_gradient_steps == 0 and opt.fused_steps == 12
        runs.append((opt.current_point.to_host(), opt.current_objective_value, opt.current_step_size, opt.previous_step_size))
    assert np.array_equal(runs[0][0], runs[1][0]) and runs[0][1:] == runs[1][1:]
    # a read, then the host overwrites current_gradient through a handle of its own: the next step walks along what it wrote
    g = opt.current_gradient.to_host()
    gd = dzo.DeviceArray(n, np.float64, ptr=opt._get_ptr(2), owner=False)     # (the pointer a caller would have kept from the constructor)
    opt.step()                                                                 # (that hand-out is consumed by this step's check)
    before = opt.host_gradient_steps
    _ = opt.current_point.to_host()                                            # read-only look
    g2 = opt.current_gradient.to_host() * 0.5
    gd.upload(g2)                                                              # dzo_memcpy_h2d: the look goes on record
    opt.step()
    assert opt.host_gradient_steps == before + 1


def test_a_failed_allocation_does_not_poison_later_calls():
    """hipMalloc leaves its error with the runtime until somebody calls hipGetLastError(): an out-of-memory dzo_malloc used to
    come back as a spurious 'HIP error 2 (out of memory)' from the next unrelated launch check (found by tools/soak.py's
    memory probe).  Every out-of-memory return of the library clears it now."""
    with pytest.raises(dzo.DzoError):
        dzo.DeviceArray(1 << 46)                                            # 512 TiB
    n = 4100
    x0 = orc.rosenbrock_chain_x0(n)
    opt = dzo.LBFGSOptimizer(None, dzo.Problem(dzo.ROSENBROCK_CHAIN, n), None, dzo.DeviceArray.from_host(x0), 1.0, 5)
    for _ in range(3):
        opt.step()
    assert np.isfinite(opt.current_objective_value)
