"""The fuzz sweeps (tests/fuzz_*.py: random shapes, every step compared with the oracle or with the other scheduling) under pytest with a
fixed seed and 20 cases each; the scripts themselves take FUZZ_CASES / FUZZ_SEED for longer runs by hand."""
import pytest

pytestmark = pytest.mark.gpu


def test_fuzz_point_ring_steps_against_the_oracle():
    import fuzz_points
    out = fuzz_points.run(cases=20, seed=2468)
    assert out["steps"] >= 100, out
    assert out["worst"]["float64"] <= 1e-10, out              # north_star's per-step bar (the script's own assert is looser)


def test_fuzz_lbfgs_two_pass_steps_against_the_oracle():
    import fuzz_lbfgs
    out = fuzz_lbfgs.run(cases=20, seed=12345)
    assert out["steps"] >= 100 and out["worst"] <= 1e-10, out


def test_fuzz_adgd_steps_against_the_oracle():
    import fuzz_adgd
    out = fuzz_adgd.run(cases=20, seed=4321)
    assert out["fused_steps"] >= 100, out


def test_fuzz_dense_bfgs_device_driven_searches_against_the_host_driven_ones():
    import fuzz_bfgs_search
    from dzo_loader import dzo
    steps, terminated = fuzz_bfgs_search.run(cases=25, seed=97531)
    assert steps >= 150, (steps, terminated)
    assert dzo.unsealed_first_reads() >= 0                     # (the diagnostic counter of wait_sealed is reachable)


def test_fuzz_batched_bfgs_steps_against_per_instance_oracles():
    import fuzz_batched
    out = fuzz_batched.run(cases=12, seed=24680)
    assert out["instance_steps"] >= 100, out


def test_regression_case_of_the_round_3_memory_fault():
    """gpurun_out/r03_fuzzdbg.log (round 3, Oct 4 15:31): "Memory access fault by GPU ... on address (nil)" in tools/fuzz_dbg.py,
    case `n 7936 m 15 warm 30 mode 1`, right after the ring went from the point layout to the pair layout at step 1 (the
    script installs the oracle's state -- set_history -- after every step).  The binary was an uncommitted working tree of
    the edge-array experiment (between c46da1b and 5842ee1; the array and everything that read it were removed again in
    1b37763).  The exact sequence of that script -- its case 0 (n 1000, m 6, CHAIN mode, 35 steps)
    and then the faulting case 1, same seeds -- runs here ONCE per test session as a regression test; every pass launch
    now checks its pointers on the host first (fused_params_ok)."""
    import numpy as np
    from dzo_loader import dzo
    from oracle import oracle as orc
    for n, m, warm, mode, seed in ((1000, 6, 35, 0, 204176), (7936, 15, 30, 1, 391109)):
        x0 = (orc.pcg_fill(n, seed) - 0.5) * 2.0
        ref = orc.LBFGS(orc.Problem(orc.ROSENBROCK_CHAIN, n), x0.copy(), 0.5, m)
        opt = dzo.LBFGSOptimizer(None, dzo.Problem(dzo.ROSENBROCK_CHAIN, n), None, dzo.DeviceArray.from_host(x0), 0.5, m)
        assert opt.ring_layout == 2
        opt.set_two_loop_mode(mode)
        assert opt.ring_layout == (0 if mode == 0 else 2)
        layouts = []
        for it in range(warm):
            if ref.is_stuck:
                break
            layouts.append(opt.ring_layout)
            opt.step(); ref.step()
            if ref.is_stuck or opt.is_stuck:
                break
            e = np.linalg.norm(opt.step_direction.to_host() - ref.step_direction) / np.linalg.norm(ref.step_direction)
            assert e <= 1e-8, (n, it, e)
            S, Y = ref.history_arrays()
            opt.current_point.upload(ref.current_point); opt.current_gradient.upload(ref.current_gradient)
            opt.set_objective_value(ref.current_objective_value)
            opt.set_history(S, Y, ref.rho_history, iteration_count=ref.iteration_count)
        if mode == 1:
            assert layouts[:2] == [2, 1] and set(layouts[1:]) == {1}      # points at step 0, pairs from step 1 on: where it faulted
        opt.close(); ref.close()


def test_fuzz_looks_between_steps_change_nothing():
    import fuzz_looks
    out = fuzz_looks.run(cases=60, seed=1357)
    assert out["cases"] == 60
