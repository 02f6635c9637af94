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
