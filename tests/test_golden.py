"""Golden vectors (tests/golden/hot_path_golden.json; self-pinned, see its header and
make_golden.py): the oracle must keep reproducing them (CPU), and the HIP path must match
them (GPU)."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "hot_path_golden.json")))


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


def test_header_says_self_pinned():
    assert "self-pinned" in G["header"] and "not executable" in G["header"]


def test_oracle_reproduces_pcg_and_two_loop_goldens():
    assert orc.pcg_raw(8, G["pcg32"]["seed"]).tolist() == G["pcg32"]["first_u32"]
    for c in G["two_loop"]:
        g, S, Y, rho = (np.array(c[k]) for k in ("g", "S", "Y", "rho"))
        d, alpha = orc.lbfgs_direction(g, S.reshape(c["k"], c["n"]), Y.reshape(c["k"], c["n"]), rho)
        assert rel(d, c["d"]) <= c["tolerance_rel_l2"]
        assert np.allclose(alpha, c["alpha"], rtol=1e-11)


def test_oracle_reproduces_update_and_trajectory_goldens():
    for c in G["bfgs_update"]:
        H = np.asfortranarray(np.array(c["H"]))
        d = np.array(c["d"])
        orc.bfgs_update(H, c["lambda"], d, np.array(c["y"]))
        assert rel(np.ascontiguousarray(H), c["H_new"]) <= c["tolerance_rel_fro"]
        assert rel(d, c["d_scaled"]) <= 1e-14
    t = G["lbfgs_trajectory"]
    opt = orc.LBFGS(orc.Problem(orc.ROSENBROCK_CHAIN, t["n"]), np.array(t["x0"]), 1.0, t["m"])
    for row in t["steps"]:
        opt.step()
        assert np.array_equal(opt.current_point, np.array(row["x"])) and opt.current_objective_value == row["f"]
    b = G["rosenbrock2d_bfgs"]
    o = orc.BFGS(orc.Problem(orc.ROSENBROCK2D, 2), np.array(b["x0"]), 1.0)
    for row in b["steps"]:
        o.step()
        assert np.array_equal(o.current_point, np.array(row["x"])) and o.last_step_type == row["type"]


@pytest.mark.gpu
def test_hip_two_loop_matches_goldens():
    from dzo_loader import dzo
    for mode in (dzo.TWOLOOP_CHAIN, dzo.TWOLOOP_GRAM):
        for c in G["two_loop"]:
            n, k = c["n"], c["k"]
            x, g = dzo.DeviceArray.zeros(n), dzo.DeviceArray.from_host(np.array(c["g"]))
            opt = dzo.LBFGSOptimizer(None, lambda x_: 0.0, lambda g_, x_: None, x, 0.0, g, 1.0, max(k, 1))
            opt.set_two_loop_mode(mode)
            opt.set_history(np.array(c["S"]).reshape(k, n), np.array(c["Y"]).reshape(k, n), c["rho"])
            assert rel(opt.compute_step_direction().to_host(), c["d"]) <= 1e-11
            assert np.allclose(opt.alpha_history[:k], c["alpha"], rtol=1e-10)


@pytest.mark.gpu
def test_hip_update_and_trajectories_match_goldens():
    from dzo_loader import dzo
    for c in G["bfgs_update"]:
        n = c["n"]
        H = dzo.DeviceArray.from_host(np.array(c["H"]))
        d = dzo.DeviceArray.from_host(np.array(c["d"]))
        dzo.update_inverse_hessian_(H, c["lambda"], d, dzo.DeviceArray.from_host(np.array(c["y"])), dzo.DeviceArray(n))
        assert rel(H.to_host(), c["H_new"]) <= 1e-12
        assert rel(d.to_host(), c["d_scaled"]) <= 1e-14
    t = G["lbfgs_trajectory"]
    opt = dzo.LBFGSOptimizer(None, dzo.Problem(dzo.ROSENBROCK_CHAIN, t["n"]), None,
                             dzo.DeviceArray.from_host(np.array(t["x0"])), 1.0, t["m"])
    for row in t["steps"]:
        opt.step()
        assert rel(opt.current_point.to_host(), row["x"]) <= t["tolerance_rel"]
        assert abs(opt.current_objective_value - row["f"]) <= t["tolerance_rel"] * abs(row["f"])
        assert opt.last_trials == row["trials"]
    b = G["rosenbrock2d_bfgs"]
    o = dzo.BFGSOptimizer(dzo.Problem(dzo.ROSENBROCK2D, 2), None, dzo.DeviceArray.from_host(np.array(b["x0"])), 1.0)
    for row in b["steps"]:
        o.step()
        assert rel(o.current_point.to_host(), row["x"]) <= b["tolerance_rel"] and o.last_step_type == row["type"]


def _options_opt(make, c):
    opt = make(c)
    opt.set_safeguards(True, True)
    if c["wolfe"]:
        opt.set_line_search(1, c["c1"], c["c2"], 40)
    return opt


def test_oracle_reproduces_option_goldens():
    for c in G["lbfgs_options"]:
        opt = _options_opt(lambda c: orc.LBFGS(orc.Problem(orc.ROSENBROCK_CHAIN, c["n"]), np.array(c["x0"]), 1.0, c["m"]), c)
        for row in c["steps"]:
            opt.step()
            assert opt.last_trials == row["trials"] and opt.last_step_kind == row["kind"]
            assert rel(opt.current_point, row["x"]) <= c["tolerance_rel"]
            assert opt.last_step_length == pytest.approx(row["last_step_length"], rel=1e-12)
            assert row["sy"] > 0 or not c["wolfe"]


@pytest.mark.gpu
def test_hip_options_match_goldens():
    from dzo_loader import dzo
    for c in G["lbfgs_options"]:
        x = dzo.DeviceArray.from_host(np.array(c["x0"]))
        opt = _options_opt(lambda c: dzo.LBFGSOptimizer(None, dzo.Problem(dzo.ROSENBROCK_CHAIN, c["n"]), None, x, 1.0, c["m"]), c)
        for row in c["steps"]:
            opt.step()
            assert opt.last_trials == row["trials"] and opt.last_step_kind == row["kind"]
            assert rel(opt.current_point.to_host(), row["x"]) <= c["tolerance_rel"]
            assert opt.current_objective_value == pytest.approx(row["f"], rel=1e-10)
            assert opt.last_step_length == pytest.approx(row["last_step_length"], rel=1e-10)
