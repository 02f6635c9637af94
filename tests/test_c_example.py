"""The plain-C consumers of include/dzo.h (examples/readme_rosenbrock.c = the reference's README
example, README.md:25-41; examples/batched_shards.c = one process driving a shard of independent optimizers on
every visible GPU): they must compile and link against libdzo_hip.so with gcc alone
(CPU test), and converge to (1, ..., 1) on a device (GPU test).  No Python, no PyTorch in
that process: the C ABI is the product, the ctypes mirror only a second binding."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "dzoptimization.jl_amd")


def _build(tmp_path, name="readme_rosenbrock"):
    import __graft_entry__ as entry
    entry.build()
    exe = str(tmp_path / name)
    cmd = ["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", name + ".c"), "-L" + PKG, "-ldzo_hip",
           "-Wl,-rpath," + PKG, "-lm", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


@pytest.mark.parametrize("name", ["readme_rosenbrock", "batched_shards", "bench_lbfgs"])
def test_c_example_compiles_and_links(tmp_path, name):
    exe = _build(tmp_path, name)
    assert os.path.exists(exe)
    # every dzo_* symbol the example needs is resolved by libdzo_hip.so
    out = subprocess.run(["nm", "-u", exe], check=True, capture_output=True, text=True).stdout
    wanted = {l.split()[-1].split("@")[0] for l in out.splitlines() if " dzo_" in l}
    exported = subprocess.run(["nm", "-D", "--defined-only", os.path.join(PKG, "libdzo_hip.so")],
                              check=True, capture_output=True, text=True).stdout
    have = {l.split()[-1] for l in exported.splitlines()}
    assert wanted and wanted <= have, wanted - have


@pytest.mark.gpu
def test_c_example_runs_and_converges(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout


@pytest.mark.gpu
def test_c_batched_shards_example_runs_on_every_visible_gpu(tmp_path):
    """examples/batched_shards.c: one host process, one shard of independent BFGS instances per visible GPU
    (dzo_bfgs_batch_create_on), the convergence flag through dzo_comm_init_all + dzo_bfgs_batch_all_done
    (SURVEY 8(b)/(e)'s single-process form).  On the 1-GPU box: one shard, a one-rank communicator."""
    exe = _build(tmp_path, "batched_shards")
    r = subprocess.run([exe, "48", "16"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout and "all_done = 1" in r.stdout, r.stdout


@pytest.mark.gpu
def test_c_bench_driver_runs_the_same_trajectory_as_the_python_binding(tmp_path):
    """examples/bench_lbfgs.c: the headline workload through the C ABI alone (own PCG32, own host loop).  Same inputs, same
    library -> after the same number of steps the objective value is bit for bit what the ctypes binding gets."""
    import json
    import sys
    sys.path.insert(0, ROOT)
    import bench
    from dzo_loader import dzo
    n, m, steps, warm = 400_000, 20, 30, 5          # (a size whose run does not end stuck within these steps)
    exe = _build(tmp_path, "bench_lbfgs")
    r = subprocess.run([exe, str(n), str(m), str(steps), str(warm)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr
    line = json.loads(next(l for l in r.stdout.splitlines() if l.startswith("{")))
    assert line["iteration_count"] == m + warm + steps and line["value"] > 0
    dzo.init(0)
    opt = dzo.LBFGSOptimizer(None, dzo.Problem(dzo.ROSENBROCK_CHAIN, n), None, dzo.DeviceArray.from_host(bench.rosenbrock_chain_x0(n, seed=5)), 1.0, m)
    for _ in range(m + warm):
        opt.step()
    assert float("%.10e" % opt.current_objective_value) == line["f_start"]
    for _ in range(steps):
        opt.step()
    assert float("%.10e" % opt.current_objective_value) == line["f_end"]
