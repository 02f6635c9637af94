"""The plain-C consumer of include/dzo.h (examples/readme_rosenbrock.c = the reference's README
example, README.md:25-41): it must compile and link against libdzo_hip.so with gcc alone
(CPU test), and converge to (1, ..., 1) on a device (GPU test).  No Python, no PyTorch in
that process: the C ABI is the product, the ctypes mirror only a second binding."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "dzoptimization.jl_amd")


def _build(tmp_path):
    import __graft_entry__ as entry
    entry.build()
    exe = str(tmp_path / "readme_rosenbrock")
    cmd = ["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "readme_rosenbrock.c"), "-L" + PKG, "-ldzo_hip",
           "-Wl,-rpath," + PKG, "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_c_example_compiles_and_links(tmp_path):
    exe = _build(tmp_path)
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
