"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads
without a GPU, exports every symbol include/dzo.h declares, and fails loudly (no CPU
fallback) when there is no device.  No compute calls here."""
import ctypes
import os
import re

import pytest

from dzo_loader import dzo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "dzo.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dzo_[a-z0-9_]+)\s*\(", text)) - {"dzo_constraint_fn", "dzo_objective_fn", "dzo_gradient_fn"})


def test_library_builds_and_exports_every_declared_symbol():
    path = dzo.build()
    lib = ctypes.CDLL(path)
    names = _declared_symbols()
    assert len(names) >= 70
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"include/dzo.h declares symbols the library lacks: {missing}"


def test_python_abi_table_covers_the_header():
    declared = set(_declared_symbols())
    bound = set(dzo.ABI) | {"dzo_last_error", "dzo_version"}
    assert declared == bound, (declared - bound, bound - declared)


def test_version_and_error_string_without_gpu():
    lib = dzo.lib()
    assert lib.dzo_version() == 100
    assert isinstance(lib.dzo_last_error(), bytes)


def test_no_cpu_fallback_product_fails_loudly_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(dzo.DzoError) as e:
        dzo.init(0)
    assert "no CPU path" in str(e.value) or "HIP" in str(e.value)
    with pytest.raises(dzo.DzoError):
        dzo.DeviceArray(8)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under the package or include/ may name it."""
    pkg = os.path.join(ROOT, "dzoptimization.jl_amd")
    for d, _, files in os.walk(pkg):
        if "build" in d:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".jl", "Makefile")):
                text = open(os.path.join(d, f), errors="replace").read()
                for bad in ("libdzo_oracle", "from oracle", "import oracle", "orc_"):
                    assert bad not in text, f"{f} references the oracle ({bad})"


def test_kernels_were_compiled_for_gfx950():
    import subprocess
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", dzo.LIB_PATH],
                         capture_output=True, text=True).stdout
    blob = open(dzo.LIB_PATH, "rb").read()
    assert b"gfx950" in blob, out[:200]
