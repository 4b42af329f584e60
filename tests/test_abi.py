"""C-ABI surface: the shared library loads without a GPU and exports every symbol include/cude.h declares;
the product path fails loudly (never falls back to a CPU path) when no GPU / no library is present."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "cude.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(cude_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from cude import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _header_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/cude.h but not exported"
    assert set(_lib.exported_symbols()) == set(names)


def test_n_params_matches_simplechains_layout():
    from cude.engine import n_params
    assert n_params(2, 4, 2) == 37      # c-peptide/02-conditional.jl:22
    assert n_params(2, 6, 2) == 67      # source_data/neural_network_parameters.jld2
    assert n_params(3, 4, 2) == 41      # c-peptide/07-covariate-inclusion.jl:32
    assert n_params(4, 3, 5) == 67      # suppression/suppression.jl:18
    assert n_params(1, 0, 0) == 1       # symbolic model: the literal 1.78 of c-peptide/03-symreg.jl:38


def test_no_silent_fallback_without_gpu():
    from cude import engine
    from cude._lib import CudeError
    try:
        has_gpu = engine.device_count() > 0
    except CudeError:
        has_gpu = False
    if has_gpu:
        pytest.skip("GPU present")
    with pytest.raises(CudeError):
        engine.Engine("cpep", (2, 6, 2))


def test_missing_library_is_an_error(monkeypatch):
    from cude import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libcude_hip.so")
    with pytest.raises(_lib.CudeError):
        _lib.load()


def test_bad_arguments_return_status_not_abort():
    from cude import _lib
    from cude._lib import CudeError
    lib = _lib.load()
    assert lib.cude_n_params(0, 4, 2) < 0
    assert b"bad network shape" in lib.cude_last_error()
    with pytest.raises(CudeError):
        _lib.check(lib.cude_create(None, None))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "conditional-ude_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".jl")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "cude_oracle" not in src and "c_oracle" not in src, os.path.join(dirpath, f)
