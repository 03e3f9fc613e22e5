"""CPU-side checks of the C ABI: the library loads and exports every symbol include/dvg.h declares."""
import os
import re

import pytest

from image_generation_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "dvg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dvg_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    _lib.build()
    handle = _lib.lib()
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in include/dvg.h but not exported"
    # and the binding table covers exactly the header
    assert sorted(_lib.SIGNATURES) == declared


def test_version_and_error_string():
    L = _lib.lib()
    assert L.dvg_version() >= 100
    assert isinstance(L.dvg_last_error(), bytes)
    assert L.dvg_prof_num_kernels() > 10
    names = {L.dvg_prof_kernel_name(i).decode() for i in range(L.dvg_prof_num_kernels())}
    assert "gibbs_sweeps" in names and "conv_igemm_kernel<64,64,2,2,1>" in names


def test_argument_validation_without_gpu():
    """Bad arguments are rejected before any device work (safe on a CPU-only box)."""
    L = _lib.lib()
    rc = L.dvg_adam_step(None, None, None, None, 10, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, 1.0, None, 0, None)
    assert rc == -1 and b"null" in L.dvg_last_error()
    rc = L.dvg_gibbs_sample(None, None, None, 1.0, -1, 1, -1, 1, 1.0, None, 4, 0, 0, 0, 1, 1, None, None, None)
    assert rc == -1


def test_missing_library_is_loud(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.DvgError, match="no CPU fallback"):
        _lib.lib()
