"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and
exports every symbol include/msckf_mi355x.h declares; without a GPU the product
path fails loudly (no CPU fallback)."""
import os
import re

import pytest

from conftest import ROOT


def header_functions():
    txt = open(os.path.join(ROOT, "include", "msckf_mi355x.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(msckf_[a-z_0-9]+)\s*\(", txt)))


def test_header_symbols_are_bound_and_exported(engine_lib):
    from msckf_amd import _ffi
    names = header_functions()
    assert len(names) >= 20
    assert set(names) == set(_ffi.SYMBOLS)
    for n in names:
        assert hasattr(engine_lib, n), n


def test_struct_sizes_match_header():
    import ctypes as C
    from msckf_amd import _ffi
    assert C.sizeof(_ffi.Config) == 10 * 4          # ABI v2: + dtype, reserved0
    assert _ffi.Config.dtype.offset == 32 and _ffi.ABI_VERSION == 2
    assert C.sizeof(_ffi.Stats) == 8 * 4 + 8 * 4
    assert C.sizeof(_ffi.SelectParamsC) == 6 * 4 + 8 + 9 * 8
    assert _ffi.SelectParamsC.min_parallax_deg.offset == 24 and _ffi.SelectParamsC.K.offset == 32


def test_no_cpu_fallback(engine_lib):
    """On a box without a gfx950 the engine refuses to come up."""
    from msckf_amd import _ffi
    from msckf_amd.api import UpdateEngine
    if engine_lib.msckf_device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(_ffi.EngineError) as e:
        UpdateEngine()
    assert e.value.code == _ffi.ERR_NO_DEVICE


def test_strerror(engine_lib):
    assert engine_lib.msckf_strerror(0) == b"ok"
    assert b"gate" in engine_lib.msckf_strerror(1)
    assert b"slot" in engine_lib.msckf_strerror(-6)


def test_product_package_never_imports_oracle():
    """The oracle is test infrastructure: nothing under the product package, bench's
    GPU leg, or the ABI sources may reference it."""
    pkg = os.path.join(ROOT, "monocular-visual-inertial-msckf_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "msckf_oracle" not in src, f
