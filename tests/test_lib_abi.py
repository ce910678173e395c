"""CPU-side checks of the C-ABI boundary: the library loads and exports every declared symbol."""
import ctypes
import os

import pytest

from gshell_amd import _lib


def test_library_is_built():
    assert os.path.isfile(_lib.LIB_PATH), "run __graft_entry__.build() first"


def test_every_declared_symbol_is_exported():
    names = _lib.declared_symbols()
    assert len(names) >= 9
    L = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, f"include/gshell_hip.h declares symbols the .so lacks: {missing}"


def test_version_and_error_string():
    L = _lib.lib()
    assert L.gs_version() >= 100
    assert isinstance(L.gs_last_error(), bytes)


def test_product_path_refuses_cpu_tensors():
    import torch
    with pytest.raises(_lib.GShellHipError):
        _lib.ptr(torch.zeros(4), torch.float32, "x")


def test_the_in_tree_library_is_the_shipped_build():
    """gs_build_flags() lists every GS_TUNABLE that is off its default and every GS_EXPERIMENT block compiled in (csrc/common.hpp);
    the library that travels to the GPU box must report none.  Host-side registry: no device call."""
    L = _lib.lib()
    flags = L.gs_build_flags().decode()
    assert flags == "", f"libgshell_hip.so was built with non-default compile-time flags: {flags}"
    assert len(_lib.declared_symbols()) >= 130


def test_oracle_variant_reports_itself_and_the_shipped_library_holds_one_design_per_stage():
    """lib/variants/oracles.so = the same sources + the oracle / alternate-design kernels (csrc/common.hpp GS_ORACLE_KERNELS): it exports the same
    ABI and announces itself through gs_build_flags() (bench.py refuses such a library); the SHIPPED library refuses to select the register-resident
    forward (host-side calls only: no device work)."""
    path = _lib.variant_path("oracles")
    if not os.path.isfile(path):
        pytest.skip("variant not built")
    V = _lib._load(path)
    assert "GS_ORACLE_KERNELS=1" in V.gs_build_flags().decode()
    assert not [n for n in _lib.declared_symbols() if not hasattr(V, n)]
    assert V.gs_sdf_mlp_h1_impl(ctypes.c_int(-1)) == 0
    L = _lib._load(_lib.LIB_PATH)
    assert L.gs_sdf_mlp_h1_impl(ctypes.c_int(1)) == -1 and b"alternate-design" in L.gs_last_error()
    assert L.gs_sdf_mlp_h1_impl(ctypes.c_int(-1)) == 0
    with _lib.use_variant("oracles") as inside:
        assert "GS_ORACLE_KERNELS=1" in inside.gs_build_flags().decode() and _lib.lib().gs_build_flags() == inside.gs_build_flags()
    assert _lib.lib().gs_build_flags().decode() == ""
