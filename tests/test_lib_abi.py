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
