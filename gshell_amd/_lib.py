"""ctypes binding of libgshell_hip.so (the C ABI declared in include/gshell_hip.h).

The product path has NO fallback: if the HIP library is missing or fails to load this
module raises, loudly.  Build it with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C gshell_amd/csrc -j`.
"""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# GSHELL_HIP_LIB: an alternative build of the same library (tools/build_variant.sh: compile-time experiments built on the CPU box)
LIB_PATH = os.environ.get("GSHELL_HIP_LIB") or os.path.join(_HERE, "lib", "libgshell_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "gshell_hip.h")

_lib = None

c_void_p, c_int64, c_int, c_float = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float


class GShellHipError(RuntimeError):
    pass


def declared_prototypes():
    """{name: return type} of every entry point declared in the public header."""
    with open(HEADER_PATH) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return {m.group(2): m.group(1) for m in re.finditer(r"\b(int64_t|int|const char\s*\*)\s+(gs_[a-z0-9_]+)\s*\(", src)}


def declared_symbols():
    return sorted(declared_prototypes())


def lib():
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise GShellHipError(
                f"{LIB_PATH} not found: the gfx950 HIP library is not built. "
                "Run `make -C gshell_amd/csrc -j` (or __graft_entry__.build()). There is no CPU fallback.")
        try:
            _lib = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise GShellHipError(f"failed to load {LIB_PATH}: {e}") from e
        _lib.gs_last_error.restype = ctypes.c_char_p
        for name, ret in declared_prototypes().items():
            fn = getattr(_lib, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype = {"int": c_int, "int64_t": c_int64}.get(ret, ctypes.c_char_p)
    return _lib


def _load(path):
    L = ctypes.CDLL(path)
    L.gs_last_error.restype = ctypes.c_char_p
    for name, ret in declared_prototypes().items():
        fn = getattr(L, name)
        fn.restype = {"int": c_int, "int64_t": c_int64}.get(ret, ctypes.c_char_p)
    return L


_variants = {}


def variant_path(name):
    return os.path.join(_HERE, "lib", "variants", f"{name}.so")


class use_variant:
    """`with use_variant("oracles"):` -- inside the block every entry point resolves in gshell_amd/lib/variants/<name>.so instead of the shipped library.
    "oracles" = the same sources built with -DGS_ORACLE_KERNELS=1 (csrc/common.hpp): the exact-fp32 SDF forward, the register-resident one-product
    forward, the tangent-row eikonal instantiations and the sampler-replay shading backward, which tests compare the shipped kernels with.  Objects made
    by one library (BVH, topologies, packed weights) are plain device / host data of identical layout and may be handed to the other."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        global _lib
        path = variant_path(self.name)
        if self.name not in _variants:
            if not os.path.isfile(path):
                raise GShellHipError(f"{path} not built (make -C gshell_amd/csrc builds it next to the shipped library)")
            _variants[self.name] = _load(path)
        lib()                                   # make sure the shipped library is what we swap back to
        self._saved, _lib = _lib, _variants[self.name]
        return _variants[self.name]

    def __exit__(self, *exc):
        global _lib
        _lib = self._saved
        return False


def check(status, what=""):
    if status != 0:
        msg = lib().gs_last_error().decode("utf-8", "replace")
        raise GShellHipError(f"{what}: {msg}" if what else msg)


def ptr(t, dtype=None, name="tensor"):
    """Device pointer of a contiguous CUDA(HIP) tensor (None -> NULL).  The CALLER must keep `t` referenced until the
    launch has been enqueued: never pass an inline temporary (`ptr(x.int())`) -- torch frees it when ptr() returns and the
    next temporary may be handed the same block.  Mirrors the
    reference's CHECK_TENSOR guards (render/renderutils/c_src/torch_bindings.cpp:27-31)."""
    if t is None:
        return c_void_p(0)
    if not t.is_cuda:
        raise GShellHipError(f"{name} must live in HBM (got device {t.device}); the HIP path has no CPU fallback")
    if dtype is not None and t.dtype != dtype:
        raise GShellHipError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise GShellHipError(f"{name} must be contiguous")
    return c_void_p(t.data_ptr())


def stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


# ---- optional per-op timing (bench.py): HIP events recorded on the launch stream around each C-ABI call -------------
_timing = {"on": False, "only": None, "pending": [], "done": {}}


class _TimedLib:
    """Proxy over the ctypes library that brackets every gs_* launch with torch.cuda events on the current stream
    (the stream the kernels are launched on).  Only used when bench.py enables it."""

    def __init__(self, real):
        object.__setattr__(self, "_real", real)

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if not _timing["on"] or not name.startswith("gs_") or name.endswith(("_bytes", "_partials", "_params", "_info", "_create", "_destroy", "_padded", "_words", "last_error", "version")):
            return fn
        if _timing["only"] is not None and name not in _timing["only"]:
            return fn

        def timed(*args):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args)
            e1.record()
            _timing["pending"].append((name, e0, e1))
            return rc
        return timed


_real_lib_fn = lib


def lib():      # noqa: F811  (wraps the loader defined above)
    real = _real_lib_fn()
    return _TimedLib(real) if _timing["on"] else real


def enable_op_timing(flag, only=None):
    """only: a set of entry-point names -- nothing else is bracketed with events (bench.py's default run times just the candidates
    for the dominant kernel; --op-times times everything)"""
    _timing["on"] = bool(flag)
    _timing["only"] = None if only is None else set(only)


def reset_op_timing():
    torch.cuda.synchronize()
    _timing["pending"].clear()
    _timing["done"].clear()


def op_timing_summary():
    """{op: {"ms": mean launch-to-completion ms, "n": launches}} since the last reset."""
    torch.cuda.synchronize()
    for name, e0, e1 in _timing["pending"]:
        rec = _timing["done"].setdefault(name, {"ms": 0.0, "n": 0})
        rec["ms"] += e0.elapsed_time(e1)
        rec["n"] += 1
    _timing["pending"].clear()
    return {k: {"ms": v["ms"] / max(v["n"], 1), "n": v["n"]} for k, v in _timing["done"].items()}
