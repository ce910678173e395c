"""Loader for the *real* reference modules (test infrastructure only).

ORACLE / TEST INFRASTRUCTURE -- never imported by the product path (gshell_amd/).

Loads pure-PyTorch reference files from /root/reference on CPU without touching
the reference tree (hazards 1 and 2):
  * sources are read as text and exec'd into fresh module objects (no
    __pycache__, no cpp_extension hipify side effects),
  * `render.util` / nvdiffrast / imageio are stubbed before anything is exec'd,
  * a TorchFunctionMode rewrites device='cuda' -> 'cpu' and Tensor.cuda() -> id.

Only usable inside the build container (the GPU box has no /root/reference);
golden fixtures produced through this loader are committed under tests/golden/.
"""
import os
import sys
import types
import torch
from torch.overrides import TorchFunctionMode

REF_ROOT = os.environ.get("GSHELL_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isfile(os.path.join(REF_ROOT, "geometry", "gshell_tets.py"))


class CudaToCpu(TorchFunctionMode):
    """Make the reference's hard-coded device='cuda' / .cuda() run on CPU."""

    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        dev = kwargs.get("device", None)
        if dev is not None and "cuda" in str(dev):
            kwargs["device"] = "cpu"
        if func is torch.Tensor.cuda:
            return args[0]
        return func(*args, **kwargs)


def _exec_module(name, relpath, extra_globals=None):
    path = os.path.join(REF_ROOT, relpath)
    with open(path, "r") as f:
        src = f.read()
    mod = types.ModuleType(name)
    mod.__file__ = path
    if extra_globals:
        mod.__dict__.update(extra_globals)
    code = compile(src, path, "exec", dont_inherit=True)
    exec(code, mod.__dict__)
    return mod


def _util_stub():
    """Minimal stand-in for render/util.py (reference render/util.py:19-31);
    only dot / safe_normalize are used by geometry/gshell_tets.py."""
    m = types.ModuleType("render.util")

    def dot(x, y):
        return torch.sum(x * y, -1, keepdim=True)

    def length(x, eps=1e-20):
        return torch.sqrt(torch.clamp(dot(x, x), min=eps))

    def safe_normalize(x, eps=1e-20):
        return x / length(x, eps)

    m.dot, m.length, m.safe_normalize = dot, length, safe_normalize
    return m


def load_gshell_tets():
    """Returns the reference module geometry/gshell_tets.py (exec'd on CPU)."""
    saved = {k: sys.modules.get(k) for k in ("render", "render.util")}
    render_pkg = types.ModuleType("render")
    util = _util_stub()
    render_pkg.util = util
    sys.modules["render"] = render_pkg
    sys.modules["render.util"] = util
    try:
        with CudaToCpu():
            mod = _exec_module("ref_gshell_tets", "geometry/gshell_tets.py")
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod


def load_flexicubes():
    with CudaToCpu():
        tables = _exec_module("ref_flexicubes_table", "geometry/flexicubes_table.py")
        saved = sys.modules.get("geometry.flexicubes_table")
        pkg = types.ModuleType("geometry")
        sys.modules.setdefault("geometry", pkg)
        sys.modules["geometry.flexicubes_table"] = tables
        try:
            path = os.path.join(REF_ROOT, "geometry", "gshell_flexicubes.py")
            src = open(path).read().replace("from .flexicubes_table import", "from geometry.flexicubes_table import")
            mod = types.ModuleType("ref_gshell_flexicubes")
            mod.__file__ = path
            exec(compile(src, path, "exec", dont_inherit=True), mod.__dict__)
        finally:
            if saved is None:
                sys.modules.pop("geometry.flexicubes_table", None)
    return mod


def load_simple(relpath, name=None):
    """Load a dependency-free reference file (e.g. render/renderutils/loss.py)."""
    with CudaToCpu():
        return _exec_module(name or ("ref_" + os.path.basename(relpath)[:-3]), relpath)
