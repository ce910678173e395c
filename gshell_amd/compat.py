"""Drop-in shim: make the reference's own import names resolve to this package, so that a script written against
lzzcd001/GShell (`from geometry.gshell_tets_geometry import GShellTetsGeometry`, `from render import render, light`,
`import nvdiffrast.torch as dr`, `import tinycudann as tcnn`, `from denoiser.denoiser import BilateralDenoiser` ...)
runs on the MI355X path unmodified.

    import gshell_amd.compat; gshell_amd.compat.install()      # before the reference-style imports

Modules of the reference that are OUT OF SCOPE here (datasets, obj/material IO, texture.Texture2D, xatlas) are not
aliased: put the reference tree behind this package on sys.path if a script needs them."""
import importlib
import sys
import types

_ALIASES = {
    "geometry": "gshell_amd.geometry",
    "geometry.gshell_tets": "gshell_amd.geometry.gshell_tets",
    "geometry.gshell_tets_geometry": "gshell_amd.geometry.gshell_tets_geometry",
    "geometry.gshell_flexicubes": "gshell_amd.geometry.gshell_flexicubes",
    "geometry.gshell_flexicubes_geometry": "gshell_amd.geometry.gshell_flexicubes_geometry",
    "geometry.mlp": "gshell_amd.geometry.mlp",
    "render": "gshell_amd.render",
    "render.render": "gshell_amd.render.render",
    "render.light": "gshell_amd.render.light",
    "render.mesh": "gshell_amd.render.mesh",
    "render.util": "gshell_amd.render.util",
    "render.mlptexture": "gshell_amd.render.mlptexture",
    "render.regularizer": "gshell_amd.render.regularizer",
    "render.renderutils": "gshell_amd.render.renderutils",
    "render.optixutils": "gshell_amd.render.optixutils",
    "denoiser": "gshell_amd.denoiser",
    "denoiser.denoiser": "gshell_amd.denoiser.denoiser",
    "nvdiffrast.torch": "gshell_amd.render.rast",
}


def install(force=False):
    """Register the aliases in sys.modules (idempotent).  Existing modules of the same name are kept unless force=True."""
    for ref_name, ours in _ALIASES.items():
        if ref_name in sys.modules and not force:
            continue
        sys.modules[ref_name] = importlib.import_module(ours)
    if "nvdiffrast" not in sys.modules or force:
        pkg = types.ModuleType("nvdiffrast")
        pkg.torch = sys.modules["nvdiffrast.torch"]
        sys.modules["nvdiffrast"] = pkg
    if "tinycudann" not in sys.modules or force:
        tcnn = types.ModuleType("tinycudann")
        tcnn.Encoding = sys.modules["render.mlptexture"].HashGridEncoding
        tcnn.free_temporary_memory = lambda: None
        sys.modules["tinycudann"] = tcnn
    if "kaolin" not in sys.modules or force:      # kaolin.ops.mesh.sample_points (gshell_tets_geometry.py:236)
        geo = sys.modules["geometry.gshell_tets_geometry"]
        kaolin, ops, meshmod = types.ModuleType("kaolin"), types.ModuleType("kaolin.ops"), types.ModuleType("kaolin.ops.mesh")

        def sample_points(vertices, faces, num_samples):
            pts, fid = geo.sample_points(vertices[0], faces, num_samples)
            return pts[None], fid[None]
        meshmod.sample_points = sample_points
        ops.mesh, kaolin.ops = meshmod, ops
        sys.modules.update({"kaolin": kaolin, "kaolin.ops": ops, "kaolin.ops.mesh": meshmod})
    return sorted(_ALIASES)
