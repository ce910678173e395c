"""gshell_amd -- MI355X-native (HIP/gfx950) hot path for G-Shell inverse rendering.

Drop-in for the reference's geometry.gshell_tets_geometry / render.render path
(see DESIGN.md for the boundary and INTEGRATION.md for how the reference binds it).
"""
__version__ = "0.1.0"
