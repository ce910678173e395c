"""Raster / shading stage with the call surface of the reference render package (render.render, renderutils, optixutils, ...)."""
