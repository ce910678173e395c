"""Geometry stage: SDF network, G-MarchingTets / G-FlexiCubes extractors and the two geometry classes of the reference."""
