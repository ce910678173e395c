"""Bilateral denoiser module with the reference interface (denoiser/denoiser.py)."""
