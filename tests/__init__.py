"""GPU (pytest -m gpu) and CPU (pytest -m "not gpu") tests of the HIP path, the oracles and the host logic."""
