import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The CPU oracles are thousands of SMALL torch operators (a 19 x 19-tap bilateral filter under autograd is ~15 000 of them).  On a GPU box with
# 100+ host cores torch's default intra-op pool (one thread per core) turns each of them into a barrier across all cores: the same 40 x 40-pixel
# oracle render that takes 3 s on 8 threads took 350 s there (gpurun_out/r06a, round 6), and the GPU suite ran into the driver's limit in round 5.
# The C checkers (oracle/anyhit_c.c, oracle/_ref) keep their own OpenMP setting.
import torch  # noqa: E402

TORCH_THREADS = min(8, os.cpu_count() or 8)
torch.set_num_threads(TORCH_THREADS)


@pytest.fixture(autouse=True)
def _bounded_torch_threads():
    """Re-assert the bound before EVERY test: torch and the C checkers share one OpenMP runtime, and a checker that asks for all cores
    (refnative.set_threads(0) -> omp_set_num_threads(omp_get_num_procs())) would otherwise leave every later torch operator of the session on 100+
    threads (round 6: the suite passed in 1187 s with the render / bilateral oracles of 40 x 40 frames at 330 s each, and in ~250 s with this)."""
    torch.set_num_threads(TORCH_THREADS)
    yield


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def oracle_kernels():
    """The test runs against gshell_amd/lib/variants/oracles.so (= the shipped sources + the oracle / alternate-design kernels, csrc/common.hpp
    GS_ORACLE_KERNELS): for tests whose CHECKER is one of those kernels (exact-fp32 SDF forward, tangent-row eikonal, sampler-replay backward) or
    that keep an unshipped design correct (k_h1r_fwd)."""
    from gshell_amd import _lib
    if not os.path.isfile(_lib.variant_path("oracles")):
        pytest.skip("gshell_amd/lib/variants/oracles.so not built")
    with _lib.use_variant("oracles") as L:
        yield L
