import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _host_eigh_gauge():
    """Parity tests compare particle coordinates and sampler draws with the CPU oracle, so the eigendecompositions of setup
    and prediction are pinned to host LAPACK (the oracle's eigenvector gauge).  The library default is "auto" (where the
    matrix lives, like the reference's torch.linalg.eigh) and normals from the device generator; the device paths have their
    own gauge-invariant / statistical tests."""
    from projected_langevin_sampling_amd import samplers

    prev = (samplers.DEFAULT_EIGH_DEVICE, samplers.DEFAULT_NORMAL_STREAM)
    samplers.DEFAULT_EIGH_DEVICE = "cpu"
    samplers.DEFAULT_NORMAL_STREAM = "reference"  # the reference's host torch.normal stream, sample for sample
    yield
    samplers.DEFAULT_EIGH_DEVICE, samplers.DEFAULT_NORMAL_STREAM = prev
