import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Randomised-geometry tests draw the same examples on every run (no example database, no deadline): a failure found here is
    # reproducible as-is on the GPU box.  UG_HYPOTHESIS_RANDOM=1 explores new examples.
    try:
        from hypothesis import settings
        settings.register_profile("repro", derandomize=True, deadline=None, database=None, print_blob=True)
        settings.register_profile("explore", deadline=None, database=None, print_blob=True)
        settings.load_profile("explore" if os.environ.get("UG_HYPOTHESIS_RANDOM") else "repro")
    except ImportError:
        pass


@pytest.fixture(scope="session")
def po():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def hip():
    """The product library + torch device plumbing; GPU tests only."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible")
    from ultragrid_amd import codec, lib
    lib.load()  # fails loudly if the HIP library is missing -- there is no fallback
    return codec
