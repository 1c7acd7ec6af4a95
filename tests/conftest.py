import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def hipb():
    """The HIP backend.  No skip: on a GPU box a missing GPU/library must fail loudly."""
    import video_filler_amd  # noqa: F401
    from video_filler_amd.backend import get_backend
    return get_backend()
