import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "video-stitcher_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


try:        # hypothesis tests draw the same examples on every run (a round-end `pytest -x` must not depend on luck); MS_TEST_RANDOM=1 explores
    from hypothesis import settings as _hs
    _hs.register_profile("ms_fixed", derandomize=True, database=None)
    _hs.register_profile("ms_random", database=None)
    _hs.load_profile("ms_random" if os.environ.get("MS_TEST_RANDOM") == "1" else "ms_fixed")
except ImportError:
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): built on demand with gcc."""
    import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def ms():
    """The product library through its C-ABI; fails loudly if it is not built."""
    import msstitch
    msstitch.load()
    return msstitch


@pytest.fixture(scope="session")
def cuda(ms):
    import torch
    assert torch.cuda.is_available(), "gpu-marked test running without a GPU"
    assert ms.device_count() > 0, "libmsstitch sees no HIP device"
    return torch.device("cuda:0")
