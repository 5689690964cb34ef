import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "video-stitcher_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


os.environ.setdefault("MS_CHECK_DIVIDE", "1")      # ms_init_blender checks the shared-reciprocal division over the context's own denominators

try:        # hypothesis suites: a fresh seed per calendar day (printed in the header, reproducible with MS_TEST_SEED=n), not the same examples forever;
            # MS_TEST_FIXED=1 derandomises, MS_TEST_RANDOM=1 uses hypothesis' own entropy
    from hypothesis import settings as _hs
    _hs.register_profile("ms_fixed", derandomize=True, database=None)
    _hs.register_profile("ms_seeded", database=None, print_blob=True)
    _hs.load_profile("ms_fixed" if os.environ.get("MS_TEST_FIXED") == "1" else "ms_seeded")
    _HAVE_HYPOTHESIS = True
except ImportError:
    _HAVE_HYPOTHESIS = False


def _daily_seed():
    import time
    return int(os.environ.get("MS_TEST_SEED", int(time.time()) // 86400))


def pytest_report_header(config):
    if _HAVE_HYPOTHESIS and os.environ.get("MS_TEST_FIXED") != "1" and os.environ.get("MS_TEST_RANDOM") != "1":
        return "hypothesis seed of the day: %d (MS_TEST_SEED=%d reproduces)" % (_daily_seed(), _daily_seed())


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if _HAVE_HYPOTHESIS and os.environ.get("MS_TEST_FIXED") != "1" and os.environ.get("MS_TEST_RANDOM") != "1":
        if getattr(config.option, "hypothesis_seed", None) is None:
            try:
                config.option.hypothesis_seed = str(_daily_seed())
            except Exception:
                pass


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
