import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "video-stitcher_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _memoise_synth_frames():
    """synth.frame is pure numpy in fp64: 0.2 s for a 1080p frame, about a second for a 4K one -- a dozen cfg5 tests regenerated the same 12 x 4K frames (5 - 8 s each, most of
    their run time).  Same bytes, generated once per session; the tests only read them (to_dev copies to the device; none edits a frame in place)."""
    import functools
    import synth
    orig = synth.frame

    @functools.lru_cache(maxsize=160)
    def cached(w, h, i, t, noise):
        return orig(w, h, i, t, noise)

    def frame(w, h, i, t, noise=True):
        if w * h < 1280 * 720:
            return orig(w, h, i, t, noise)
        return cached(w, h, i, t, noise)
    synth.frame = frame


_memoise_synth_frames()
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


# Whole-pipeline tests (rigs, frames, masks as the path produces them): the oracle must never evaluate static_cast<short>(float) outside the int16
# range there -- outside it the two conversions nvcc may emit (through int32, or the saturating cvt.rzi.s16.f32) differ and the oracle's convention
# would be a guess (ms_oracle_prims.c, trunc_s16f).  Per-op tests feed arbitrary int16 / float values on purpose and are exempt.
_PIPELINE_MODULES = {"test_compositor_gpu", "test_from_inputs_gpu", "test_random_rigs_gpu", "test_reference_sequence_gpu", "test_tables_gpu",
                     "test_host_app_gpu", "test_bench_gpu", "test_ms_dist_gpu"}


@pytest.fixture(autouse=True)
def _oracle_stays_inside_int16(request):
    if request.module.__name__ not in _PIPELINE_MODULES:
        yield
        return
    import oracle as O
    O.lib().orc_trunc_s16_range_reset()
    yield
    n = O.trunc_s16_range_violations()
    assert n == 0, "the oracle truncated %d float values outside the int16 range to short in %s: its nvcc convention is unpinned there" % (n, request.node.name)


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
