import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle is many small fp32 ops: on the GPU box's 100+ host cores torch's default thread count makes every one of
    # them slower (the same effect bench.py's cpu_baseline measured: 335 s vs 7 s per DDIM step) -- and the GPU sits idle meanwhile.
    import torch
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))


@pytest.fixture(scope="session")
def golden_processors():
    import torch
    return torch.load(os.path.join(GOLDEN, "processors.pt"), weights_only=False)


@pytest.fixture(scope="session")
def golden_resampler():
    import torch
    return torch.load(os.path.join(GOLDEN, "resampler.pt"), weights_only=False)


@pytest.fixture(scope="session")
def golden_full():
    """Reference-source outputs at the benchmarked kernel shape (N = M = 4096, d = 40), a spiked ragged case and
    CacheAttnProcessor2_0 at real head dims (oracle/make_golden.py::main_full)."""
    import torch
    return torch.load(os.path.join(GOLDEN, "processors_full.pt"), weights_only=False)


@pytest.fixture(scope="session")
def golden_geometry():
    """Reference-source outputs at the token counts of the reference scripts' default geometry, 512 wide x 640 high with a
    640 x 512 garment (inference_IMAGdressing.py:182-183): (C, N = M) = (320, 5120), (640, 1280), (1280, 320), (1280, 80)
    (oracle/make_golden.py::main_geometry)."""
    import torch
    return torch.load(os.path.join(GOLDEN, "processors_default_geometry.pt"), weights_only=False)
