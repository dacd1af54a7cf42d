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


@pytest.fixture(scope="session")
def golden_legacy():
    """Reference-source outputs of the two exported-but-unused processor classes in their garment forms: ``SAttnProcessor2_0`` (one
    softmax over [self; garment] keys) and ``RefCAttnProcessor2_0`` (oracle/make_golden.py::main_legacy)."""
    import torch
    return torch.load(os.path.join(GOLDEN, "processors_legacy.pt"), weights_only=False)


_KNOB_NAMES = ("PATCH_CONV", "SPLITK_IN_KERNEL", "FUSED_FF", "FUSED_GN_STATS", "CFG_PAIR_DEDUP", "CFG_PAIR_ATTN", "FUSED_GN_FINISH", "FUSED_GN_CONV", "FUSED_GN_PROJ", "FUSED_LN", "FUSED_OUT_PROJ", "ATTN_FP8")


@pytest.fixture(autouse=True)
def _restore_tuning_knobs():
    """The library's tuning knobs (imd_set_tuning) and the ops-level switches are PROCESS-GLOBAL: a test that flips one and fails (or
    forgets) would silently change every later test.  Snapshot before, restore after -- every test starts from the shipped settings."""
    from imagdressing_amd import _lib, ops
    py = {k: getattr(ops, k) for k in _KNOB_NAMES if hasattr(ops, k)}
    lib = _lib._lib                      # only if some earlier test already loaded it (never force a load here)
    native = None if lib is None else [lib.imd_get_tuning(k) for k in range(3)]
    yield
    for k, v in py.items():
        setattr(ops, k, v)
    lib = _lib._lib
    if lib is not None and native is not None:
        for k, v in enumerate(native):
            if lib.imd_get_tuning(k) != v:
                lib.imd_set_tuning(k, v)
