"""Deterministic synthetic tensors shared by the golden-fixture generator and the tests
(TEST INFRASTRUCTURE).  numpy PCG64 streams -> float32, so the same (seed, shape, scale) gives
bit-identical tensors wherever this image runs; fixtures record a checksum of every regenerated
input so a stream change is detected instead of being misread as a parity failure."""
from __future__ import annotations

import hashlib

import numpy as np
import torch


def seeded(seed: int, *shape, scale: float = 1.0) -> torch.Tensor:
    a = np.random.default_rng(seed).standard_normal(shape, dtype=np.float32) * np.float32(scale)
    return torch.from_numpy(a)


def digest(t: torch.Tensor) -> str:
    return hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()[:16]
