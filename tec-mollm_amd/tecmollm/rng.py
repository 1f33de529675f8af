"""NumPy mirror of the device dropout mask (csrc/common.h: tecm_hash24 / tecm_drop_mult) so that tests can
feed the *same* masks to an independent implementation."""
from __future__ import annotations

import numpy as np

_G = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def hash24(seed: int, idx: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx.astype(np.uint64) * _G
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(40)).astype(np.uint32)


def keep_mult(seed: int, idx: np.ndarray, p: float) -> np.ndarray:
    """Multiplier applied by the kernels: 0 where dropped, 1/(1-p) where kept (fp32 like the device)."""
    thresh = np.uint32(np.float32(p) * np.float32(16777216.0))
    inv = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    return np.where(hash24(seed, idx) >= thresh, inv, np.float32(0.0)).astype(np.float32)
